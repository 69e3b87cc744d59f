#!/bin/bash
# Test infrastructure: builds oracle/_ref/libwisp_ref.so from the reference's own kernel sources where they lie
# (only possible where /root/reference exists; the .so then travels to the GPU box with the repo snapshot).
# Nothing from the reference is copied into tracked files: the extracted fragments live in the git-ignored oracle/_ref/.
set -e
REF=${WISP_REFERENCE:-/root/reference}
HERE="$(cd "$(dirname "$0")" && pwd)"
SRC="$REF/wisp/csrc/ops"
[ -f "$SRC/hashgrid_interpolate_cuda.cu" ] || { echo "reference sources not present at $REF - skipping"; exit 0; }
mkdir -p "$HERE/_ref"
# kernels only: stop before the ATen launcher functions (which need CUDA's <<<>>> syntax)
# (a tap macro after the line that floors the scaled position lets ref_wrap.cpp read the cell the kernel computed:
#  REF_TAP expands to nothing unless a tap buffer is armed; the reference's own expressions are left untouched)
awk '/^void hashgrid_interpolate_cuda_impl\(/{exit} {print}' "$SRC/hashgrid_interpolate_cuda.cu" \
  | sed -e 's/^\( *\)int3 pos = make_int3(floor(x.x), floor(x.y), floor(x.z));$/&\n\1REF_TAP3(i, x, pos);/' > "$HERE/_ref/hashgrid_kernels.inc"
grep -c "REF_TAP3" "$HERE/_ref/hashgrid_kernels.inc" > /dev/null || { echo "tap injection failed"; exit 1; }
# uniform sampler: kernel only; give the per-thread body an explicit thread index (tidx is the kernel's only use of the grid)
awk '/^std::vector<at::Tensor> uniform_sample_cuda_impl\(/{exit} {print}' "$SRC/uniform_sample_cuda.cu" \
  | sed -e 's/^uniform_sample_cuda_kernel(/uniform_sample_cuda_kernel_at(uint tidx_in,/' \
        -e 's/uint tidx = blockDim.x \* blockIdx.x + threadIdx.x;/uint tidx = tidx_in;/' > "$HERE/_ref/uniform_kernels.inc"
# corner query (no blend): forward kernel = everything before its ATen launcher; backward kernel = the template between the two launchers
awk '/^void hashgrid_query_cuda_impl\(/{exit} {print}' "$SRC/hashgrid_query_cuda.cu" \
  | sed -e 's/^#include "hash_utils.cuh"//' > "$HERE/_ref/query_fwd_kernel.inc"
awk '/^hashgrid_query_backward_cuda_kernel\(/{f=1; print "template<typename scalar_t>\n__global__ void"} f&&/^void hashgrid_query_backward_cuda_impl\(/{exit} f{print}' \
  "$SRC/hashgrid_query_cuda.cu" > "$HERE/_ref/query_bwd_kernel.inc"
# depth-bound search of the SDF tracer: kernel only, explicit thread index (no grid-stride loop in the kernel)
awk '/^void find_depth_bound_cuda_impl\(/{exit} {print}' "$REF/wisp/csrc/render/find_depth_bound_cuda.cu" \
  | sed -e 's/^find_depth_bound_cuda_kernel(/find_depth_bound_cuda_kernel_at(uint tidx_in,/' \
        -e 's/uint tidx = blockDim.x \* blockIdx.x + threadIdx.x;/uint tidx = tidx_in;/' > "$HERE/_ref/depth_bound_kernel.inc"
g++ -O2 -std=c++17 -shared -fPIC -ffp-contract=off -I "$HERE/ref_shim" -I "$SRC" "$HERE/ref_wrap.cpp" -o "$HERE/_ref/libwisp_ref.so"
echo "built $HERE/_ref/libwisp_ref.so"
