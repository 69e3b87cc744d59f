#!/bin/bash
# A/B of library variants (kaolin-wisp_amd/csrc/ab/*.so against the in-tree build) on the hash-grid / decoder kernels
export TMPDIR=/tmp
mkdir -p gpurun_out
CS=$PWD/kaolin-wisp_amd/csrc
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "hashgrid or flagship or stress or spill or golden" > gpurun_out/pytest_hg.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_hg.log
tail -4 gpurun_out/pytest_hg.log
run() { WISP_HIP_LIB=$1 timeout 300 python scripts/ab_kernels.py 2>&1 | grep -v amdgpu.ids | tail -1; }
for rep in 1 2 3; do
run $CS/libwisp_hip.so
for lib in $(ls $CS/ab/*.so 2>/dev/null); do run $lib; done
done | tee gpurun_out/ab_hg2.log
