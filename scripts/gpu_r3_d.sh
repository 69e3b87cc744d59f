#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3d
OUT=gpurun_out/r3d
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "decoder or mlp or direct_step or flagship or per_ray" > $OUT/pytest_mlp.log 2>&1
echo "pytest exit: $?" >> $OUT/pytest_mlp.log; tail -6 $OUT/pytest_mlp.log | cut -c1-300
for rep in 1 2 3; do
  WISP_MLP_BWD_DUAL=0 timeout 300 python scripts/ab_kernels.py 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/single /"
  WISP_MLP_BWD_DUAL=1 timeout 300 python scripts/ab_kernels.py 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/dual   /"
done | tee $OUT/ab_mlp.log
