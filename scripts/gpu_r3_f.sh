#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3f
OUT=gpurun_out/r3f
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "hashgrid" > $OUT/pytest_hg.log 2>&1
echo "pytest exit: $?" >> $OUT/pytest_hg.log; tail -4 $OUT/pytest_hg.log | cut -c1-300
for rep in 1 2; do
  WISP_HG_EMIT_PIECE=1024 timeout 300 python scripts/ab_kernels.py 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/piece1024 /"
  timeout 300 python scripts/ab_kernels.py 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/auto      /"
done | tee $OUT/ab.log
