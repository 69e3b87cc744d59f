#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r4g; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_0_parity.py tests/test_gpu_1_selfcheck.py -m gpu -q --tb=short -p no:cacheprovider -x \
  -k "tracer or dropin or flagship or validation or prune or psnr or direct_step" > $OUT/pytest_new.log 2>&1
echo "tests exit $?: $(tail -1 $OUT/pytest_new.log)"
grep -E "^(FAILED|ERROR)|^E " $OUT/pytest_new.log | head -20
timeout 400 python scripts/prof_dropin2.py 2>&1 | grep -v amdgpu.ids | tee $OUT/prof_dropin2.txt | tail -16
WISP_FUSED_TRACE=0 timeout 400 python scripts/prof_dropin2.py 2>&1 | grep -v amdgpu.ids | grep "^iterate" | head -3
DENSE=1 timeout 400 python scripts/prof_dropin3.py 2>&1 | grep -v amdgpu.ids | sed -n 5,12p
