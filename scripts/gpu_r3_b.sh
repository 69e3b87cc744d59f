#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3b
OUT=gpurun_out/r3b
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "loss_scale or non_finite or dropin or cells" > $OUT/pytest_new.log 2>&1
echo "pytest exit: $?" >> $OUT/pytest_new.log; tail -30 $OUT/pytest_new.log | cut -c1-600
timeout 600 python scripts/prof_dropin.py 2>&1 | grep -v amdgpu.ids > $OUT/prof_dropin.log; head -90 $OUT/prof_dropin.log | cut -c1-220
