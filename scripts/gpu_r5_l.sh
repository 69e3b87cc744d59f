#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r5l; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_0_parity.py tests/test_gpu_1_selfcheck.py -x -q --tb=short -p no:cacheprovider -k "composite or direct or tracer_end_to_end" > $OUT/pytest_sel.log 2>&1
echo "pytest (selection) exit $?: $(tail -1 $OUT/pytest_sel.log)"
bash scripts/regime_stats.sh $OUT r05l --dropin-steps 0 2>&1 | grep -E "composite|headline|reference_regime|wall" 
