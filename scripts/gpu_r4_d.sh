#!/bin/bash
# round 4: the driver's sequence - GPU suite with -x, smoke, the default bench line
export TMPDIR=/tmp
OUT=gpurun_out/r4d; mkdir -p $OUT
rm -f $OUT/margins.jsonl
WISP_TEST_MARGINS=$PWD/$OUT/margins.jsonl timeout 1500 python -m pytest tests -m gpu -x -q --tb=short -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?: $(tail -1 $OUT/pytest_gpu.log)"
grep -E "^(FAILED|ERROR)|^E " $OUT/pytest_gpu.log | head -20
timeout 300 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids > $OUT/smoke.log; echo "smoke exit $?" >> $OUT/smoke.log; tail -2 $OUT/smoke.log
timeout 1200 python bench.py 2>&1 | grep -v amdgpu.ids > $OUT/bench_default.log; tail -1 $OUT/bench_default.log | cut -c1-1200
