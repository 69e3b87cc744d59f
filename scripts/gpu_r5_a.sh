#!/bin/bash
# Round 5, first GPU call: the GPU suite on the round's CPU-side changes, the default bench line, per-regime kernel stats.
export TMPDIR=/tmp
OUT=gpurun_out/r5a; mkdir -p $OUT
REPO="$PWD"
rocm-smi --showproductname 2>/dev/null | head -5 > $OUT/device.txt
WISP_TEST_MARGINS=$REPO/$OUT/margins.jsonl timeout 1200 python -m pytest tests -m gpu -x -q --tb=short -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest exit $?: $(tail -1 $OUT/pytest.log)"
timeout 300 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids | tail -2
timeout 1500 python bench.py 2>&1 | grep -v amdgpu.ids > $OUT/bench_default.log; tail -1 $OUT/bench_default.log | cut -c1-700
bash scripts/regime_stats.sh $OUT r05a
