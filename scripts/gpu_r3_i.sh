#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3i
OUT=gpurun_out/r3i
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "rgb_loss or direct_step or trainer or prefetched" > $OUT/pytest.log 2>&1
echo "pytest exit: $?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log | cut -c1-300
(cd /tmp && rm -rf /tmp/prof && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 60 --pretrain 300 --eval-rays 0 --no-cpu-baseline --no-pmc --no-configs --dropin-steps 0 > "$GRAFT_REPO_ROOT/$OUT/prof.log" 2>&1)
python scripts/trace_gaps.py /tmp/prof hashgrid_fwd 100 2>&1 | head -16
python scripts/trace_gaps.py /tmp/prof 2>&1 | head -16
