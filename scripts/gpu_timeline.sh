#!/bin/bash
# kernel timeline of the default bench's 2^21 regime (and the 2^18 regime) under rocprofv3 --kernel-trace
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
REPO="$PWD"
(cd /tmp && rm -rf /tmp/prof && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o bench -- python "$REPO/bench.py" --steps 60 --pretrain 300 --eval-rays 0 --no-cpu-baseline --no-pmc > "$REPO/gpurun_out/prof.log" 2>&1)
python scripts/trace_gaps.py /tmp/prof hashgrid_fwd 100 > gpurun_out/prof/r02_step_timeline_2p21.txt 2>&1; cat gpurun_out/prof/r02_step_timeline_2p21.txt | head -70
