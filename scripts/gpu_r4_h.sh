#!/bin/bash
# kernel timeline of the graph-replayed NGLOD step
export TMPDIR=/tmp
REPO="$PWD"; mkdir -p gpurun_out/r4h
(cd /tmp && rm -rf /tmp/prof_ng && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_ng -o p -- python "$REPO/bench.py" --config nglod --steps 200 --pretrain 100 > /dev/null 2>&1)
python scripts/trace_gaps.py /tmp/prof_ng sdf_train_kernel | tee gpurun_out/r4h/r04_nglod_step_timeline.txt | head -40
