#!/bin/bash
# kernel trace of the default bench command (stats + step timelines), as in gpu_r3_final.sh, on its own
export TMPDIR=/tmp
TAG=${1:-r03}
mkdir -p gpurun_out/$TAG
OUT="$PWD/gpurun_out/$TAG"
REPO="$PWD"
(cd /tmp && rm -rf /tmp/prof && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python "$REPO/bench.py" --steps 60 --pretrain 300 --eval-rays 0 --no-cpu-baseline --no-pmc --no-configs --dropin-steps 0 > "$OUT/prof.log" 2>&1)
F=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
if [ -n "$F" ]; then head -41 "$F" > "$OUT/${TAG}_bench_default_kernel_stats.csv"; fi
timeout 120 python scripts/trace_gaps.py /tmp/prof hashgrid_fwd 100 > $OUT/${TAG}_step_timeline_2p21.txt 2>&1 < /dev/null
timeout 120 python scripts/trace_gaps.py /tmp/prof > $OUT/${TAG}_step_timeline_2p18.txt 2>&1 < /dev/null
head -24 $OUT/${TAG}_step_timeline_2p21.txt
