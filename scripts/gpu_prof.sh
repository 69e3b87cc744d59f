#!/bin/bash
# rocprofv3 kernel-trace summary of the bench command (only the small stats CSVs are copied back)
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
REPO="$PWD"
ARGS="${1:---steps 10 --warmup 2 --no-cpu-baseline}"
(cd /tmp && rm -rf /tmp/prof && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python "$REPO/bench.py" $ARGS > "$REPO/gpurun_out/prof.log" 2>&1)
find /tmp/prof -name "*stats*.csv" -exec cp {} gpurun_out/prof/ \;
tail -2 gpurun_out/prof.log | cut -c1-600
ls -la gpurun_out/prof; head -40 gpurun_out/prof/*kernel_stats.csv
