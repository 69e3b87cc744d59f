#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r4f; mkdir -p $OUT
REPO="$PWD"
(cd /tmp && rm -rf /tmp/prof_ng && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ng -o p -- python "$REPO/bench.py" --config nglod --steps 200 --pretrain 100 > "$REPO/$OUT/prof_nglod.log" 2>&1)
find /tmp/prof_ng -name "*kernel_stats.csv" -exec cp {} $OUT/r04_nglod_kernel_stats.csv \;
python - <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/r4f/r04_nglod_kernel_stats.csv')))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
for r in rows[:16]:
    print(r['Name'][:70].ljust(70), r['Calls'].rjust(6), f"{float(r['AverageNs'])/1e3:9.1f} us", r['Percentage'])
PY
