#!/bin/bash
# A/B: normal library vs debug variant (emit kernel without record writes) - kernel timings only
export TMPDIR=/tmp
REPO="$PWD"
run() {
  (cd /tmp && rm -rf /tmp/prof && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python "$REPO/bench.py" --steps 8 --warmup 2 --no-cpu-baseline > /tmp/prof.log 2>&1)
  python - <<PY
import csv,glob
f=glob.glob("/tmp/prof/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "hashgrid_bwd" in r["Name"]: print("$1", r["Calls"], round(float(r["AverageNs"])/1e3,1), "us", r["Name"][:45])
PY
}
run normal
cp kaolin-wisp_amd/csrc/libwisp_hip.so /tmp/keep.so
cp kaolin-wisp_amd/csrc/libwisp_hip_nowrite.so kaolin-wisp_amd/csrc/libwisp_hip.so
run nowrite
cp /tmp/keep.so kaolin-wisp_amd/csrc/libwisp_hip.so
