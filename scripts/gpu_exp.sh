#!/bin/bash
export TMPDIR=/tmp
REPO="$PWD"
prof() {
  (cd /tmp && rm -rf /tmp/prof && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o e -- python "$REPO/scripts/exp_bwd.py" > /tmp/exp.log 2>&1)
  tail -1 /tmp/exp.log
  python - <<PY
import csv,glob
f=glob.glob("/tmp/prof/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "hashgrid_bwd" in r["Name"]: print("   $1", r["Calls"], round(float(r["AverageNs"])/1e3,1), "us", r["Name"][:40])
PY
}
prof fix64_s8
WISP_RD_SPLITS=64 prof fix64_s64
WISP_RD_SPLITS=1 prof fix64_s1
WISP_RD_F32=1 WISP_RD_SPLITS=64 prof f32_s64
