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
cp kaolin-wisp_amd/csrc/libwisp_hip.so /tmp/keep.so
prof w4
for w in 5 6 8; do cp kaolin-wisp_amd/csrc/libwisp_hip_w$w.so kaolin-wisp_amd/csrc/libwisp_hip.so; prof w$w; done
cp /tmp/keep.so kaolin-wisp_amd/csrc/libwisp_hip.so
