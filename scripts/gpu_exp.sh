#!/bin/bash
export TMPDIR=/tmp
REPO="$PWD"
prof() {
  (cd /tmp && rm -rf /tmp/prof && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o e -- python "$REPO/scripts/exp_bwd.py" > /tmp/exp.log 2>&1)
  tail -1 /tmp/exp.log | cut -c1-100
  python - <<PY
import csv,glob
f=glob.glob("/tmp/prof/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "hashgrid_bwd_emit" in r["Name"]: print("   $1", r["Calls"], round(float(r["AverageNs"])/1e3,1), "us", r["Name"][:40])
PY
}
cp kaolin-wisp_amd/csrc/libwisp_hip.so /tmp/keep.so
prof t256g2
for v in t512g2 t1024g2 t512g1 t256g4; do cp kaolin-wisp_amd/csrc/libwisp_hip_$v.so kaolin-wisp_amd/csrc/libwisp_hip.so; prof $v; done
cp /tmp/keep.so kaolin-wisp_amd/csrc/libwisp_hip.so
