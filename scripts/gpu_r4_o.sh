#!/bin/bash
export TMPDIR=/tmp
# the reduce kernel with its LDS atomics compiled out (build_ab.sh noatom "-DHG_EXP_NOATOM" hashgrid.hip) and with interleaved accumulators
# (build_ab.sh accinterleaved "-DHG_ACC_PLANES=0" hashgrid.hip) against the product library: whole call and per-kernel averages
for lib in base noatom.so accinterleaved.so; do
  if [ "$lib" = base ]; then unset WISP_HIP_LIB; else export WISP_HIP_LIB=$PWD/kaolin-wisp_amd/csrc/ab/$lib; fi
  echo "== $lib"; timeout 200 python scripts/bench_hashbwd.py 2>&1 | grep -v amdgpu.ids | tail -1
  (cd /tmp && rm -rf /tmp/p_$lib && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$lib -o p -- python $GRAFT_REPO_ROOT/scripts/bench_hashbwd.py > /dev/null 2>&1; f=$(find /tmp/p_$lib -name "*kernel_stats.csv" | head -1); python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'hashgrid_bwd' in r['Name']: print('   ', r['Name'].split('(')[0][-60:], r['Calls'], 'avg us %.1f' % (float(r['AverageNs']) / 1e3))
PY
)
done
