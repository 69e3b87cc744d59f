#!/bin/bash
# Kernel trace of the default bench command (without its CPU baseline / PMC children / secondary configs), cut into one
# kernel-stats table per regime by scripts/regime_stats.py.   usage: scripts/regime_stats.sh OUT_DIR TAG [extra bench args]
export TMPDIR=/tmp
OUT=$1; TAG=$2; shift 2
REPO="$PWD"; mkdir -p $OUT
(cd /tmp && rm -rf /tmp/prof_regime && WISP_BENCH_SENTINELS=1 timeout 900 rocprofv3 --kernel-trace --marker-trace --stats --output-format csv \
   -d /tmp/prof_regime -o bench -- python "$REPO/bench.py" --steps 60 --eval-rays 0 --no-cpu-baseline --no-pmc --no-configs --quality-budget 0 --dp-steps 0 "$@" > "$REPO/$OUT/${TAG}_regime_prof.log" 2>&1)
python scripts/regime_stats.py /tmp/prof_regime $OUT/${TAG} | tee $OUT/${TAG}_regime_summary.txt
# rocprofv3's own --stats table of the whole command (all regimes averaged together: the per-regime tables above are the ones to read)
find /tmp/prof_regime -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_whole_command_kernel_stats.csv \;
find /tmp/prof_regime -name "*marker_api_trace.csv" -exec sh -c 'head -40 "$1" > '"$OUT/${TAG}"'_marker_trace_head.csv; wc -l "$1"' _ {} \;
# one timeline per regime, each cut from that regime's own sentinel bracket
python scripts/trace_gaps.py /tmp/prof_regime hashgrid_fwd 0 headline > $OUT/${TAG}_step_timeline_2p18.txt 2>&1; head -22 $OUT/${TAG}_step_timeline_2p18.txt
python scripts/trace_gaps.py /tmp/prof_regime hashgrid_fwd 0 large_batch_regime > $OUT/${TAG}_step_timeline_2p21.txt 2>&1; head -8 $OUT/${TAG}_step_timeline_2p21.txt
python scripts/trace_gaps.py /tmp/prof_regime hashgrid_fwd 0 dropin_regime > $OUT/${TAG}_step_timeline_dropin.txt 2>&1; head -8 $OUT/${TAG}_step_timeline_dropin.txt
tail -1 $OUT/${TAG}_regime_prof.log | cut -c1-300
