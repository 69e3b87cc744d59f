#!/bin/bash
# Secondary configs (C3/C4/C5 stand-ins): one bench line each + a rocprofv3 kernel-trace summary each.
export TMPDIR=/tmp
mkdir -p gpurun_out/configs
REPO="$PWD"
for cfg in v8 vqad nglod; do
  timeout 600 python bench.py --config $cfg --steps ${STEPS:-100} --pretrain ${PRETRAIN:-200} 2>&1 | grep -v amdgpu.ids > gpurun_out/configs/bench_$cfg.log
  tail -c 1500 gpurun_out/configs/bench_$cfg.log; echo
  if [ -n "$PROF" ]; then
    (cd /tmp && rm -rf /tmp/prof_$cfg && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$cfg -o p -- python "$REPO/bench.py" --config $cfg --steps 30 --pretrain 50 > "$REPO/gpurun_out/configs/prof_$cfg.log" 2>&1)
    find /tmp/prof_$cfg -name "*kernel_stats.csv" -exec cp {} gpurun_out/configs/r02_${cfg}_kernel_stats.csv \;
    head -8 gpurun_out/configs/r02_${cfg}_kernel_stats.csv | cut -c1-160
  fi
done
