#!/bin/bash
# A/B of environment switches on bench.py's headline step (the reference batch), alternating runs inside ONE GPU call.
# usage: scripts/ab_bench_env.sh OUT.txt ROUNDS "ENV_A" "ENV_B" ...      (an ENV is "VAR=1 VAR2=x" or "-" for none)
OUT=$1; ROUNDS=$2; shift 2
: > $OUT
for r in $(seq 1 $ROUNDS); do
  for e in "$@"; do
    envs=""; [ "$e" != "-" ] && envs="$e"
    line=$(env $envs python bench.py --steps 200 --no-configs --no-cpu-baseline --no-pmc --quality-budget 0 --dp-steps 0 --dropin-steps 0 ${AB_BENCH_ARGS:---large-target-samples 0} 2>/dev/null | tail -1)
    python - "$e" "$r" "$line" >> $OUT <<'PY'
import json, sys
e, r, line = sys.argv[1:4]
d = json.loads(line)
k = d["roofline"]["all_kernels"]
lg = d.get("large_batch_regime")
print(f"[{e}] round {r}: ms/step {d['ms_per_step']:.4f} (window {d['timed_window']['ms_per_step']:.4f})  R {d['config']['rays_per_step_per_gpu']}  "
      + "  ".join(f"{n} {v['avg_ms']*1e3:.1f}us" for n, v in k.items())
      + (f"  | large {lg['ms_per_step']:.4f} " + "  ".join(f"{n} {v['avg_ms']*1e3:.1f}us" for n, v in lg['roofline']['all_kernels'].items()) if lg else ""))
PY
  done
done
cat $OUT
