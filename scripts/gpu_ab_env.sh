#!/bin/bash
# A/B of one environment switch on the default bench line: gpu_ab_env.sh VAR "v1 v2" [extra bench args]
export TMPDIR=/tmp
VAR=$1; VALS=$2; shift 2
for rep in 1 2; do
for v in $VALS; do
  env $VAR=$v timeout 600 python bench.py --steps 100 --warmup 10 --no-configs --dropin-steps 0 "$@" 2>&1 | grep '^{' > gpurun_out/ab_${VAR}_${v}_${rep}.json
  python - <<PY
import json
r=json.loads(open("gpurun_out/ab_${VAR}_${v}_${rep}.json").read())
print("$VAR=$v rep $rep: ms/step %.4f  ref-regime %.4f  psnr %.2f" % (r["ms_per_step"], r["reference_regime"]["ms_per_step"], r.get("psnr_db_train_rays", 0)))
PY
done
done
