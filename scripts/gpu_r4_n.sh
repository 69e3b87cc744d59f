#!/bin/bash
# A/B inside one box: the reduce kernel's bucket accumulators as one plane per feature (product) vs interleaved by feature
# (ab/accinterleaved.so, -DHG_ACC_PLANES=0: as they were)
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -k "hashgrid or folded or flagship" 2>&1 | tail -2
run() { local label=$1; shift
  env "$@" timeout 600 python bench.py --steps 200 --no-pmc --no-configs --no-cpu-baseline --dropin-steps 0 2>&1 | grep -v amdgpu.ids | tail -1 > /tmp/b.json
  python - $label <<'PY'
import json, sys
j = json.loads(open('/tmp/b.json').read())
k = j['roofline']['all_kernels']
print(sys.argv[1].ljust(12), 'ms/step %.4f' % j['ms_per_step'], 'ref-regime %.4f (no prunes %.4f)' % (j['reference_regime']['ms_per_step'], j['reference_regime']['ms_per_step_without_its_prunes']),
      'psnr %.2f' % j['psnr_db'], {n: round(v['avg_ms'], 4) for n, v in k.items()}, 'frac', round(k['hashgrid_bwd']['frac'], 3))
PY
}
for rep in 1 2 3; do
  run interleaved WISP_HIP_LIB=$PWD/kaolin-wisp_amd/csrc/ab/accinterleaved.so
  run planes X=1
done
