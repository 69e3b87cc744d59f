#!/bin/bash
# A/B inside one box: the reduce kernel's flush in 8-byte pairs (every level of nerf_hash.yaml from the fourth on starts 8 bytes off a
# 16-byte boundary: the 16-byte path never ran there) and the table's AdamW step folded into that flush.
#   base  = library without the pair path (ab/noflushpairs.so), separate optimizer pass
#   pairs = product library, separate optimizer pass (WISP_ADAM_IN_FLUSH=0)
#   fused = product library, default
export TMPDIR=/tmp
mkdir -p gpurun_out
if [ "${TESTS:-1}" = 1 ]; then
timeout 900 python -m pytest tests -m gpu -q -k "folded or off_a_16_byte or adamw or hashgrid_backward or direct_step_equals_modular_step or flagship or trainer_step or psnr" 2>&1 | tail -5
fi
run() { # label, env...
  local label=$1; shift
  env "$@" timeout 600 python bench.py --steps ${STEPS:-200} --no-pmc --no-configs --no-cpu-baseline --dropin-steps 0 2>&1 | grep -v amdgpu.ids | tail -1 > /tmp/b.json
  python - $label <<'PY'
import json, sys
j = json.loads(open('/tmp/b.json').read())
r = j['roofline']
k = r['all_kernels']
print(sys.argv[1].ljust(6), 'ms/step %.4f' % j['ms_per_step'], 'ref-regime %.4f (no prunes %.4f)' % (j['reference_regime']['ms_per_step'], j['reference_regime']['ms_per_step_without_its_prunes']),
      'psnr %.2f' % j['psnr_db'], {n: round(v['avg_ms'], 4) for n, v in k.items()}, 'bwd frac', round(k['hashgrid_bwd']['frac'], 3),
      k['hashgrid_bwd'].get('fused_optimizer', {}).get('frac_on_the_backward_bytes_alone'))
PY
}
for rep in $(seq 1 ${REPS:-2}); do
  run base WISP_HIP_LIB=$PWD/kaolin-wisp_amd/csrc/ab/noflushpairs.so WISP_ADAM_IN_FLUSH=0
  run pairs WISP_ADAM_IN_FLUSH=0
  run fused WISP_ADAM_IN_FLUSH=1
done
