#!/bin/bash
# Round 5, second GPU call: the pipelined hidden-128 dW kernel against the old one (time + bitwise gradients), the decoder
# tests, the grad_coords tests, the composite-loss workgroup shapes, bench lines of both dW kernels.
export TMPDIR=/tmp
OUT=gpurun_out/r5b; mkdir -p $OUT
REPO="$PWD"
WISP_WIDE_DW=1 timeout 300 python scripts/bench_wide_dw.py /tmp/dw_old.pt 2>&1 | grep -v amdgpu.ids | tee $OUT/wide_dw_old.txt
timeout 300 python scripts/bench_wide_dw.py /tmp/dw_new.pt 2>&1 | grep -v amdgpu.ids | tee $OUT/wide_dw_new.txt
python - <<'PY' 2>&1 | tee $OUT/wide_dw_compare.txt
import torch
a, b = torch.load('/tmp/dw_old.pt'), torch.load('/tmp/dw_new.pt')
for S in a:
    gp0, gf0 = a[S]; gp1, gf1 = b[S]
    print(S, 'grad_params bitwise equal:', torch.equal(gp0, gp1), ' max|diff|', float((gp0 - gp1).abs().max()), ' scale', float(gp0.abs().max()),
          '| grad_feats equal:', torch.equal(gf0, gf1))
PY
timeout 900 python -m pytest tests/test_gpu_0_parity.py -x -q --tb=short -p no:cacheprovider -k "wide or decoder or grad_coords or prune or composite" > $OUT/pytest_sel.log 2>&1
echo "pytest (selection) exit $?: $(tail -1 $OUT/pytest_sel.log)"
timeout 600 python -m pytest tests/test_gpu_1_selfcheck.py -x -q --tb=short -p no:cacheprovider -k "128 or composite or direct" > $OUT/pytest_sel2.log 2>&1
echo "pytest (selfcheck selection) exit $?: $(tail -1 $OUT/pytest_sel2.log)"
for w in 1 2 4; do
  WISP_COMPOSITE_WAVES=$w timeout 600 python bench.py --steps 100 --no-cpu-baseline --no-pmc --no-configs --dropin-steps 0 --eval-rays 0 2>&1 | grep -v amdgpu.ids | tail -1 > $OUT/bench_cw$w.json
  python - $w $OUT/bench_cw$w.json <<'PY'
import json, sys
j = json.load(open(sys.argv[2]))
print('composite waves', sys.argv[1], 'ms/step %.4f' % j['timed_window']['ms_per_step'], 'ref-regime %.4f (no prunes %.4f)' % (j['reference_regime']['ms_per_step'], j['reference_regime']['ms_per_step_without_its_prunes']))
PY
done
for v in 1 0; do
  WISP_WIDE_DW=$v timeout 600 python bench.py --hidden 128 --steps 60 --no-cpu-baseline --no-pmc --no-configs --dropin-steps 0 2>&1 | grep -v amdgpu.ids | tail -1 > $OUT/bench_h128_dw$v.json
  python - $v $OUT/bench_h128_dw$v.json <<'PY'
import json, sys
j = json.load(open(sys.argv[2]))
print('hidden128 WISP_WIDE_DW=' + sys.argv[1], 'ms/step %.4f' % j['timed_window']['ms_per_step'], 'psnr %.2f' % j['psnr_db'],
      {k: round(v['avg_ms'], 4) for k, v in j['roofline']['all_kernels'].items()})
PY
done
