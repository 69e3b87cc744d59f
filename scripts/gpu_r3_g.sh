#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3g
OUT=gpurun_out/r3g
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -m gpu -q --tb=short -p no:cacheprovider -x -k "direct_step or trainer or flagship or psnr or composite or multi or forced" > $OUT/pytest.log 2>&1
echo "pytest exit: $?" >> $OUT/pytest.log; tail -4 $OUT/pytest.log | cut -c1-300
for rep in 1 2; do
for v in 0 1; do
WISP_AHEAD_STREAM=$v timeout 600 python bench.py --steps 200 --no-cpu-baseline --no-pmc --no-configs --dropin-steps 0 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('ahead=$v', round(d['ms_per_step'], 4), 'ref', round(d['reference_regime']['ms_per_step'], 4), 'psnr', round(d['psnr_db'], 2), 'loss_k', round(d['roofline']['all_kernels']['hashgrid_bwd']['avg_ms'], 4))
"
done
done | tee $OUT/ab_ahead.log
