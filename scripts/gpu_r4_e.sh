#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r4e; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_0_parity.py tests/test_gpu_1_selfcheck.py -m gpu -q --tb=short -p no:cacheprovider -x \
  -k "sdf or nglod or scratch_follows or hashgrid_backward" > $OUT/pytest_new.log 2>&1
echo "tests exit $?: $(tail -1 $OUT/pytest_new.log)"
grep -E "^(FAILED|ERROR)|^E " $OUT/pytest_new.log | head -20
timeout 600 python bench.py --config nglod --steps 300 --pretrain 200 2>&1 | grep -v amdgpu.ids > $OUT/bench_nglod.log
python - <<'PY'
import json
s=open('gpurun_out/r4e/bench_nglod.log').read()
j=json.loads(s[s.index('{"metric"'):].splitlines()[0])
print('ms/step', j['ms_per_step'], 'eager', j['eager']['ms_per_step'], 'err', j['mean_abs_sdf_error'], 'loss', j['final_loss'])
print({k: round(v['avg_ms'],4) for k,v in j['kernels'].items()})
print(j['roofline'])
PY
