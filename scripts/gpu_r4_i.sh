#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r4i; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_0_parity.py tests/test_gpu_1_selfcheck.py -m gpu -q --tb=short -p no:cacheprovider -x \
  -k "decoder or direct_step or octree or codebook or triplanar or hidden" > $OUT/pytest_new.log 2>&1
echo "tests exit $?: $(tail -1 $OUT/pytest_new.log)"
grep -E "^(FAILED|ERROR)|^E " $OUT/pytest_new.log | head -20
for cfg in vqad v8; do
timeout 600 python bench.py --config $cfg --steps 100 --pretrain 200 2>&1 | grep -v amdgpu.ids | tail -1 > $OUT/bench_$cfg.json
python - $cfg <<'PY'
import json, sys
j=json.loads(open(f'gpurun_out/r4i/bench_{sys.argv[1]}.json').read())
print(sys.argv[1], 'ms/step', j['ms_per_step'], 'value', j['value'], 'busy', j.get('gpu_busy_fraction'))
print({k: round(v['avg_ms'],4) for k,v in list(j['kernels'].items())[:9]})
PY
done
