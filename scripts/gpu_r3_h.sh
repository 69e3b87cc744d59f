#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3h
OUT=gpurun_out/r3h
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "codebook or vqad or every_march" > $OUT/pytest.log 2>&1
echo "pytest exit: $?" >> $OUT/pytest.log; tail -4 $OUT/pytest.log | cut -c1-300
timeout 600 python bench.py --config vqad --steps 30 --pretrain 30 --warmup 3 2>&1 | grep -v amdgpu.ids | tail -1 > $OUT/vqad.json
python - <<'PY'
import json
d = json.load(open('gpurun_out/r3h/vqad.json'))
print(d['value'], d['ms_per_step'], d['gpu_busy_fraction'])
for k, v in list(d['kernels'].items())[:8]:
    print(k, round(v['avg_ms'], 4), v['launches'], round(v['share'], 3))
PY
