#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3j
OUT=gpurun_out/r3j
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "small_decoder or sdf or nglod or neural_sdf" > $OUT/pytest.log 2>&1
echo "pytest exit: $?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log | cut -c1-300
timeout 600 python bench.py --config nglod --steps 200 --pretrain 40 --warmup 5 2>&1 | grep -v amdgpu.ids | tail -1 > $OUT/nglod.json
python - <<'PY'
import json
d = json.load(open('gpurun_out/r3j/nglod.json'))
print(d['value'], d['ms_per_step'], d['eager'])
for k, v in list(d['kernels'].items())[:8]:
    print(k, round(v['avg_ms'], 4), v['launches'], round(v['share'], 3))
PY
