#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python bench.py --hidden 128 --no-pmc --no-cpu-baseline --steps 60 --pretrain 200 2>&1 | grep -v amdgpu.ids > gpurun_out/bench_h128.log
python - <<'PY'
import json
d=json.loads([x for x in open('gpurun_out/bench_h128.log') if x.startswith('{')][-1])
print('h128', d['value'], d['ms_per_step'], 'ref', d['reference_regime']['value'], d['reference_regime']['ms_per_step'], 'psnr', d['psnr_db'])
for k,v in d['roofline']['all_kernels'].items(): print('  ', k, round(v['avg_ms'],4), round(v['frac'],3))
PY
