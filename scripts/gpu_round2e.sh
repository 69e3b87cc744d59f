#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/configs
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "sdf or codebook or slot_overflow or nerf_hash_shape or repeatable" > gpurun_out/pytest_sub.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_sub.log; tail -8 gpurun_out/pytest_sub.log
for cfg in vqad nglod; do
  timeout 600 python bench.py --config $cfg --steps 50 --pretrain 100 2>&1 | grep -v amdgpu.ids > gpurun_out/configs/bench_$cfg.log
done
python - <<'PY'
import json
for c in ("vqad", "nglod"):
    l=[x for x in open(f'gpurun_out/configs/bench_{c}.log') if x.startswith('{')]
    if not l: print(c, 'NO LINE', open(f'gpurun_out/configs/bench_{c}.log').read()[-1500:]); continue
    d=json.loads(l[-1])
    print(c, d['value'], d['unit'], 'ms/step', d['ms_per_step'], 'busy', d.get('gpu_busy_fraction'), 'render', d.get('render',{}).get('ms'), d.get('render',{}).get('hit_fraction'))
    for k,v in list(d['kernels'].items())[:7]: print('   ',k, round(v['avg_ms'],4), v['launches'], round(v['share'],3))
    for k,v in list(d.get('render',{}).get('kernels',{}).items())[:5]: print('   render',k, round(v['avg_ms'],4), v['launches'], round(v['share'],3))
PY
timeout 600 python bench.py --no-pmc --no-cpu-baseline 2>&1 | grep -v amdgpu.ids > gpurun_out/bench_c.log
WISP_OVERLAP_OPTIMIZER=1 timeout 600 python bench.py --no-pmc --no-cpu-baseline 2>&1 | grep -v amdgpu.ids > gpurun_out/bench_c_overlap.log
python - <<'PY'
import json
for f in ("bench_c", "bench_c_overlap"):
    d=json.loads([x for x in open(f'gpurun_out/{f}.log') if x.startswith('{')][-1])
    print(f, d['value'], d['ms_per_step'], 'ref', d['reference_regime']['value'], d['reference_regime']['ms_per_step'], 'psnr', d['psnr_db'], {k: round(v['avg_ms'],4) for k,v in d['roofline']['all_kernels'].items()})
PY
