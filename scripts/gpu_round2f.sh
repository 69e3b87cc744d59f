#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/configs gpurun_out/prof
REPO="$PWD"
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "sdf" > gpurun_out/pytest_sub.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_sub.log; tail -4 gpurun_out/pytest_sub.log
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
timeout 300 python scripts/psnr_parity.py --backend hip --out gpurun_out/r02_psnr_parity_hip.log > /dev/null 2>&1
timeout 300 python scripts/psnr_parity.py --backend hip --perturb 1e-6 --out gpurun_out/r02_psnr_parity_hip_perturbed.log > /dev/null 2>&1
tail -2 gpurun_out/r02_psnr_parity_hip.log gpurun_out/r02_psnr_parity_hip_perturbed.log
(cd /tmp && rm -rf /tmp/prof18 && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof18 -o p -- python "$REPO/bench.py" --target-samples 262144 --pretrain 300 --steps 40 --eval-rays 0 --no-pmc --no-cpu-baseline > "$REPO/gpurun_out/prof18.log" 2>&1)
python scripts/trace_gaps.py /tmp/prof18 | tee gpurun_out/prof/r02_step_timeline_2p18.txt
