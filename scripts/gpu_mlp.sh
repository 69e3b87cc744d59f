#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "mlp or decoder or nerf or pipeline or step or trainer or flagship" > gpurun_out/pytest_mlp.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_mlp.log
tail -4 gpurun_out/pytest_mlp.log
timeout 900 python bench.py --no-cpu-baseline --no-pmc 2>&1 | grep -v amdgpu.ids > gpurun_out/bench_q.log; tail -1 gpurun_out/bench_q.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['psnr_db'], d['reference_regime']['ms_per_step']); print({k:(round(v['avg_ms'],4), round(v['frac'],3)) for k,v in d['roofline']['all_kernels'].items()})"
