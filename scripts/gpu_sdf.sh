#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "sdf" > gpurun_out/pytest_sdf.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_sdf.log
tail -25 gpurun_out/pytest_sdf.log | cut -c1-300
timeout 600 python bench.py --config nglod 2>&1 | grep -v amdgpu.ids > gpurun_out/bench_nglod.log; tail -1 gpurun_out/bench_nglod.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('eager'), d['render']['ms'], d['render']['rays_per_sec'], d['mean_abs_sdf_error'])"
