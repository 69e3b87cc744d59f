#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3c
OUT=gpurun_out/r3c
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit: $?" >> $OUT/pytest_gpu.log; tail -25 $OUT/pytest_gpu.log | cut -c1-400
timeout 900 python bench.py --steps 60 2>&1 | grep -v amdgpu.ids > $OUT/bench.log; tail -1 $OUT/bench.log | cut -c1-200
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r3c/bench.log').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value', 'ms_per_step')}, d['reference_regime']['ms_per_step'], d['dropin_regime'])
for k, v in d.get('configs', {}).items():
    print(k, json.dumps(v)[:900])
PY
WISP_FORCE_ALLREDUCE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 40 --no-cpu-baseline --no-pmc --no-configs --dropin-steps 0 2>&1 | grep -v amdgpu.ids > $OUT/bench_forced.log; tail -1 $OUT/bench_forced.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('forced rank:', d['ms_per_step'], d['comm'])"
WISP_FORCE_ALLREDUCE=1 WISP_SHARDED_OPTIM=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 40 --no-cpu-baseline --no-pmc --no-configs --dropin-steps 0 2>&1 | grep -v amdgpu.ids > $OUT/bench_forced_sharded.log; tail -1 $OUT/bench_forced_sharded.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('forced sharded:', d['ms_per_step'], d['comm'])"
