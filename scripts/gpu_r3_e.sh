#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3e
OUT=gpurun_out/r3e
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -s -k "hashgrid or direct_step or flagship" > $OUT/pytest_hg.log 2>&1
echo "pytest exit: $?" >> $OUT/pytest_hg.log; grep -E "scratch:|passed|failed|Error|assert" $OUT/pytest_hg.log | tail -12 | cut -c1-400
for rep in 1 2; do
  WISP_HG_SLOT_FIT=0 timeout 300 python scripts/ab_kernels.py 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/nofit /"
  WISP_HG_SLOT_FIT=1 timeout 300 python scripts/ab_kernels.py 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/fit   /"
done | tee $OUT/ab_fit.log
timeout 900 python bench.py --steps 100 --no-configs --no-cpu-baseline 2>&1 | grep -v amdgpu.ids > $OUT/bench.log; tail -1 $OUT/bench.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d[k] for k in ('value', 'ms_per_step')}, 'ref', d['reference_regime']['ms_per_step'], 'dropin', d['dropin_regime']['ms_per_step'])
print({k: (v['avg_ms'], round(v['frac'], 3)) for k, v in d['roofline']['all_kernels'].items()}, d['roofline'].get('traffic'))
"
