#!/bin/bash
# The drop-in regime (a host-bound loop) with and without this rank's host threads bound to the GPU's NUMA node, inside one box.
export TMPDIR=/tmp
for rep in $(seq 1 ${REPS:-4}); do
  for extra in "WISP_NUMA_BIND=1" "WISP_NUMA_BIND=0"; do
  echo "== $extra (rep $rep)"
  env $extra timeout 300 python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-configs 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); d=j['dropin_regime']; print('  headline %.4f ms  2^18 %.4f ms (%.4f)  drop-in %.3f ms  %s' % (j['ms_per_step'], j['reference_regime']['ms_per_step'], j['reference_regime']['ms_per_step_without_its_prunes'], d['ms_per_step'], j['host_binding']))"
  done
done
