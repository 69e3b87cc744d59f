#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "codebook or vqad or octree_radiance" > gpurun_out/pytest_cb.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_cb.log
tail -5 gpurun_out/pytest_cb.log | cut -c1-200
for q in 0 1; do
WISP_CODEBOOK_DECODE_ROWS=$q timeout 600 python bench.py --config vqad --steps 60 --pretrain 100 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('decode_rows=$q', round(d['ms_per_step'],4), '%.4g'%d['value'], {n:round(v['avg_ms'],4) for n,v in k.items() if 'codebook' in n or 'trilinear' in n})"
done
