#!/bin/bash
# A/B inside one box: per-ray view codes for the narrow fp32 rows of the VQAD field (default) vs per-sample directions (WISP_MLP_RAYS=wide)
export TMPDIR=/tmp
for rep in 1 2; do for mode in wide default; do
  WISP_MLP_RAYS=$mode timeout 600 python bench.py --config vqad --steps 100 --pretrain 100 2>&1 | grep -v amdgpu.ids | tail -1 > /tmp/v.json
  python - $mode <<'PY'
import json, sys
j=json.loads(open('/tmp/v.json').read())
k=j['kernels']
print(sys.argv[1].ljust(8), 'ms/step %.4f' % j['ms_per_step'], {n: round(v['avg_ms'],4) for n,v in k.items() if 'mlp' in n or 'codebook_trilinear' in n})
PY
done; done
