#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
for rep in 1 2; do
for q in 0 2 3 4; do WISP_HG_EMIT_WGS_PER_CU=$q timeout 300 python scripts/ab_kernels.py 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/per_cu=$q /"; done
done | tee gpurun_out/ab_hg2.log
