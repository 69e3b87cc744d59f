#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "raytrace or codebook or voxel or octree_radiance or tracer_end or sdf or octree_as" > gpurun_out/pytest_sub.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_sub.log; tail -8 gpurun_out/pytest_sub.log
STEPS=50 PRETRAIN=100 bash scripts/gpu_configs.sh 2>&1 | python -c "
import sys, json
for line in sys.stdin:
    i = line.find('\"kernels\"')
    print(line[:300] if i < 0 else line[max(0, i - 200):i + 900])
"
