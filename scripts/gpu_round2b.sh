#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "spc_build or prune or end_to_end or octree_as or raytrace or query or direct_step or checkpoint or forced" > gpurun_out/pytest_sub.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_sub.log; tail -15 gpurun_out/pytest_sub.log
timeout 300 python scripts/time_prune.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/time_prune.log
timeout 600 python bench.py --no-pmc --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bench_b.log | cut -c1-1800
