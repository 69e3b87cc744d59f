#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "wide or hidden_128" > gpurun_out/pytest_sub.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_sub.log; tail -5 gpurun_out/pytest_sub.log | cut -c1-300
bash scripts/gpu_wide2.sh
