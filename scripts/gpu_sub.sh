#!/bin/bash
# usage: gpu_sub.sh "<pytest -k expression>"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "$1" > gpurun_out/pytest_sub.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_sub.log; tail -25 gpurun_out/pytest_sub.log | cut -c1-300
