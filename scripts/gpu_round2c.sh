#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "spc_build or prune or sdf or optim or reference_named" > gpurun_out/pytest_sub.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_sub.log; tail -8 gpurun_out/pytest_sub.log
STEPS=50 PRETRAIN=100 bash scripts/gpu_configs.sh
