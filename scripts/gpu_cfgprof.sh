#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/configs
PROF=1 STEPS=100 PRETRAIN=200 bash scripts/gpu_configs.sh > gpurun_out/configs/run.log 2>&1
tail -5 gpurun_out/configs/run.log | cut -c1-300
ls gpurun_out/configs
