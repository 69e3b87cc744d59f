#!/bin/bash
# bench only (both precisions) + optional pytest filter
export TMPDIR=/tmp
mkdir -p gpurun_out
if [ -n "$1" ]; then timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "$1" 2>&1 | tail -15; fi
timeout 600 python bench.py --steps 20 --warmup 3 --precision fp32 --no-cpu-baseline > gpurun_out/bench_fp32.log 2>&1; tail -1 gpurun_out/bench_fp32.log | cut -c1-2500
timeout 900 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_bf16.log 2>&1; tail -1 gpurun_out/bench_bf16.log | cut -c1-2500
timeout 900 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --target-samples 262144 > gpurun_out/bench_bf16_2e18.log 2>&1; tail -1 gpurun_out/bench_bf16_2e18.log | cut -c1-1200
