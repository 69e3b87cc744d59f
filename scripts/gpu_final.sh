#!/bin/bash
# Short end-of-round check for a small GPU budget: the whole parity suite, smoke, one quick bench line (no CPU baseline, no PMC).
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 280 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider > gpurun_out/final_pytest_gpu.log 2>&1
echo "pytest exit: $?" >> gpurun_out/final_pytest_gpu.log
tail -4 gpurun_out/final_pytest_gpu.log
timeout 60 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids > gpurun_out/final_smoke.log; echo "smoke exit ${PIPESTATUS[0]}" >> gpurun_out/final_smoke.log; tail -2 gpurun_out/final_smoke.log
timeout 100 python bench.py --no-cpu-baseline --no-pmc --steps 60 2>&1 | grep -v amdgpu.ids > gpurun_out/final_bench.log; tail -1 gpurun_out/final_bench.log | cut -c1-900
