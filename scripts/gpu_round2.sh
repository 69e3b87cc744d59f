#!/bin/bash
# GPU box: full parity suite, flagship PSNR-parity run (HIP side), default bench with live PMC.
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_gpu.log
grep -E "passed|failed|FAILED|ERROR|pytest exit|worst \|grad|DP_RESULT|'world'" gpurun_out/pytest_gpu.log | tail -40
timeout 600 python scripts/psnr_parity.py --backend hip --out gpurun_out/r02_psnr_parity_hip.log > gpurun_out/psnr_hip.out 2>&1; tail -12 gpurun_out/r02_psnr_parity_hip.log
timeout 900 python bench.py > gpurun_out/bench_default.log 2>&1; echo "bench exit $?"; tail -3 gpurun_out/bench_default.log
