#!/bin/bash
# Runs on the GPU box (via gpurun): parity tests, smoke, a short bench in both precisions, and a rocprofv3 kernel trace.
# Everything is logged under gpurun_out/.
export TMPDIR=/tmp
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
rocm-smi --showproductname 2>/dev/null | head -5 > gpurun_out/device.txt
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_gpu.log
tail -40 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log; tail -3 gpurun_out/smoke.log
timeout 600 python bench.py --steps 20 --warmup 3 --precision fp32 --no-cpu-baseline > gpurun_out/bench_fp32.log 2>&1; tail -2 gpurun_out/bench_fp32.log
timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_bf16.log 2>&1; tail -2 gpurun_out/bench_bf16.log
REPO="$PWD"
(cd /tmp && rm -rf /tmp/prof && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python "$REPO/bench.py" --steps 10 --warmup 2 --no-cpu-baseline > "$REPO/gpurun_out/prof.log" 2>&1)
mkdir -p gpurun_out/prof && find /tmp/prof -name "*stats*.csv" -exec cp {} gpurun_out/prof/ \;
ls -la gpurun_out/prof; head -30 gpurun_out/prof/*kernel_stats.csv
