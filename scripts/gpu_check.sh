#!/bin/bash
# Runs on the GPU box (via gpurun): the whole parity suite, smoke, the default bench line (live PMC traffic + CPU baseline),
# an fp32 bench line, and a rocprofv3 kernel trace of the default command.  Everything is logged under gpurun_out/.
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
REPO="$PWD"
rocm-smi --showproductname 2>/dev/null | head -5 > gpurun_out/device.txt
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_gpu.log
tail -6 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids > gpurun_out/smoke.log; echo "smoke exit $?" >> gpurun_out/smoke.log; tail -3 gpurun_out/smoke.log
timeout 900 python bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/bench_bf16.log; tail -1 gpurun_out/bench_bf16.log | cut -c1-1500
timeout 600 python bench.py --precision fp32 --no-cpu-baseline --no-pmc --steps 40 2>&1 | grep -v amdgpu.ids > gpurun_out/bench_fp32.log; tail -1 gpurun_out/bench_fp32.log | cut -c1-600
timeout 600 python bench.py --hidden 128 --no-cpu-baseline --no-pmc --steps 60 2>&1 | grep -v amdgpu.ids > gpurun_out/bench_h128.log; tail -1 gpurun_out/bench_h128.log | cut -c1-600
(cd /tmp && rm -rf /tmp/prof && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python "$REPO/bench.py" --steps 60 --pretrain 300 --eval-rays 0 --no-cpu-baseline --no-pmc > "$REPO/gpurun_out/prof.log" 2>&1)
find /tmp/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/prof/r02_bench_default_kernel_stats.csv \;
python scripts/trace_gaps.py /tmp/prof > gpurun_out/prof/r02_step_timeline.txt 2>&1; cat gpurun_out/prof/r02_step_timeline.txt | head -24
python scripts/trace_gaps.py /tmp/prof hashgrid_fwd 100 > gpurun_out/prof/r02_step_timeline_2p21.txt 2>&1; cat gpurun_out/prof/r02_step_timeline_2p21.txt | head -30
head -16 gpurun_out/prof/r02_bench_default_kernel_stats.csv | cut -c1-200
