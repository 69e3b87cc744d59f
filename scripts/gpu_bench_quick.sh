export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python bench.py --no-cpu-baseline 2>&1 | grep -v amdgpu.ids > gpurun_out/bench_q.log; tail -1 gpurun_out/bench_q.log | cut -c1-1800
