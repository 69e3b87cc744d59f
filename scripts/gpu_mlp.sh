#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_0_parity.py tests/test_gpu_1_selfcheck.py -m gpu -q --tb=short -p no:cacheprovider -x -k "mlp or decoder or nerf or pipeline or step or trainer or flagship" > gpurun_out/pytest_mlp.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_mlp.log
tail -15 gpurun_out/pytest_mlp.log | cut -c1-300
for rep in 1 2; do
for q in 0 2; do WISP_MLP_FWD_PIN=$q timeout 300 python scripts/ab_kernels.py 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/fwd_pin=$q /"; done
done | tee gpurun_out/ab_mlp.log
