#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "mlp or decoder or nerf or pipeline" > gpurun_out/pytest_mlp.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_mlp.log
tail -4 gpurun_out/pytest_mlp.log
for rep in 1 2; do
for q in 0 1 2 3 4; do WISP_MLP_FWD_PIN=$q timeout 300 python scripts/ab_kernels.py 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/pin=$q /"; done
done | tee gpurun_out/ab_mlp.log
