#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3k
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "hashgrid_forward or golden_reference or autograd_module or flagship or image_field" 2>&1 | tail -3
timeout 300 python scripts/exp_l2.py 2>&1 | grep -v amdgpu.ids | head -2
timeout 300 python scripts/ab_kernels.py 2>&1 | grep -v amdgpu.ids | tail -1
