#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python scripts/psnr_parity.py --backend hip --out gpurun_out/r02_psnr_parity_hip.log > gpurun_out/psnr_hip.out 2>&1
tail -3 gpurun_out/r02_psnr_parity_hip.log
python scripts/psnr_parity.py --compare gpurun_out/r02_psnr_parity_hip.log profiles/r02_psnr_parity_oracle.log | tail -4
