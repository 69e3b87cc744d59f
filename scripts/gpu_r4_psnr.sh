#!/bin/bash
# 1000-step PSNR parity runs of the three GPU regimes against the CPU oracle logs under profiles/ (the fp32 oracle for the fused
# fp32 / bf16 steps, the fp16-autocast-faithful oracle for the unchanged-trainer regime)
export TMPDIR=/tmp
OUT=gpurun_out/r4psnr; mkdir -p $OUT
timeout 600 python scripts/psnr_parity.py --backend hip --out $OUT/r04_psnr_parity_hip.log > /dev/null 2>&1
timeout 600 python scripts/psnr_parity.py --backend hip --amp --out $OUT/r04_psnr_parity_hip_bf16.log > /dev/null 2>&1
timeout 900 python scripts/psnr_parity.py --backend dropin --out $OUT/r04_psnr_parity_dropin.log > /dev/null 2>&1
python scripts/psnr_parity.py --compare $OUT/r04_psnr_parity_hip.log profiles/r02_psnr_parity_oracle.log > $OUT/r04_psnr_parity_compare.txt; tail -1 $OUT/r04_psnr_parity_compare.txt
python scripts/psnr_parity.py --compare $OUT/r04_psnr_parity_hip_bf16.log profiles/r02_psnr_parity_oracle.log > $OUT/r04_psnr_parity_hip_bf16_compare.txt; tail -1 $OUT/r04_psnr_parity_hip_bf16_compare.txt
python scripts/psnr_parity.py --compare $OUT/r04_psnr_parity_dropin.log profiles/r04_psnr_parity_oracle_half.log > $OUT/r04_psnr_parity_dropin_vs_half_oracle.txt; tail -1 $OUT/r04_psnr_parity_dropin_vs_half_oracle.txt
python scripts/psnr_parity.py --compare $OUT/r04_psnr_parity_dropin.log profiles/r02_psnr_parity_oracle.log > $OUT/r04_psnr_parity_dropin_vs_fp32_oracle.txt; tail -1 $OUT/r04_psnr_parity_dropin_vs_fp32_oracle.txt
