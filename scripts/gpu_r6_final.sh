#!/bin/bash
# Round-6 validation on one GPU box: REPS repetitions of the driver's `pytest -m gpu -x` (statistical margins logged), smoke, the
# default bench line (live PMC + CPU baseline), per-regime kernel stats + timelines of the same command, the other configs,
# PMC / SQ counter passes, the 1000-step PSNR parity runs.  Everything lands under gpurun_out/r6final/.
#   STAGES="tests bench trace configs pmc sq psnr" (default: all)   REPS=2
export TMPDIR=/tmp
OUT=gpurun_out/r6final; mkdir -p $OUT
REPO="$PWD"
STAGES=${STAGES:-"tests bench trace configs pmc sq psnr"}
REPS=${REPS:-2}
has() { [[ " $STAGES " == *" $1 "* ]]; }
rocm-smi --showproductname 2>/dev/null | head -5 > $OUT/device.txt
if has tests; then
  for i in $(seq 1 $REPS); do
    WISP_TEST_MARGINS=$REPO/$OUT/margins_rep$i.jsonl timeout 1200 python -m pytest tests -m gpu -x -q --tb=short -p no:cacheprovider > $OUT/pytest_rep$i.log 2>&1
    echo "rep $i exit $?: $(tail -1 $OUT/pytest_rep$i.log)"
  done
  timeout 300 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids > $OUT/smoke.log; echo "smoke exit ${PIPESTATUS[0]}" >> $OUT/smoke.log; tail -2 $OUT/smoke.log
fi
if has bench; then
  timeout 1500 python bench.py 2>&1 | grep -v amdgpu.ids > $OUT/bench_default.log; tail -1 $OUT/bench_default.log > $OUT/r06_bench_default.json; cut -c1-600 $OUT/r06_bench_default.json
  ( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>&1 | grep -v amdgpu.ids | tail -1 > $OUT/r06_bench_driver_command.json ) 2> $OUT/driver_command_time.txt; cut -c1-300 $OUT/r06_bench_driver_command.json; tail -3 $OUT/driver_command_time.txt
  timeout 600 python bench.py --precision fp32 --no-cpu-baseline --no-pmc --no-configs --dropin-steps 0 --quality-budget 0 --dp-steps 0 --steps 40 2>&1 | grep -v amdgpu.ids | tail -1 > $OUT/r06_bench_fp32.json; cut -c1-300 $OUT/r06_bench_fp32.json
fi
if has trace; then
  bash scripts/regime_stats.sh $OUT r06
fi
if has configs; then
  for cfg in v8 vqad nglod; do
    timeout 600 python bench.py --config $cfg --steps 100 --pretrain 200 2>&1 | grep -v amdgpu.ids | tail -1 > $OUT/r06_bench_$cfg.json
    cut -c1-300 $OUT/r06_bench_$cfg.json; echo
  done
  timeout 600 python bench.py --hidden 128 --no-cpu-baseline --no-pmc --no-configs --dropin-steps 0 --quality-budget 0 --dp-steps 0 --steps 60 2>&1 | grep -v amdgpu.ids | tail -1 > $OUT/r06_bench_hidden128.json
  cut -c1-300 $OUT/r06_bench_hidden128.json; echo
fi
RX="hashgrid|mlp_|raymarch|composite|adamw"
if has pmc; then
  # one counter group per pass (MI355X_MICROARCH.md), for the headline batch (2^18 samples per step) and the large one (2^21)
  for tgt in 262144 2097152; do
    for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
      tag=$(echo $grp | tr ' ' '_')_$tgt
      (cd /tmp && rm -rf /tmp/pmc_$tag && timeout 600 rocprofv3 --pmc $grp --kernel-trace --kernel-include-regex "$RX" --output-format csv -d /tmp/pmc_$tag -o p -- python "$REPO/bench.py" --pmc-child --target-samples $tgt > "$REPO/$OUT/pmc_$tag.log" 2>&1)
      python scripts/pmc_summary.py /tmp/pmc_$tag > $OUT/r06_pmc_$tag.csv 2>> $OUT/pmc_$tag.log
      wc -l $OUT/r06_pmc_$tag.csv
    done
  done
fi
if has sq; then
  i=0
  for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" \
             "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    (cd /tmp && rm -rf /tmp/sq_$i && timeout 400 rocprofv3 --pmc $grp --kernel-trace --kernel-include-regex "hashgrid|mlp_|nerf_mlp" --output-format csv -d /tmp/sq_$i -o p -- python "$REPO/bench.py" --pmc-child --target-samples ${SQ_TARGET:-262144} > "$REPO/$OUT/sq_$i.log" 2>&1)
    python scripts/pmc_summary.py /tmp/sq_$i > $OUT/r06_pmc_sq_group$i.csv 2>> $OUT/sq_$i.log
    wc -l $OUT/r06_pmc_sq_group$i.csv
  done
  python scripts/sq_summary.py $OUT/r06_pmc_sq_group > $OUT/r06_sq_summary.txt 2>&1; head -30 $OUT/r06_sq_summary.txt
fi
if has psnr; then
  timeout 600 python scripts/psnr_parity.py --backend hip --out $OUT/r06_psnr_parity_hip.log > /dev/null 2>&1
  timeout 600 python scripts/psnr_parity.py --backend hip --amp --out $OUT/r06_psnr_parity_hip_bf16.log > /dev/null 2>&1
  timeout 900 python scripts/psnr_parity.py --backend dropin --out $OUT/r06_psnr_parity_dropin.log > /dev/null 2>&1
  python scripts/psnr_parity.py --compare $OUT/r06_psnr_parity_hip.log profiles/r02_psnr_parity_oracle.log > $OUT/r06_psnr_parity_compare.txt; tail -1 $OUT/r06_psnr_parity_compare.txt
  python scripts/psnr_parity.py --compare $OUT/r06_psnr_parity_hip_bf16.log profiles/r02_psnr_parity_oracle.log > $OUT/r06_psnr_parity_hip_bf16_compare.txt; tail -1 $OUT/r06_psnr_parity_hip_bf16_compare.txt
  python scripts/psnr_parity.py --compare $OUT/r06_psnr_parity_dropin.log profiles/r04_psnr_parity_oracle_half.log > $OUT/r06_psnr_parity_dropin_vs_half_oracle.txt; tail -1 $OUT/r06_psnr_parity_dropin_vs_half_oracle.txt
  python scripts/psnr_parity.py --compare $OUT/r06_psnr_parity_dropin.log profiles/r02_psnr_parity_oracle.log > $OUT/r06_psnr_parity_dropin_vs_fp32_oracle.txt; tail -1 $OUT/r06_psnr_parity_dropin_vs_fp32_oracle.txt
fi
