#!/bin/bash
# Round 3, first GPU session: new drop-in / loss-scale tests, the whole parity suite, the default bench line (with
# dropin_regime), and SQ counter passes (MFMA busy, instruction mix, LDS stalls) for decoder + hash-grid kernels.
export TMPDIR=/tmp
mkdir -p gpurun_out/r3a
REPO="$PWD"
OUT="$REPO/gpurun_out/r3a"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "loss_scale or non_finite or dropin" > $OUT/pytest_new.log 2>&1
echo "pytest exit: $?" >> $OUT/pytest_new.log; tail -30 $OUT/pytest_new.log | cut -c1-400
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit: $?" >> $OUT/pytest_gpu.log; tail -8 $OUT/pytest_gpu.log | cut -c1-300
timeout 900 python bench.py 2>&1 | grep -v amdgpu.ids > $OUT/bench.log; tail -1 $OUT/bench.log | cut -c1-3000
# SQ counters: one group per pass, kernel-trace only
RX="hashgrid|mlp_|nerf_mlp"
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  (cd /tmp && rm -rf /tmp/sq_$i && timeout 400 rocprofv3 --pmc $grp --kernel-trace --kernel-include-regex "$RX" --output-format csv -d /tmp/sq_$i -o p -- python "$REPO/bench.py" --pmc-child > "$OUT/sq_$i.log" 2>&1)
  python scripts/pmc_summary.py /tmp/sq_$i > $OUT/r03_pmc_sq_group$i.csv 2>> $OUT/sq_$i.log
  tail -2 $OUT/sq_$i.log | cut -c1-200
  wc -l $OUT/r03_pmc_sq_group$i.csv
done
