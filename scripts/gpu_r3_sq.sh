export TMPDIR=/tmp
mkdir -p gpurun_out/r3sq
REPO="$PWD"; OUT="$REPO/gpurun_out/r3sq"
RX="hashgrid|mlp_|nerf_mlp"
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  (cd /tmp && rm -rf /tmp/sq_$i && timeout 400 rocprofv3 --pmc $grp --kernel-trace --kernel-include-regex "$RX" --output-format csv -d /tmp/sq_$i -o p -- python "$REPO/bench.py" --pmc-child > "$OUT/sq_$i.log" 2>&1)
  python scripts/pmc_summary.py /tmp/sq_$i > $OUT/r03_pmc_sq_group$i.csv 2>> $OUT/sq_$i.log
  wc -l $OUT/r03_pmc_sq_group$i.csv
done
