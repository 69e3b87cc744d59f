#!/bin/bash
# Round 3 validation on the GPU box: whole parity suite, smoke, the default bench line (live PMC traffic, CPU baseline, secondary
# configs, drop-in regime), fp32 line, rocprofv3 kernel stats + step timelines of the same command, FETCH/WRITE/TCC counter
# passes, PSNR parity of the HIP path against the stored oracle log.
export TMPDIR=/tmp
TAG=${1:-r03}
mkdir -p gpurun_out/$TAG
OUT="$PWD/gpurun_out/$TAG"
REPO="$PWD"
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit: $?" >> $OUT/pytest_gpu.log; tail -6 $OUT/pytest_gpu.log | cut -c1-300
timeout 300 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids > $OUT/smoke.log; echo "smoke exit $?" >> $OUT/smoke.log; tail -2 $OUT/smoke.log
timeout 1200 python bench.py 2>&1 | grep -v amdgpu.ids > $OUT/bench_bf16.log; tail -1 $OUT/bench_bf16.log > $OUT/${TAG}_bench_default.json; cut -c1-400 $OUT/${TAG}_bench_default.json
timeout 600 python bench.py --precision fp32 --no-cpu-baseline --no-pmc --no-configs --dropin-steps 0 --steps 40 2>&1 | grep -v amdgpu.ids | tail -1 > $OUT/${TAG}_bench_fp32.json; cut -c1-300 $OUT/${TAG}_bench_fp32.json
(cd /tmp && rm -rf /tmp/prof && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python "$REPO/bench.py" --steps 60 --pretrain 300 --eval-rays 0 --no-cpu-baseline --no-pmc --no-configs --dropin-steps 0 > "$OUT/prof.log" 2>&1)
find /tmp/prof -name "*kernel_stats.csv" -exec sh -c 'head -41 "$1" > '"$OUT/${TAG}_bench_default_kernel_stats.csv" _ {} \;
python scripts/trace_gaps.py /tmp/prof hashgrid_fwd 100 > $OUT/${TAG}_step_timeline_2p21.txt 2>&1; head -30 $OUT/${TAG}_step_timeline_2p21.txt
python scripts/trace_gaps.py /tmp/prof > $OUT/${TAG}_step_timeline_2p18.txt 2>&1
RX="hashgrid|mlp_|raymarch|composite|adamw|optim"
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum"; do
  tag=$(echo $grp | tr ' ' '_')
  (cd /tmp && rm -rf /tmp/pmc_$tag && timeout 400 rocprofv3 --pmc $grp --kernel-trace --kernel-include-regex "$RX" --output-format csv -d /tmp/pmc_$tag -o p -- python "$REPO/bench.py" --pmc-child > "$OUT/pmc_$tag.log" 2>&1)
  python scripts/pmc_summary.py /tmp/pmc_$tag > $OUT/${TAG}_pmc_$tag.csv 2>> $OUT/pmc_$tag.log
  wc -l $OUT/${TAG}_pmc_$tag.csv
done
timeout 900 python scripts/psnr_parity.py --backend hip --out $OUT/${TAG}_psnr_parity_hip.log > $OUT/psnr_hip.out 2>&1
python scripts/psnr_parity.py --compare $OUT/${TAG}_psnr_parity_hip.log profiles/r02_psnr_parity_oracle.log > $OUT/${TAG}_psnr_parity_compare.txt 2>&1; tail -3 $OUT/${TAG}_psnr_parity_compare.txt
