#!/bin/bash
# PMC passes (one counter group per run, kernel-trace only) for the hot-path kernels of the default bench command.
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
REPO="$PWD"
RX="hashgrid|mlp_|raymarch|composite|adamw"
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum"; do
  tag=$(echo $grp | tr ' ' '_')
  (cd /tmp && rm -rf /tmp/pmc_$tag && timeout 600 rocprofv3 --pmc $grp --kernel-trace --kernel-include-regex "$RX" --output-format csv -d /tmp/pmc_$tag -o p -- python "$REPO/bench.py" --pmc-child > "$REPO/gpurun_out/pmc/$tag.log" 2>&1)
  python scripts/pmc_summary.py /tmp/pmc_$tag > gpurun_out/pmc/$tag.csv 2>> gpurun_out/pmc/$tag.log
  tail -1 gpurun_out/pmc/$tag.log | cut -c1-200
  cat gpurun_out/pmc/$tag.csv | cut -c1-220
done
