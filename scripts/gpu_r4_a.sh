#!/bin/bash
# round 4, first GPU pass: the new order-free octree / codebook backward (tests + VQAD / NGLOD bench lines + kernel stats), then the
# whole GPU suite once without -x with the statistical margins logged.
export TMPDIR=/tmp
mkdir -p gpurun_out/r4a
REPO="$PWD"
OUT=gpurun_out/r4a
rm -f $OUT/margins.jsonl
WISP_TEST_MARGINS=$REPO/$OUT/margins.jsonl timeout 900 python -m pytest tests/test_gpu_0_parity.py tests/test_gpu_1_selfcheck.py -m gpu -q --tb=short -p no:cacheprovider \
  -k "octree or codebook or sdf or nglod" > $OUT/pytest_new.log 2>&1
echo "new tests exit $?: $(tail -1 $OUT/pytest_new.log)"
grep -E "^(FAILED|ERROR)" $OUT/pytest_new.log | head -20
for cfg in vqad nglod; do
  timeout 600 python bench.py --config $cfg --steps 100 --pretrain 200 2>&1 | grep -v amdgpu.ids > $OUT/bench_$cfg.log
  tail -c 1800 $OUT/bench_$cfg.log; echo
done
(cd /tmp && rm -rf /tmp/prof_vqad && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_vqad -o p -- python "$REPO/bench.py" --config vqad --steps 30 --pretrain 50 > "$REPO/$OUT/prof_vqad.log" 2>&1)
find /tmp/prof_vqad -name "*kernel_stats.csv" -exec cp {} $OUT/r04_vqad_kernel_stats.csv \;
head -14 $OUT/r04_vqad_kernel_stats.csv | cut -c1-170
WISP_TEST_MARGINS=$REPO/$OUT/margins.jsonl timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest_all.log 2>&1
echo "all tests exit $?: $(tail -1 $OUT/pytest_all.log)"
grep -E "^(FAILED|ERROR)" $OUT/pytest_all.log | head -30
