#!/bin/bash
# round 4: microbench + kernel split of the octree / codebook backward, then the tests that touch it
export TMPDIR=/tmp
OUT=gpurun_out/r4b; mkdir -p $OUT
REPO="$PWD"
for m in voxel ray; do
  MARCH=$m timeout 300 python scripts/bench_spcbwd.py 2>&1 | grep -v amdgpu.ids | tail -2
  (cd /tmp && rm -rf /tmp/prof_sb && MARCH=$m timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sb -o p -- python "$REPO/scripts/bench_spcbwd.py" > /dev/null 2>&1)
  find /tmp/prof_sb -name "*kernel_stats.csv" -exec cp {} $OUT/spcbwd_${m}_kernel_stats.csv \;
  grep -E "spc_grad|codebook_grad|codebook_dict|fillBuffer" $OUT/spcbwd_${m}_kernel_stats.csv | cut -d, -f1,2,4 | sed 's/(.*"/"/' 
done
timeout 900 python -m pytest tests/test_gpu_0_parity.py tests/test_gpu_1_selfcheck.py -m gpu -q --tb=short -p no:cacheprovider -x \
  -k "octree or codebook or sdf or nglod or flat_params" > $OUT/pytest_new.log 2>&1
echo "tests exit $?: $(tail -1 $OUT/pytest_new.log)"
grep -E "^(FAILED|ERROR)|^E " $OUT/pytest_new.log | head -20
timeout 600 python bench.py --config vqad --steps 100 --pretrain 200 2>&1 | grep -v amdgpu.ids > $OUT/bench_vqad.log
grep -o '"ms_per_step": [0-9.]*' $OUT/bench_vqad.log
