#!/bin/bash
# A/B of library variants (kaolin-wisp_amd/csrc/ab/*.so; "base" = the product build) and environment switches on the bench's own
# step: gpu_r5_c.sh lib[,ENV=VAL,...] ...
export TMPDIR=/tmp
for rep in $(seq 1 ${REPS:-2}); do
for spec in "$@"; do
  IFS=, read -r -a parts <<< "$spec"
  lib=${parts[0]}
  (
  if [ "$lib" != base ]; then export WISP_HIP_LIB=$PWD/kaolin-wisp_amd/csrc/ab/$lib.so; fi
  for kv in "${parts[@]:1}"; do export "$kv"; done
  echo "== $spec (rep $rep)"; timeout 200 python scripts/ab_bench_fields.py 2>&1 | grep -v amdgpu.ids | tail -3
  )
done
done
