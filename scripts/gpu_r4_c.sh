#!/bin/bash
# round 4: where the scatter kernel's time goes - A/B variants (no atomics issued / one tail per wave)
export TMPDIR=/tmp
for lib in base noatomic fewtails; do
  if [ "$lib" = base ]; then unset WISP_HIP_LIB; else export WISP_HIP_LIB=$PWD/kaolin-wisp_amd/csrc/ab/$lib.so; fi
  for m in voxel ray; do echo "== $lib $m"; MARCH=$m timeout 300 python scripts/bench_spcbwd.py 2>&1 | grep -v amdgpu.ids | tail -1; done
done
