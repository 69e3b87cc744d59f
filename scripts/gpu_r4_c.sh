#!/bin/bash
# round 4: A/B library variants of the octree / codebook scatter (scripts/build_ab.sh NAME "-DFLAG" spc_grad.hip): LIBS="base noatomic ..."
export TMPDIR=/tmp
for lib in ${LIBS:-base noatomic fewtails}; do
  if [ "$lib" = base ]; then unset WISP_HIP_LIB; else export WISP_HIP_LIB=$PWD/kaolin-wisp_amd/csrc/ab/$lib.so; fi
  for m in voxel ray; do echo "== $lib $m"; MARCH=$m timeout 300 python scripts/bench_spcbwd.py 2>&1 | grep -v amdgpu.ids | tail -1; done
done
