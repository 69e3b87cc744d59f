#!/bin/bash
# A/B of library variants built under kaolin-wisp_amd/csrc/ab/: gpu_ab_lib.sh script.py lib1.so lib2.so ...  ("base" = the product library)
export TMPDIR=/tmp
SCRIPT=$1; shift
for rep in 1; do
for lib in "$@"; do
  if [ "$lib" = base ]; then unset WISP_HIP_LIB; else export WISP_HIP_LIB=$PWD/kaolin-wisp_amd/csrc/ab/$lib; fi
  echo "== $lib (rep $rep)"; timeout 200 python $SCRIPT 2>&1 | grep -v amdgpu.ids | tail -${TAIL:-3}
done
done
