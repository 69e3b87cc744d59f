#!/bin/bash
# build_ab.sh name "-DFLAG ..." [file.hip ...]: an A/B variant of the library under kaolin-wisp_amd/csrc/ab/<name>.so - the named sources
# recompiled with the extra flags, everything else taken from the product build's objects.
set -e
cd "$(dirname "$0")/../kaolin-wisp_amd/csrc"
NAME=$1; FLAGS=$2; shift 2
mkdir -p ab/obj_$NAME
OBJS=""
for o in hashgrid spc spc_interp spc_grad raymarch render misc nerf_mlp nerf_mlp_bf16 nerf_mlp_wide; do
  if [[ " $* " == *" $o.hip "* ]]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function $FLAGS -c $o.hip -o ab/obj_$NAME/$o.o
    OBJS="$OBJS ab/obj_$NAME/$o.o"
  else
    OBJS="$OBJS $o.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/$NAME.so $OBJS
rm -rf ab/obj_$NAME
echo built ab/$NAME.so
