#!/bin/bash
# developer build: ONE source file recompiled with extra -D switches, linked with the regular objects into
# lsqfit_amd/build/lib<name>.so (select it with LSQAMD_LIBPATH).  usage: tools/build_variant.sh <name> <file.hip> -DX [-DY ..]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; SRC=$2; shift 2
python -m lsqfit_amd.build > /dev/null
OUT=$ROOT/lsqfit_amd/build/var_$NAME
mkdir -p $OUT
EXTRA=""; [ $SRC = potf2_mfma.hip ] && EXTRA="-mllvm -amdgpu-mfma-vgpr-form"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $EXTRA "$@" -c $ROOT/lsqfit_amd/csrc/$SRC -o $OUT/${SRC%.hip}.o
OBJS=""
for o in $ROOT/lsqfit_amd/build/*.o; do
  b=$(basename $o)
  if [ $b = ${SRC%.hip}.o ]; then OBJS="$OBJS $OUT/$b"; else OBJS="$OBJS $o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/lsqfit_amd/build/lib$NAME.so $OBJS -ldl
echo $ROOT/lsqfit_amd/build/lib$NAME.so
