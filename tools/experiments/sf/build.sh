#!/bin/bash
# VARIANT build of the round-5 experiment "factorisation streamed behind the J^T J product" (built, correct,
# measured, slower -- not part of the product library since round 6).  Links the regular objects + sf_chol.o into
# lsqfit_amd/build/libsf.so; exp_sf.py / check_sf.py in this directory load that library and bind the two entry points
# themselves (they are declared in sf_chol.h, not in include/lsqfit_amd.h).
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(cd "$HERE/../../.." && pwd)
cd $ROOT
python -m lsqfit_amd.build > /dev/null
OUT=$ROOT/lsqfit_amd/build/var_sf
mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -c $HERE/sf_chol.hip -o $OUT/sf_chol.o
hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/lsqfit_amd/build/libsf.so $ROOT/lsqfit_amd/build/*.o $OUT/sf_chol.o -ldl
echo $ROOT/lsqfit_amd/build/libsf.so
