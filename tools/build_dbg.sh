#!/bin/bash
# developer build with cycle stamps inside the diagonal-block Cholesky kernel
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/lsqfit_amd/build/dbg
mkdir -p $OUT
for f in gemm_tn_f64 chol potf2_mfma model vecops api batch scipy_methods comm whiten qr jit rankdef robust; do
  EXTRA=""; [ $f = potf2_mfma ] && EXTRA="-mllvm -amdgpu-mfma-vgpr-form"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $EXTRA -DLSQAMD_POTF2_TIMING -c $ROOT/lsqfit_amd/csrc/$f.hip -o $OUT/$f.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/lsqfit_amd/build/libdbg.so $OUT/*.o -ldl
echo $ROOT/lsqfit_amd/build/libdbg.so
