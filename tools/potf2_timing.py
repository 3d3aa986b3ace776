"""Developer tool (GPU): correctness + in-kernel phase timing of the diagonal-block
Cholesky kernel, using the -DLSQAMD_POTF2_TIMING build made by tools/build_dbg.sh."""
import ctypes as C
import os
import sys

import numpy as np
import torch

os.environ['LSQAMD_POTF2'] = 'v3'   # these stamps are the four-wave formulation's (potf2_timing_v4.py: the default kernel)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, 'lsqfit_amd/build/libdbg.so'))
lib.lsqamd_op_potrf_work_bytes.restype = C.c_size_t
lib.lsqamd_op_potrf_work_bytes.argtypes = [C.c_int64]
lib.lsqamd_op_potrf_upper.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p,
                                      C.c_size_t, C.c_void_p]
lib.lsqamd_debug_set_potf2_stamps.argtypes = [C.c_void_p]
stamps = torch.zeros(64, dtype=torch.int64, device='cuda')
lib.lsqamd_debug_set_potf2_stamps(stamps.data_ptr())
for n in [128, 33, 96, 300]:
    rng = np.random.default_rng(n)
    G = rng.standard_normal((n + 20, n))
    A = G.T @ G + 0.1 * np.eye(n)
    wb = lib.lsqamd_op_potrf_work_bytes(n)
    work = torch.zeros(wb // 8 + 8, dtype=torch.float64, device='cuda')
    info = torch.zeros(4, dtype=torch.int32, device='cuda')
    for rep in range(2):
        dA = torch.from_numpy(np.triu(A)).cuda()
        lib.lsqamd_op_potrf_upper(None, dA.data_ptr(), n, n, n, work.data_ptr(), wb, info.data_ptr())
        torch.cuda.synchronize()
    U = np.triu(dA.cpu().numpy())
    Uref = np.linalg.cholesky(A).T
    nb0 = min(n, 128)
    w = work.cpu().numpy()[:128 * 128].reshape(128, 128)[:nb0, :nb0]
    print(n, 'info', int(info[0]), 'U err %.2e inv err %.2e' % (
        np.abs(U - Uref).max(), np.abs(w @ Uref[:nb0, :nb0] - np.eye(nb0)).max()))
    if n == 128:
        t = stamps.cpu().numpy()
        t0 = t[60]
        print('kernel: load %d | slabs %d | store %d | total %d cycles' % (t[61] - t[60], t[62] - t[61], t[63] - t[62], t[63] - t[60]))
        for ti in range(8):
            p0, p1 = t[3 * ti:3 * ti + 2]
            a0, a1, a2 = t[32 + 3 * ti:35 + 3 * ti]
            print('  slab %d  leaf start %6d  leaf %5d | wave 0: scaled @%6d  B2 @%6d  updated @%6d'
                  % (ti, p0 - t0, p1 - p0, a0 - t0, a1 - t0, a2 - t0))
