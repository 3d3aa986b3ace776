"""Developer tool (GPU): cycle stamps of the pivot-wave formulation (LSQAMD_POTF2=v4) of the diagonal-block
kernel; needs the -DLSQAMD_POTF2_TIMING build of tools/build_dbg.sh."""
import ctypes as C
import os
import numpy as np
import torch
os.environ['LSQAMD_POTF2'] = 'v4'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, 'lsqfit_amd/build/libdbg.so'))
lib.lsqamd_op_potrf_work_bytes.restype = C.c_size_t
lib.lsqamd_op_potrf_work_bytes.argtypes = [C.c_int64]
lib.lsqamd_op_potrf_upper.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p]
lib.lsqamd_debug_set_potf2_stamps.argtypes = [C.c_void_p]
stamps = torch.zeros(64, dtype=torch.int64, device='cuda')
lib.lsqamd_debug_set_potf2_stamps(stamps.data_ptr())
n = 128
rng = np.random.default_rng(n)
G = rng.standard_normal((n + 20, n))
A = G.T @ G + 0.1 * np.eye(n)
wb = lib.lsqamd_op_potrf_work_bytes(n)
work = torch.zeros(wb // 8 + 8, dtype=torch.float64, device='cuda')
info = torch.zeros(4, dtype=torch.int32, device='cuda')
for rep in range(3):
    dA = torch.from_numpy(np.triu(A)).cuda()
    lib.lsqamd_op_potrf_upper(None, dA.data_ptr(), n, n, n, work.data_ptr(), wb, info.data_ptr())
    torch.cuda.synchronize()
print('U err %.2e' % np.abs(np.triu(dA.cpu().numpy()) - np.linalg.cholesky(A).T).max())
t = stamps.cpu().numpy()
t0 = t[60]
print('total %d cycles' % (t[63] - t0))
for ti in range(8):
    a, b, c = t[4 * ti:4 * ti + 3]
    print('  slab %d: pivot waits from %6d, tile arrives %6d (+%5d), leaf done %6d (%5d) | tile wave 2 done with slab @%6d'
          % (ti, a - t0, b - t0, b - a, c - t0, c - b, t[32 + ti] - t0))
