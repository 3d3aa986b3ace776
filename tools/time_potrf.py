"""Developer tool (GPU): time potrf_upper at n = 4096 (with the augmented pad) and check it."""
import ctypes as C
import os
import sys
import numpy as np
import torch
from lsqfit_amd import _lib

lib = _lib.load()
n, ld = 4096, 4096 + 128
rng = np.random.default_rng(0)
G = rng.standard_normal((n + 64, n))
A = G.T @ G + 0.1 * np.eye(n)
wb = lib.lsqamd_op_potrf_work_bytes(n)
work = torch.zeros(wb // 8 + 8, dtype=torch.float64, device='cuda')
info = torch.zeros(4, dtype=torch.int32, device='cuda')
Ah = np.zeros((n, ld))
Ah[:, :n] = np.triu(A)
Ah[:, n] = rng.standard_normal(n)
src = torch.from_numpy(Ah).cuda()
dA = src.clone()
stream = torch.cuda.Stream()          # a real stream (the product runs on one; graph capture needs one)
torch.cuda.set_stream(stream)
st = stream.cuda_stream
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for rep in range(6):
    dA.copy_(src)
    torch.cuda.synchronize()
    ev0.record()
    rc = lib.lsqamd_op_potrf_upper(C.c_void_p(st), C.c_void_p(dA.data_ptr()), n, ld, n + 128, C.c_void_p(work.data_ptr()), wb,
                                   C.c_void_p(info.data_ptr()))
    assert rc == 0, rc
    ev1.record()
    torch.cuda.synchronize()
    ts.append(ev0.elapsed_time(ev1))
U = np.triu(dA.cpu().numpy()[:, :n])
Uref = np.linalg.cholesky(A).T
y = dA.cpu().numpy()[:, n]
yref = np.linalg.solve(Uref.T, Ah[:, n])
print('potrf n=4096: %s ms; info %d; U err %.2e; fwd-subst err %.2e' % (
    ' '.join('%.3f' % t for t in ts), int(info[0]), np.abs(U - Uref).max() / np.abs(Uref).max(),
    np.abs(y - yref).max() / np.abs(yref).max()))
