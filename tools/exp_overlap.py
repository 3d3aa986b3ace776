"""Developer experiment (GPU): can the blocked Cholesky of a P = 4096 matrix run CONCURRENTLY with a saturating
fp64-MFMA product on another stream?  (Round-2 review, item 2: hiding the replicated factorisation behind the
SYRK.)  Stream A runs C = X^T X (the TN GEMM the SYRK is made of; 4096 x 4096 output, K rows), stream B -- high
priority -- potrf_upper of an independent SPD matrix.  Serial time = t(GEMM) + t(potrf); perfect overlap =
max of the two + what the factorisation's GEMMs steal."""
import ctypes as C
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
from lsqfit_amd import _lib

lib = _lib.load()
P = 4096
dev = torch.device('cuda')
rng = np.random.default_rng(0)
A0 = rng.standard_normal((P, P))
A0 = A0 @ A0.T / P + 4.0 * np.eye(P)
lda = P + 128
Apad = np.zeros((P, lda))
Apad[:, :P] = A0
A_t = torch.tensor(Apad, device=dev)
A_w = torch.empty_like(A_t)
work = torch.empty(lib.lsqamd_op_potrf_work_bytes(P), dtype=torch.uint8, device=dev)
info = torch.zeros(1, dtype=torch.int32, device=dev)
sa = torch.cuda.Stream()
sb = torch.cuda.Stream(priority=-1)        # high priority


def gemm(stream, X, Cm, K):
    rc = lib.lsqamd_op_gemm_tn(C.c_void_p(stream.cuda_stream), P, P, K, 1.0, C.c_void_p(X.data_ptr()), P, C.c_void_p(X.data_ptr()), P, 0.0,
                               C.c_void_p(Cm.data_ptr()), P, 1, 0)
    assert rc == 0


def potrf(stream):
    A_w.copy_(A_t)
    rc = lib.lsqamd_op_potrf_upper(C.c_void_p(stream.cuda_stream), C.c_void_p(A_w.data_ptr()), P, lda, lda, C.c_void_p(work.data_ptr()),
                                   work.numel(), C.c_void_p(info.data_ptr()))
    assert rc == 0


def timed(fn, reps=5):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


for K in (8192, 16384, 65536):
    X = torch.randn(K, P, dtype=torch.float64, device=dev)
    Cm = torch.empty(P, P, dtype=torch.float64, device=dev)
    with torch.cuda.stream(sb):
        A_w.copy_(A_t)
    torch.cuda.synchronize()
    t_g = timed(lambda: gemm(sa, X, Cm, K))

    def only_potrf():
        with torch.cuda.stream(sb):
            potrf(sb)
    t_p = timed(only_potrf)

    def both():
        gemm(sa, X, Cm, K)
        with torch.cuda.stream(sb):
            potrf(sb)
    t_b = timed(both)

    def both_late():          # the factorisation is queued when the product is a third of the way through
        gemm(sa, X, Cm, K)
        time.sleep(t_g / 3e3)
        with torch.cuda.stream(sb):
            potrf(sb)
    t_l = timed(both_late)
    print('K = %6d: product %.2f ms, potrf_upper(4096) %.2f ms (incl. a 134 MB copy), serial %.2f | both streams at once %.2f ms, '
          'factorisation queued a third into the product %.2f ms' % (K, t_g, t_p, t_g + t_p, t_b, t_l))
    assert int(info.item()) == 0
