"""Developer experiment (GPU), round 5 review item 1: does reserving CUs make the P = 4096 factorisation run BESIDE a
saturating fp64-MFMA product?  Round 3 (tools/exp_overlap.py) found that two ordinary streams do not overlap: the
product's workgroups hold every CU.  Here stream A is created with hipExtStreamCreateWithCUMask on all CUs but R, stream B
on the R reserved ones (mask bit i -> XCD i mod 8, so R = 8 r reserves r CUs on every XCD).
Reported per R: product alone on its mask, factorisation alone on its mask, both at once."""
import ctypes as C
import sys
import time

import numpy as np
import torch

import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import sf_lib

lib = sf_lib.load()
torch.zeros(1, device='cuda')
hip_path = [ln.split()[-1] for ln in open('/proc/self/maps') if 'libamdhip64' in ln][0]
hip = C.CDLL(hip_path)
print('# HIP runtime:', hip_path)


def masked_stream(bits):
    words = (C.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    if rc != 0:
        raise RuntimeError('hipExtStreamCreateWithCUMask -> %d' % rc)
    return s


P = 4096
K = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
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
X = torch.randn(K, P, dtype=torch.float64, device=dev)
Cm = torch.empty(P, P, dtype=torch.float64, device=dev)
torch.cuda.synchronize()


def gemm(s):
    rc = lib.lsqamd_op_gemm_tn(s, P, P, K, 1.0, C.c_void_p(X.data_ptr()), P, C.c_void_p(X.data_ptr()), P, 0.0,
                               C.c_void_p(Cm.data_ptr()), P, 1, 0)
    assert rc == 0


def potrf(s):
    rc = lib.lsqamd_op_potrf_upper(s, C.c_void_p(A_w.data_ptr()), P, lda, lda, C.c_void_p(work.data_ptr()), work.numel(),
                                   C.c_void_p(info.data_ptr()))
    assert rc == 0


def timed(fn, reps=5):
    best = 1e9
    for _ in range(reps):
        A_w.copy_(A_t)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


plain_a, plain_b = torch.cuda.Stream(), torch.cuda.Stream(priority=-1)
sa0, sb0 = C.c_void_p(plain_a.cuda_stream), C.c_void_p(plain_b.cuda_stream)
tg = timed(lambda: gemm(sa0))
tp = timed(lambda: potrf(sb0))
tb = timed(lambda: (gemm(sa0), potrf(sb0)))
print('K = %d  unmasked: product %.3f ms, potrf_upper(4096) %.3f ms, serial %.3f | both at once %.3f ms' % (K, tg, tp, tg + tp, tb))
for R in (16, 32, 48, 64, 96):
    try:
        sa = masked_stream(range(R, 256))
        sb = masked_stream(range(0, R))
    except Exception as e:          # noqa
        print('R = %d: %s' % (R, e))
        break
    tg = timed(lambda: gemm(sa))
    tp = timed(lambda: potrf(sb))
    tb = timed(lambda: (gemm(sa), potrf(sb)))
    # and the factorisation queued first
    tb2 = timed(lambda: (potrf(sb), gemm(sa)))
    print('R = %3d reserved CUs: product on %d CUs %.3f ms (x%.3f of 256/(256-R) = %.3f), potrf on R CUs %.3f ms | both %.3f ms, '
          'potrf queued first %.3f ms' % (R, 256 - R, tg, tg / timed(lambda: gemm(sa0)), 256.0 / (256 - R), tp, tb, tb2))
    U = np.triu(A_w.cpu().numpy()[:, :P])
    err = np.abs(U.T @ U - A0).max() / np.abs(A0).max()
    print('        factor check after the concurrent run: |U^T U - A| / |A| = %.2e, info %d' % (err, int(info[0])))
