"""Developer tool (GPU): few parameters, many rows (the usual shape of a big fit): wall time per LM step with the normal
equations formed by the fused kernel (default) -- run again with LSQAMD_FUSED_NORMAL=0 for the J + SYRK + colsum route."""
import ctypes as C
import sys
import time
import numpy as np
sys.path.insert(0, '.')
import torch
import lsqfit_amd as amd
from lsqfit_amd import _lib
from tests.test_gpu_fused_normal import problem, TEXT, NAMES

for N in (20000, 200000, 2000000):
    x, y, sd, prior, pt = problem(N=N, seed=7)
    pr = amd.DeviceProblem(amd.expr(TEXT, NAMES), x, amd.Whitening(y, sd, *prior))
    pr.set_options((1e-12, 1e-14, 1e-14), 100000)
    lib, h = pr.lib, pr.h
    rng = np.random.default_rng(1)
    conv = [True]

    def step():
        if conv[0]:
            p0 = np.ascontiguousarray(prior[0] + 0.2 * rng.standard_normal(6))
            assert lib.lsqamd_init(h, _lib.dptr(p0)) == 0
            conv[0] = False
        info = C.c_int32(0)
        rc = lib.lsqamd_step(h, C.byref(info))
        assert rc >= 0
        if rc != 0 or info.value != 0:
            conv[0] = True
    for _ in range(30):
        step()
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            step()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 100)
    print('N = %8d, P = 6: %.4f ms per LM step  (fused normal equations: %s)' % (N, best * 1e3, bool(lib.lsqamd_debug_flags(h) & 16)))
    pr.close()
