"""Developer tool (GPU): step time of several small handles created one after the other (each on its own torch stream)."""
import sys
sys.path.insert(0, '.')
sys.argv = sys.argv[:1]
import runpy
import time
import ctypes as C
import numpy as np
import torch
import lsqfit_amd
from lsqfit_amd import _lib, synth
from lsqfit_amd.dist import sharded_problem

d = synth.make_cosmix(N=4096, P=256, seed=20261, block=0, prior_corr=False)
wh = lsqfit_amd.Whitening(d['ymean'], d['yerr'], *d['prior'])


def steps(pr, n, seed=1):
    lib, h = pr.lib, pr.h
    rng = np.random.Generator(np.random.PCG64(seed))
    P = d['p0'].size
    ps = np.concatenate([np.full(P // 2, 0.5), np.full(P // 2, 0.1)])
    conv = True
    for i in range(n + 20):
        if i == 20:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        if conv:
            lib.lsqamd_init(h, _lib.dptr(np.ascontiguousarray(d['p0'] + 0.3 * ps * rng.standard_normal(P))))
            conv = False
        info = C.c_int32(0)
        rc = lib.lsqamd_step(h, C.byref(info))
        if rc != 0 or info.value != 0:
            conv = True
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


hs = []
for k in range(10):
    pr = sharded_problem(d['model'], d['x'], wh, 0, 1)
    pr.set_options((1e-8, 1e-10, 1e-10), 1000)
    hs.append(pr)
    a = steps(pr, 200)
    b = steps(pr, 200)
    print('handle %d  stream %#x  first 200 steps %.4f ms/step, next 200: %.4f' % (k, pr.stream.cuda_stream, a, b), flush=True)
print('again, in order:', ' '.join('%.3f' % steps(pr, 100) for pr in hs))
