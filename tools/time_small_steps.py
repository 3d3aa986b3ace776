"""Developer tool (GPU): wall time per LM step of small problems with the phase timers OFF (bench.py keeps
them on, which keeps the step eager) -- captured graphs (default) against LSQAMD_STEP_GRAPH=0.
usage: time_small_steps.py [N P]"""
import ctypes as C
import sys
import time
import numpy as np
sys.path.insert(0, '.')
import torch
import lsqfit_amd as amd
from lsqfit_amd import synth, _lib

shapes = [(int(sys.argv[1]), int(sys.argv[2]))] if len(sys.argv) > 2 else [(4096, 256), (1000, 32), (8192, 1024)]
for N, P in shapes:
    d = synth.make_cosmix(N=N, P=P, seed=20261, block=0, prior_corr=False)
    pr = amd.DeviceProblem(d['model'], d['x'], amd.Whitening(d['ymean'], d['yerr'], *d['prior']))
    pr.set_options((1e-8, 1e-10, 1e-10), 1000)
    rng = np.random.default_rng(3)
    ps = np.concatenate([np.full(P // 2, 0.5), np.full(P // 2, 0.1)])
    lib, h = pr.lib, pr.h
    conv = True
    def step():
        global conv
        if conv:
            p0 = np.ascontiguousarray(d['p0'] + 0.3 * ps * rng.standard_normal(P))
            assert lib.lsqamd_init(h, _lib.dptr(p0)) == 0
            conv = False
        info = C.c_int32(0)
        rc = lib.lsqamd_step(h, C.byref(info))
        assert rc >= 0, lib.lsqamd_last_error(h)
        if rc != 0 or info.value != 0:
            conv = True
    for _ in range(40):
        step()
    best = 1e9
    for rep in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            step()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 200)
    print('N=%d P=%d: %.4f ms per step (graphs replayed: %d)' % (N, P, best * 1e3, (lib.lsqamd_debug_flags(h) >> 2) & 1))
    pr.close()
