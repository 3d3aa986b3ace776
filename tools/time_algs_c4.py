"""Developer tool (GPU): where does an iteration of the dogleg-family methods spend its time at config 4?
phase timers (HIP events) of one whole fit per algorithm + the wall clock around lsqamd_run."""
import ctypes as C
import sys
import time
import numpy as np
sys.path.insert(0, '.')
import torch
import lsqfit_amd
from lsqfit_amd import _lib, synth
from lsqfit_amd.dist import sharded_problem

N, P = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (65536, 4096)
d = synth.make_cosmix(N=N, P=P, seed=20263, block=256, prior_corr=True)
wh = lsqfit_amd.Whitening(d['ymean'], d['yerr'], *d['prior'])
pr = sharded_problem(d['model'], d['x'], wh, 0, 1)
lib, h = pr.lib, pr.h
for alg in ('lm', 'dogleg', 'subspace2D', 'ddogleg'):
    for timers in (False, True):
        pr.set_options((1e-8, 1e-10, 1e-10), 200, alg=alg)
        if timers:
            pr.timing(True)
            pr.timing_reset()
        s = _lib.Summary()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc = lib.lsqamd_run(h, _lib.dptr(np.ascontiguousarray(d['p0'])), C.byref(s))
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        if timers:
            tm = pr.timings()
            pr.timing(False)
            tot = sum(v[0] for v in tm.values())
            print('   phases (total ms, calls): ' + ', '.join('%s %.1f x%d' % (k, v[0], v[1]) for k, v in tm.items() if v[1]) + ' | sum %.1f ms of %.1f wall' % (tot, 1e3 * wall))
        else:
            print('%-10s rc %d nit %d trials %d nfev %d njev %d  wall %.1f ms = %.2f ms per iteration, device %.1f ms' % (alg, rc, s.nit, s.ntrial, s.nfev, s.njev, 1e3 * wall, 1e3 * wall / max(1, s.nit), s.t_run_ms))
