"""Developer tool (GPU): post-fit covariance at the named shape, normal-equation route and QR-grade route
(phase timer 'covar' of the handle)."""
import sys
import time
import numpy as np
sys.path.insert(0, '.')
import torch
import lsqfit_amd as amd
from lsqfit_amd import synth

N, P = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (65536, 4096)
d = synth.make_cosmix(N=N, P=P, seed=20263, block=256, prior_corr=True)
wh = amd.Whitening(d['ymean'], d['yerr'], *d['prior'])
for solver in ('cholesky', 'qr'):
    pr = amd.DeviceProblem(d['model'], d['x'], wh)
    pr.timing(True)
    t0 = time.perf_counter()
    fit = amd.nonlinear_fit(data=(d['x'], d['ymean'], d['yerr']), model=d['model'], prior=d['prior'], p0=d['p0'],
                            problem=pr, solver=solver, maxit=3)
    torch.cuda.synchronize()
    tm = pr.timings()
    print('%-8s fit (3 iterations + covariance) %.1f ms wall; covar phase %.2f ms x %d; psdev[0] %.6e' % (
        solver, (time.perf_counter() - t0) * 1e3, tm['covar'][0] / max(tm['covar'][1], 1), tm['covar'][1], fit.psdev[0]))
    pr.close()
