"""Developer timing: generation vs whitening set-up vs problem upload for a BASELINE workload."""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import lsqfit_amd
from lsqfit_amd import synth

for name, (N, P, block, corr, seed) in dict(c3=(8192, 1024, 8192, True, 20262), c4=(65536, 4096, 256, True, 20263)).items():
    t0 = time.perf_counter()
    d = synth.make_cosmix(N=N, P=P, seed=seed, block=block, prior_corr=corr)
    t1 = time.perf_counter()
    for rep in range(2):
        ta = time.perf_counter()
        wh = lsqfit_amd.Whitening(d['ymean'], d['yerr'], *d['prior'])
        tb = time.perf_counter()
        pr = lsqfit_amd.DeviceProblem(d['model'], d['x'], wh)
        tc = time.perf_counter()
        pr.close()
        print('%s rep %d: generate %.2f s | whitening %.3f s | problem %.3f s' % (name, rep, t1 - t0, tb - ta, tc - tb), flush=True)
