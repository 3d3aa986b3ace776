"""Developer tool: time the full D = dp/d[y, prior] at C4 size (run on the GPU box)."""
import time
import numpy as np
import torch
import lsqfit_amd as amd
from lsqfit_amd import synth

d = synth.make_cosmix(N=65536, P=4096, seed=20263, block=256, prior_corr=True)
wh = amd.Whitening(d['ymean'], d['yerr'], *d['prior'])
pr = amd.DeviceProblem(d['model'], d['x'], wh)
pr.normal(d['p_true'])
pr.get_cov()
pr.timing(True)
for m in (4, 4096):
    G = None if m == 4096 else np.random.default_rng(1).standard_normal((m, 4096))
    t0 = time.perf_counter()
    D = pr.dpdy(G)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print('m=%d  wall %.3f s  shape %s' % (m, t1 - t0, D.shape))
    print({k: v for k, v in pr.timings().items() if v[1]})
    pr.timing_reset()
