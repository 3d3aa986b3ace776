"""Developer tool (GPU): host-time profile of a whole small fit of CORRELATED data (one dense block)."""
import cProfile
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import lsqfit_amd as amd

N = int(sys.argv[1]) if len(sys.argv) > 1 else 48
rng = np.random.default_rng(7)
xs = np.linspace(0.1, 4.0, N)
pt = np.array([2.0, 0.9, 0.5, 0.25])
sd = 0.01 * (1.0 + xs)
cov = np.outer(sd, sd) * 0.6 ** np.abs(np.subtract.outer(np.arange(N), np.arange(N)))
ys = pt[0] * np.exp(-pt[1] * xs) + pt[2] * np.exp(-pt[3] * xs) + np.linalg.cholesky(cov) @ rng.standard_normal(N)
kw = dict(data=(xs, ys, dict(sdev=sd, blocks=[(0, cov)])), model=amd.expr('a*exp(-b*x) + c*exp(-d*x)', ['a', 'b', 'c', 'd']),
          prior=(pt, np.array([1.0, 0.5, 0.5, 0.2])), p0=pt * 1.1)
for rep in range(5):
    fit = amd.nonlinear_fit(**kw)
t0 = time.perf_counter()
for rep in range(200):
    fit = amd.nonlinear_fit(**kw)
print('N = %d: whole fit %.3f ms' % (N, 1e3 * (time.perf_counter() - t0) / 200))
prof = cProfile.Profile()
prof.enable()
for rep in range(200):
    fit = amd.nonlinear_fit(**kw)
prof.disable()
pstats.Stats(prof).sort_stats('tottime').print_stats(25)
