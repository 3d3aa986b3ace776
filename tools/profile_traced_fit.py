"""Developer tool (GPU): where the host time of a whole small TRACED fit goes (a dictionary-valued Python fit function)."""
import cProfile
import pstats
import sys
import time
sys.path.insert(0, '.')
sys.argv = sys.argv[:1]
import numpy as np
import lsqfit_amd as amd
import importlib
tm = importlib.import_module('tools.time_merged_small') if False else None
rng = np.random.default_rng(5)
nkey, nt, nexp = 3, 24, 2
t = np.arange(1.0, nt + 1)
a = rng.uniform(0.4, 1.0, (nkey, nexp))
E = np.array([0.5, 1.1])
x, y, sd = {}, {}, {}
for k in range(nkey):
    key = 'c%d' % k
    x[key] = t
    f = sum(a[k, n] * np.exp(-E[n] * t) for n in range(nexp))
    sd[key] = 0.01 * f
    y[key] = f + sd[key] * rng.standard_normal(nt)


def fcn(x, p):
    return {('c%d' % k): sum(p['a%d' % k][n] * np.exp(-p['E'][n] * x['c%d' % k]) for n in range(nexp)) for k in range(nkey)}


pm, ps = dict(E=E * 1.05), dict(E=np.full(nexp, 0.5))
for k in range(nkey):
    pm['a%d' % k] = np.full(nexp, 0.7)
    ps['a%d' % k] = np.full(nexp, 0.5)
kw = dict(data=(x, y, sd), fcn=fcn, prior=(pm, ps))
for rep in range(5):
    fit = amd.nonlinear_fit(**kw)
t0 = time.perf_counter()
for rep in range(200):
    fit = amd.nonlinear_fit(**kw)
print('whole traced fit %.3f ms' % (1e3 * (time.perf_counter() - t0) / 200))
prof = cProfile.Profile()
prof.enable()
for rep in range(200):
    fit = amd.nonlinear_fit(**kw)
prof.disable()
pstats.Stats(prof).sort_stats('cumulative').print_stats(40)
