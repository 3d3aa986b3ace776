"""Developer tool (GPU): where the host time of a WHOLE small fit goes (cProfile over repeated nonlinear_fit calls)."""
import cProfile
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import lsqfit_amd as amd
from tests.helpers import load, nist_problem

NIST = load('nist.json')
name = sys.argv[1] if len(sys.argv) > 1 else 'misra1a'
pr = nist_problem(name, NIST)
model = amd.expr(pr['expr'], ['b%d' % (i + 1) for i in range(pr['P'])], xnames=tuple(pr['columns'][1:]))
x = np.column_stack([pr['x'][c] for c in pr['columns'][1:]])
kw = dict(data=(x, pr['y'], pr['ysd']), model=model, prior=(pr['prior_mean'], pr['prior_sd']), p0=pr['p0'], tol=pr['tol'])
for rep in range(5):
    fit = amd.nonlinear_fit(**kw)
t0 = time.perf_counter()
for rep in range(200):
    fit = amd.nonlinear_fit(**kw)
print('%s: whole fit %.3f ms' % (name, 1e3 * (time.perf_counter() - t0) / 200))
prof = cProfile.Profile()
prof.enable()
for rep in range(200):
    fit = amd.nonlinear_fit(**kw)
prof.disable()
st = pstats.Stats(prof)
st.sort_stats('cumulative').print_stats(45)
