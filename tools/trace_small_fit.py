"""Developer tool (GPU, under rocprofv3 --kernel-trace --stats): one small NIST fit repeated on a resident problem."""
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
fit = amd.nonlinear_fit(**kw)
t0 = time.perf_counter()
for rep in range(50):
    again = amd.nonlinear_fit(problem=fit.problem, **kw)
dt = (time.perf_counter() - t0) / 50
print('%s: nit %d, resident fit %.3f ms, device run %.3f ms' % (name, again.nit, 1e3 * dt, again.fitter_results.summary.t_run_ms))
