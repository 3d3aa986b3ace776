"""Developer tool (GPU): wall time of WHOLE small fits through lsqfit_amd.nonlinear_fit (set-up, LM iterations,
covariance, host reductions) -- the shape lsqfit is used at every day (examples/nist.py: 2-9 parameters)."""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import lsqfit_amd as amd
from tests.helpers import load, nist_problem

NIST = load('nist.json')
for name in ('misra1a', 'chwirut2', 'thurber', 'gauss1'):
    pr = nist_problem(name, NIST)
    model = amd.expr(pr['expr'], ['b%d' % (i + 1) for i in range(pr['P'])], xnames=tuple(pr['columns'][1:]))
    x = np.column_stack([pr['x'][c] for c in pr['columns'][1:]])
    kw = dict(data=(x, pr['y'], pr['ysd']), model=model, prior=(pr['prior_mean'], pr['prior_sd']), p0=pr['p0'], tol=pr['tol'])
    fit = amd.nonlinear_fit(**kw)            # warm: hiprtc cache, first-launch costs
    ts = []
    for rep in range(20):
        t0 = time.perf_counter()
        fit = amd.nonlinear_fit(**kw)
        ts.append(time.perf_counter() - t0)
    s = fit.fitter_results.summary
    # the same problem kept resident: only lsqamd_run
    t0 = time.perf_counter()
    for rep in range(20):
        again = amd.nonlinear_fit(problem=fit.problem, **kw)
    resident = (time.perf_counter() - t0) / 20
    print('%-10s N=%4d P=%d  nit %3d  whole fit %.2f ms (min %.2f)   on a resident problem %.2f ms   device run %.2f ms'
          % (name, pr['y'].size, pr['P'], fit.nit, 1e3 * np.median(ts), 1e3 * min(ts), 1e3 * resident, s.t_run_ms))

# the everyday shape: a few dozen CORRELATED points (one dense covariance block), a handful of parameters
rng = np.random.default_rng(7)
for N in (48, 200):
    xs = np.linspace(0.1, 4.0, N)
    pt = np.array([2.0, 0.9, 0.5, 0.25])
    sd = 0.01 * (1.0 + xs)
    cov = np.outer(sd, sd) * 0.6 ** np.abs(np.subtract.outer(np.arange(N), np.arange(N)))
    ys = pt[0] * np.exp(-pt[1] * xs) + pt[2] * np.exp(-pt[3] * xs) + np.linalg.cholesky(cov) @ rng.standard_normal(N)
    kw = dict(data=(xs, ys, dict(sdev=sd, blocks=[(0, cov)])), model=amd.expr('a*exp(-b*x) + c*exp(-d*x)', ['a', 'b', 'c', 'd']),
              prior=(pt, np.array([1.0, 0.5, 0.5, 0.2])), p0=pt * 1.1)
    fit = amd.nonlinear_fit(**kw)
    ts = []
    for rep in range(20):
        t0 = time.perf_counter()
        fit = amd.nonlinear_fit(**kw)
        ts.append(time.perf_counter() - t0)
    t0 = time.perf_counter()
    for rep in range(20):
        again = amd.nonlinear_fit(problem=fit.problem, **kw)
    resident = (time.perf_counter() - t0) / 20
    print('%-10s N=%4d P=4  nit %3d  whole fit %.2f ms (min %.2f)   on a resident problem %.2f ms   device run %.2f ms'
          % ('corr-block', N, fit.nit, 1e3 * np.median(ts), 1e3 * min(ts), 1e3 * resident, fit.fitter_results.summary.t_run_ms))
