import cProfile, pstats, sys, time
import numpy as np
sys.path.insert(0, '.')
import lsqfit_amd as amd
from tests.helpers import load, nist_problem
NIST = load('nist.json')
pr = nist_problem('misra1a', NIST)
model = amd.expr(pr['expr'], ['b%d' % (i + 1) for i in range(pr['P'])], xnames=tuple(pr['columns'][1:]))
x = np.column_stack([pr['x'][c] for c in pr['columns'][1:]])
fit = amd.nonlinear_fit(data=(x, pr['y'], pr['ysd']), model=model, prior=(pr['prior_mean'], pr['prior_sd']), p0=pr['p0'], tol=1e-8)
for i in range(3): fit.bootstrapped_fits(200, seed=1)
prof = cProfile.Profile(); prof.enable()
for i in range(50): fit.bootstrapped_fits(200, seed=1)
prof.disable()
pstats.Stats(prof).sort_stats('tottime').print_stats(18)
