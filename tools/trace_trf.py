"""Developer script: convergence trace of a bounded fit (LSQAMD_TRF_TRACE=1 prints every outer
iteration).  usage: trace_trf.py N P nwall maxit [jac] [dogbox]"""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import lsqfit_amd as amd
from lsqfit_amd import synth, _lib
_lib.load()
N, P, nwall = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
d = synth.make_cosmix(N=N, P=P, seed=4242, block=256, prior_corr=True)
K = P // 2
rng = np.random.default_rng(8)
p0 = d['p0']
tol = (1e-14, 1e-10, 1e-10)
xs = 'jac' if 'jac' in sys.argv else 1.0
free = amd.nonlinear_fit(data=(d['x'], d['ymean'], d['yerr']), model=d['model'], prior=d['prior'], p0=p0,
                         fitter='mi355x_trf', tol=tol, x_scale=xs)
print('free: nit', free.nit, 'crit', free.stopping_criterion, 'chi2/dof', free.chi2 / free.dof,
      'psdev a', free.psdev[:K].min(), free.psdev[:K].max(), 'time', free.time_fit, flush=True)
lo = np.full(P, -np.inf)      # one-sided walls only
hi = np.full(P, np.inf)
walled = rng.choice(K, nwall, replace=False)
up, dn = walled[:nwall // 2], walled[nwall // 2:]
hi[up] = free.pmean[up] - 0.5 * free.psdev[up]
lo[dn] = free.pmean[dn] + 0.5 * free.psdev[dn]
p0 = d['p0'].copy()         # the prior mean, kept well off the walls
p0[:K] = np.clip(p0[:K], lo[:K] + 0.05, hi[:K] - 0.05)
if 'dogbox' in sys.argv:
    import functools
    amd.nonlinear_fit = functools.partial(amd.nonlinear_fit, method='dogbox')
fit = amd.nonlinear_fit(data=(d['x'], d['ymean'], d['yerr']), model=d['model'], prior=d['prior'], p0=p0,
                        fitter='mi355x_trf', bounds=(lo, hi), tol=tol, maxit=int(sys.argv[4]), x_scale=xs)
print('bounded: nit', fit.nit, 'crit', fit.stopping_criterion, 'time', fit.time_fit, 'chi2', fit.chi2, 'free chi2', free.chi2)
