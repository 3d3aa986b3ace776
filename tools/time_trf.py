import time, numpy as np, sys
sys.path.insert(0, '.')
import lsqfit_amd as amd
from lsqfit_amd import synth, _lib
from oracle import fit as ofit
from tests import gpu_util as gu
_lib.load()
for block, pc in [(0, False), (256, True)]:
    d = synth.make_cosmix(N=2048, P=256, seed=4242, block=block, prior_corr=pc)
    K = 128
    lo = np.concatenate([np.full(K, 0.8), np.full(K, -np.inf)]); hi = np.concatenate([np.full(K, 1.2), np.full(K, np.inf)])
    kw = dict(tol=(1e-10, 1e-10, 1e-10), maxit=400)
    for rep in range(2):
        t0 = time.time()
        fit = amd.nonlinear_fit(data=(d['x'], d['ymean'], d['yerr']), model=d['model'], prior=d['prior'], p0=d['p0'], fitter='mi355x_trf', bounds=(lo, hi), **kw)
        t1 = time.time()
        print('device', block, fit.nit, fit.fitter_results.summary.ntrial, fit.fitter_results.summary.njev, round(t1 - t0, 3), 'fit', round(fit.time_fit, 3))
    t0 = time.time()
    ref = ofit.nonlinear_fit(d['x'], d['ymean'], gu.dense_cov(d['yerr'], 2048), gu.cosmix_fcn, prior_mean=d['prior'][0], prior_err=d['prior'][1], p0=d['p0'], jac=gu.cosmix_jac, fitter='scipy_least_squares', bounds=(lo, hi), **kw)
    print('oracle', ref.nit, round(time.time() - t0, 3))
    fit2 = amd.nonlinear_fit(data=(d['x'], d['ymean'], d['yerr']), model=d['model'], prior=d['prior'], p0=d['p0'], fitter='mi355x_trf', **kw)
    print('free trf', fit2.nit, fit2.fitter_results.summary.ntrial, round(fit2.time_fit, 3))
    fit3 = amd.nonlinear_fit(data=(d['x'], d['ymean'], d['yerr']), model=d['model'], prior=d['prior'], p0=d['p0'], tol=1e-10)
    print('lm', fit3.nit, round(fit3.time_fit, 3), np.max(np.abs(fit3.pmean - fit2.pmean)))
