"""Developer tool (CPU, the oracle): why does the cosmix benchmark fit need ~100 LM iterations from the prior mean when
SURVEY.md 8d expected 5-10?  Traces mu, rho, |D dx| and chi2 per trial step of the oracle's restated gsl_multifit_nlinear
(oracle/lm.py) on a scaled-down analogue of config 4, from the prior mean and from a start near the generating values, for the
three scalers.  usage: trace_cosmix.py [N P [block]]"""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
from lsqfit_amd import synth
from oracle import lm as olm
from tests import gpu_util as gu

N, P = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (8192, 1024)
block = int(sys.argv[3]) if len(sys.argv) > 3 else 256
d = synth.make_cosmix(N=N, P=P, seed=20263, block=block, prior_corr=True)
normal_eq, chi2_fn, _ = gu.numpy_normal_equations(d)
K = P // 2
dof = N


def run(name, p0, scaler='more', verbose=False):
    olm.TRACE = []
    t0 = time.perf_counter()
    res = olm.lm_normal(p0, normal_eq, chi2_fn, tol=(1e-8, 1e-10, 1e-10), maxit=400, scaler=scaler)
    tr, olm.TRACE = olm.TRACE, None
    rej = sum(1 for r in tr if not r['rho'] > 0)
    print('%-34s scaler %-9s: %3d iterations, %3d trial steps (%d rejected), stop %d, chi2/dof %.4f, %.1f s'
          % (name, scaler, res.nit, len(tr), rej, res.stopping_criterion, res.fnorm2 / dof, time.perf_counter() - t0))
    if verbose:
        print('   it        mu       rho        chi2/dof   chi2_trial/dof    |D dx|      |dx|   da_rms    dw_rms')
        x = np.array(p0, float)
        for r in tr[:12] + [None] + tr[-6:]:
            if r is None:
                print('   ...')
                continue
            print('  %3d  %9.3e  %8.4f  %14.6e  %14.6e  %9.3e %9.3e' % (r['it'], r['mu'], r['rho'], r['chi2'] / dof, r['chi2_trial'] / dof,
                                                                       r['Ddx'], r['dx']))
        its = [r for r in tr if r['rho'] > 0]
        dec = [its[i]['chi2_trial'] / its[i]['chi2'] for i in range(len(its))]
        print('   chi2 ratio per accepted step: first ten %s ... median of the rest %.3f' % (' '.join('%.3f' % v for v in dec[:10]), float(np.median(dec[10:])) if len(dec) > 10 else float('nan')))
        print('   |p - p_true| / sigma_prior at the start: amplitudes rms %.2f, frequencies rms %.2f; phase error w x at x_max: rms %.2f rad'
              % (np.sqrt(np.mean(((p0 - d['p_true'])[:K] / 0.5) ** 2)), np.sqrt(np.mean(((p0 - d['p_true'])[K:] / 0.1) ** 2)),
                 np.sqrt(np.mean(((p0 - d['p_true'])[K:] * d['x'].max()) ** 2))))
    return res


print('cosmix analogue of config 4: N = %d, P = %d, %d-row blocks, dense correlated prior; tol (1e-8, 1e-10, 1e-10)' % (N, P, block))
run('from the prior mean (SURVEY 8d start)', d['p0'], verbose=True)
for sc in ('levenberg', 'marquardt'):
    run('from the prior mean', d['p0'], sc)
rng = np.random.default_rng(6)
for eps in (1e-2, 1e-3, 1e-4):
    run('p_true (1 + %g delta)' % eps, d['p_true'] * (1 + eps * rng.standard_normal(P)))
