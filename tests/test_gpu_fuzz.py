"""-m gpu: seeded random small problems through every method, device vs oracle.  Shapes, covariance
structure (diagonal / random blocks with gaps / dense prior / no prior), model family and the
trust-region method are all drawn at random; every case must reproduce the oracle's converged
(p, chi2, cov) to 1e-6 (the north_star tolerance) whenever the oracle converged."""
import os

import numpy as np
import pytest

from oracle import fit as ofit
from tests import gpu_util as gu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def amd():
    import lsqfit_amd
    from lsqfit_amd import _lib
    _lib.load()
    return lsqfit_amd


def random_spd(rng, sd):
    n = sd.size
    U = rng.uniform(0.1, 0.9, (n, 2 * n))
    c = U @ U.T
    d = 1.0 / np.sqrt(np.diag(c))
    return c * np.outer(d, d) * np.outer(sd, sd)


def make_case(seed):
    rng = np.random.default_rng(seed)
    big = seed % 5 == 4                                       # every fifth draw: wider, larger blocks
    K = int(rng.integers(1, 7 if big else 4))
    P = 2 * K
    N = int(rng.integers(P + 1, 400 if big else 48))
    family = ['multiexp', 'cosmix'][int(rng.integers(0, 2))]
    x = np.sort(rng.uniform(0.05, 3.0, N))
    if family == 'multiexp':
        truth = np.concatenate([rng.uniform(0.5, 1.5, K), 0.5 * np.arange(1, K + 1) + rng.uniform(-0.05, 0.05, K)])
        fcn, jac = gu.multiexp_fcn, gu.multiexp_jac
        psd = np.concatenate([np.full(K, 0.5), np.full(K, 0.15)])
        pm = np.concatenate([np.ones(K), 0.5 * np.arange(1, K + 1)])
    else:
        truth = np.concatenate([rng.uniform(0.5, 1.5, K), np.arange(1, K + 1) + rng.uniform(-0.05, 0.05, K)])
        fcn, jac = gu.cosmix_fcn, gu.cosmix_jac
        psd = np.concatenate([np.full(K, 0.5), np.full(K, 0.1)])
        pm = np.concatenate([np.ones(K), np.arange(1, K + 1.0)])
    ybar = fcn(x, truth)
    sd = 0.02 * np.maximum(np.abs(ybar), 0.05)
    blocks = []
    r = 0
    while r < N and rng.random() < 0.7:                      # random blocks with gaps between them
        r += int(rng.integers(0, 4))
        B = int(rng.integers(2, 70 if big else 9))
        if r + B > N:
            break
        blocks.append((r, random_spd(rng, sd[r:r + B])))
        r += B
    cov = np.diag(sd ** 2)
    for r0, c in blocks:
        cov[r0:r0 + c.shape[0], r0:r0 + c.shape[0]] = c
    y = ybar + np.linalg.cholesky(cov) @ rng.standard_normal(N)
    yerr = dict(sdev=sd, blocks=blocks) if blocks else sd
    prior_kind = ['diag', 'dense', 'none'][int(rng.integers(0, 3))]
    if prior_kind == 'none' and family == 'multiexp' and K >= 3:
        prior_kind = 'diag'          # three free exponentials without a prior: not a well-posed fit
    perr = psd if prior_kind != 'dense' else random_spd(rng, psd)
    p0 = pm * (1 + 0.05 * rng.standard_normal(P))
    return dict(family=family, K=K, x=x, y=y, yerr=yerr, cov=cov, pm=pm, perr=perr, prior_kind=prior_kind,
                p0=p0, fcn=fcn, jac=jac)


METHODS = [('mi355x_lm', dict(alg='lm')), ('mi355x_lm', dict(alg='lmaccel')), ('mi355x_lm', dict(alg='dogleg')),
           ('mi355x_lm', dict(alg='ddogleg')), ('mi355x_lm', dict(alg='subspace2D')),
           ('mi355x_trf', dict(method='trf')), ('mi355x_trf', dict(method='dogbox')),
           ('mi355x_trf', dict(method='lm')), ('mi355x_lm', dict(alg='lm', linear=True))]


@pytest.mark.parametrize('seed', range(int(os.environ.get('LSQAMD_FUZZ_CASES', '72'))))   # more: developer sweep
def test_random_problem(amd, seed):
    c = make_case(1000 + seed)
    fitter, opts = METHODS[seed % len(METHODS)]
    opts = dict(opts)
    if opts.pop('linear', False):
        opts['linear'] = list(range(c['K']))
    model = getattr(amd, c['family'])(c['K'])
    noprior = c['prior_kind'] == 'none'
    okw = dict(opts)
    if fitter == 'mi355x_trf':
        okw['fitter'] = 'scipy_least_squares'
        tol = (1e-10, 1e-10, 1e-10)
    else:
        okw['solver'] = 'cholesky'
        tol = 1e-10
    try:
        ref = ofit.nonlinear_fit(c['x'], c['y'], c['cov'], c['fcn'], prior_mean=None if noprior else c['pm'],
                                 prior_err=None if noprior else c['perr'], p0=c['p0'], jac=c['jac'], tol=tol, **okw)
    except np.linalg.LinAlgError:
        pytest.skip('singular normal matrix in the oracle: an ill-posed draw (no prior, nearly equal exponents)')
    sd = np.sqrt(np.diag(ref.cov))
    if not np.all(np.isfinite(sd)) or np.linalg.cond(ref.cov / np.outer(sd, sd)) > 1e9:
        pytest.skip('ill-posed draw: parameter correlations beyond 1 - 1e-9')
    fit = amd.nonlinear_fit(data=(c['x'], c['y'], c['yerr']), model=model, prior=None if noprior else (c['pm'], c['perr']),
                            p0=c['p0'], tol=tol, fitter=fitter, **opts)
    if ref.error is not None or ref.stopping_criterion == 0:
        pytest.skip('the oracle did not converge on this draw')
    assert fit.error is None, fit.error
    assert fit.dof == ref.dof and fit.nblocks == ref.nblocks
    assert abs(fit.chi2 - ref.chi2) <= 1e-6 * max(ref.chi2, 1.0)
    # 1e-6 relative (north_star) for parameters the data determine; a parameter whose error bar is
    # wider than that (no prior, nearly degenerate exponents) is compared in units of its error
    assert np.all(np.abs(fit.pmean - ref.pmean) <= 1e-6 * np.max(np.abs(ref.pmean)) + 1e-4 * ref.psdev)
    assert gu.relmax(fit.cov, ref.cov) < 1e-5
    if not noprior:
        assert fit.logGBF == pytest.approx(ref.logGBF, rel=1e-8, abs=1e-6)
