"""Pin the oracle's LM driver + covariance on the NIST StRD certified values
and on the lsqfit summaries the reference ships (examples/nist.py, nist.out)."""
import numpy as np
import pytest

from oracle import fit as ofit
from oracle import gvar_lite
from tests.helpers import load, nist_problem

NIST = load('nist.json')
NAMES = sorted(NIST)


def _sig_fmt(v, nsig):
    return float('%.*g' % (nsig, v))


@pytest.mark.parametrize('name', NAMES)
@pytest.mark.parametrize('solver', ['qr'])
def test_nist_matches_certified(name, solver):
    pr = nist_problem(name, NIST)
    fit = ofit.nonlinear_fit(pr['x'], pr['y'], pr['ysd'], pr['fcn'],
                             prior_mean=pr['prior_mean'], prior_err=pr['prior_sd'],
                             p0=pr['p0'], tol=pr['tol'], solver=solver)
    # the reference's own acceptance test (examples/nist.py:85-99): string equal,
    # or every parameter within sigma/10 of the expected string's value
    got = gvar_lite.fmt_array(fit.pmean, fit.psdev)
    em, es = gvar_lite.parse_array(pr['expected_p'][1:-1].split())
    if got != pr['expected_p']:
        assert np.all(np.abs(fit.pmean - em) <= np.maximum(es, fit.psdev) / 10.), (got, pr['expected_p'])
    # NIST certified parameters / standard deviations: the 200x-wide priors shift
    # them by ~(sd/prior_sd)^2 relative, so sigma/100 and 1e-3 are safe bounds
    assert np.all(np.abs(fit.pmean - pr['certified']) <= 1e-2 * pr['certified_sd'] + 1e-9 * np.abs(pr['certified']))
    np.testing.assert_allclose(fit.psdev, pr['certified_sd'], rtol=2e-3)
    # lsqfit's printed summary (examples/nist.out)
    out = pr['out']
    assert fit.dof == out['dof']
    assert '%.2g' % (fit.chi2 / fit.dof) == out['chi2_dof'] or abs(fit.chi2 / fit.dof - float(out['chi2_dof'])) < 0.006
    assert abs(fit.Q - float(out['Q'])) < 0.006
    assert '%.5g' % fit.logGBF == out['logGBF'], (fit.logGBF, out['logGBF'])
    # every nist.out line stops on xtol ('1e-10*')
    assert fit.stopping_criterion == 1


@pytest.mark.parametrize('name', ['misra1a', 'chwirut2', 'lanczos3', 'danwood', 'kirby2', 'rat42'])
def test_nist_cholesky_solver_agrees(name):
    pr = nist_problem(name, NIST)
    kw = dict(prior_mean=pr['prior_mean'], prior_err=pr['prior_sd'], p0=pr['p0'], tol=pr['tol'])
    a = ofit.nonlinear_fit(pr['x'], pr['y'], pr['ysd'], pr['fcn'], solver='qr', **kw)
    b = ofit.nonlinear_fit(pr['x'], pr['y'], pr['ysd'], pr['fcn'], solver='cholesky', **kw)
    np.testing.assert_allclose(a.pmean, b.pmean, rtol=1e-6)
    np.testing.assert_allclose(a.chi2, b.chi2, rtol=1e-8)


def test_misra1a_headline_numbers():
    """chi2/dof 0.86 [14], Q 0.61, logGBF -9.2998, 238.9(2.7) 0.0005502(73)."""
    pr = nist_problem('misra1a', NIST)
    fit = ofit.nonlinear_fit(pr['x'], pr['y'], pr['ysd'], pr['fcn'], prior_mean=pr['prior_mean'],
                             prior_err=pr['prior_sd'], p0=pr['p0'], tol=1e-10)
    assert gvar_lite.fmt_array(fit.pmean, fit.psdev) == '[238.9(2.7) 0.0005502(73)]'
    assert '%.2f' % (fit.chi2 / fit.dof) == '0.86' and fit.dof == 14
    assert '%.2f' % fit.Q == '0.61'
    assert '%.4f' % fit.logGBF == '-9.2998'
    # chi2 = RSS/sigma^2 (+ negligible prior term) = NIST dof
    np.testing.assert_allclose(fit.chi2, pr['rss'] / pr['rsd'] ** 2, rtol=1e-4)
    np.testing.assert_allclose(fit.pmean, pr['certified'], rtol=1e-7)
    np.testing.assert_allclose(fit.psdev, pr['certified_sd'], rtol=1e-5)


@pytest.mark.parametrize('alg', ['lmaccel', 'dogleg', 'ddogleg', 'subspace2D'])
@pytest.mark.parametrize('name', ['misra1a', 'chwirut2', 'danwood', 'rat42', 'thurber', 'boxbod'])
def test_nist_other_trust_region_methods(name, alg):
    """every trust-region sub-solver of _gsl.pyx:622-635 reaches the certified
    NIST minimum (converged values are what the reference pins; trajectories are not)."""
    pr = nist_problem(name, NIST)
    kw = dict(prior_mean=pr['prior_mean'], prior_err=pr['prior_sd'], p0=pr['p0'], tol=pr['tol'])
    a = ofit.nonlinear_fit(pr['x'], pr['y'], pr['ysd'], pr['fcn'], **kw)
    b = ofit.nonlinear_fit(pr['x'], pr['y'], pr['ysd'], pr['fcn'], alg=alg, **kw)
    assert b.error is None and b.stopping_criterion in (1, 2)
    np.testing.assert_allclose(b.pmean, a.pmean, rtol=1e-6)
    np.testing.assert_allclose(b.chi2, a.chi2, rtol=1e-8)
    np.testing.assert_allclose(b.cov, a.cov, rtol=1e-5)
    np.testing.assert_allclose(b.pmean, pr['certified'], rtol=1e-5)
