"""-m gpu: f4 of SURVEY.md 8 -- the other trust-region sub-problem solvers of gsl_multifit
(alg = lmaccel / dogleg / ddogleg / subspace2D, src/lsqfit/_gsl.pyx:622-635) on the device,
against the oracle's restatement and the reference's own assertions
(tests/test_lsqfit.py:1700-1725).  Converged values to 1e-6; trajectories are unpinned."""
import warnings

import numpy as np
import pytest

from oracle import lm as olm

from oracle import fit as ofit
from tests import gpu_util as gu
from tests.helpers import load, nist_problem

pytestmark = pytest.mark.gpu
NIST = load('nist.json')
KAT = load('kat.json')
ALGS = ['lmaccel', 'dogleg', 'ddogleg', 'subspace2D']


@pytest.fixture(scope='module')
def amd():
    import lsqfit_amd
    from lsqfit_amd import _lib
    _lib.load()
    return lsqfit_amd


def test_gsl_multifit_cases_on_device(amd):
    """tests/test_lsqfit.py:1700-1725 through the plugin: f = (x - x*)^2 + (x - x*)^4."""
    k = KAT['gsl_multifit']
    xans = np.array(k['xans'])
    terms = ' + '.join('s%d*((p%d - %r)**2 + (p%d - %r)**4)' % (i, i, float(a), i, float(a))
                       for i, a in enumerate(xans))
    model = amd.expr(terms, ['p0', 'p1', 'p2'], xnames=('s0', 's1', 's2'))
    wh = amd.Whitening(np.zeros(3), np.ones(3))
    pr = amd.DeviceProblem(model, np.eye(3), wh)
    for c in k['cases'] + [dict(x0=[0., 0., 0.], alg=a, tol=[1e-10, 0.0, 0.0], stopping_criterion=1,
                                rtol=1e-3) for a in ('dogleg', 'ddogleg')]:
        with warnings.catch_warnings():
            warnings.simplefilter('error')               # in particular: no "covariance undefined"
            ans = amd.mi355x_lm(np.array(c['x0']), 3, None, tol=tuple(c['tol']), alg=c['alg'], problem=pr)
        np.testing.assert_allclose(ans.x, xans, rtol=c['rtol'])
        assert ans.stopping_criterion == c['stopping_criterion'], c
        # the covariance the reference's plugin returns: gsl_multifit_nlinear_covar(J, 0.0) at the end point
        # (src/lsqfit/_gsl.pyx:704-706), in the oracle's restatement -- also when x0[0] sits on its optimum
        # and the first column of J is zero for the whole fit (the dropped direction: zero row and column)
        J = np.diag(2 * (ans.x - xans) + 4 * (ans.x - xans) ** 3)
        lin = olm._DenseLin('qr')
        lin.set(J, np.zeros(3))
        want = lin.covar()
        assert ans.error is None
        assert gu.relmax(ans.cov, want) < 1e-6
        assert ans.cov_dropped == int(np.sum(np.diag(J) == 0.0))
        if c['x0'][0] == 1.0:
            assert ans.cov_dropped == 1 and np.all(ans.cov[0] == 0.0) and np.all(ans.cov[:, 0] == 0.0)
    assert amd.mi355x_lm(np.zeros(3), 3, None, alg='lmaccel', problem=pr).description == \
        'methods = lmaccel/more/cholesky    avmax = 0.75'
    with pytest.raises(ValueError):
        amd.mi355x_lm(np.zeros(3), 3, None, alg='cgst', problem=pr)
    pr.close()


@pytest.mark.parametrize('alg', ALGS)
@pytest.mark.parametrize('name', ['misra1a', 'chwirut2', 'danwood', 'rat42', 'thurber', 'boxbod'])
def test_nist_other_methods_on_device(amd, name, alg):
    pr = nist_problem(name, NIST)
    model = amd.expr(pr['expr'], ['b%d' % (i + 1) for i in range(pr['P'])], xnames=tuple(pr['columns'][1:]))
    x = np.column_stack([pr['x'][c] for c in pr['columns'][1:]])
    fit = amd.nonlinear_fit(data=(x, pr['y'], pr['ysd']), model=model, prior=(pr['prior_mean'], pr['prior_sd']),
                            p0=pr['p0'], tol=pr['tol'], alg=alg)
    ref = ofit.nonlinear_fit(pr['x'], pr['y'], pr['ysd'], pr['fcn'], prior_mean=pr['prior_mean'],
                             prior_err=pr['prior_sd'], p0=pr['p0'], tol=pr['tol'], alg=alg, solver='cholesky')
    assert fit.error is None and fit.stopping_criterion in (1, 2)
    assert gu.relmax(fit.pmean, ref.pmean) < 1e-6
    assert abs(fit.chi2 / ref.chi2 - 1) < 1e-6
    assert gu.relmax(fit.cov, ref.cov) < 1e-5
    np.testing.assert_allclose(fit.pmean, pr['certified'], rtol=1e-5)
    assert fit.description.startswith('methods = %s/more/cholesky' % alg)
    # same trajectory length as the oracle's restatement up to rounding-level accept/reject ties
    # (long, ill-conditioned trajectories such as thurber's part ways after ~20 steps)
    assert abs(fit.nit - ref.nit) <= max(2, ref.nit // 4), (fit.nit, ref.nit)


@pytest.mark.parametrize('alg', ALGS)
@pytest.mark.parametrize('case', ['blocks', 'wide'])
def test_cosmix_other_methods_match_oracle(amd, case, alg):
    from lsqfit_amd import synth
    cfg = dict(blocks=dict(N=512, P=32, seed=61, block=64, prior_corr=True),
               wide=dict(N=700, P=300, seed=62, block=0, prior_corr=True))[case]
    d = synth.make_cosmix(**cfg)
    fit = amd.nonlinear_fit(data=(d['x'], d['ymean'], d['yerr']), model=d['model'], prior=d['prior'], alg=alg)
    ref = gu.oracle_fit(d)               # the minimum does not depend on the method
    assert fit.error is None
    assert gu.relmax(fit.pmean, ref.pmean) < 1e-6
    assert abs(fit.chi2 / ref.chi2 - 1) < 1e-6
    assert gu.relmax(fit.cov, ref.cov) < 1e-6


def test_methods_step_counts_c2(amd):
    """doc/source/overview.rst:2107-2118 claims geodesic acceleration / dogleg can cut the
    iteration count; report it for config 2 (4096, 256) from the prior mean."""
    from lsqfit_amd import synth
    d = synth.make_cosmix(N=4096, P=256, seed=20261, block=0, prior_corr=False)
    base = None
    for alg in ['lm'] + ALGS:
        fit = amd.nonlinear_fit(data=(d['x'], d['ymean'], d['yerr']), model=d['model'], prior=d['prior'], alg=alg)
        assert fit.error is None
        if base is None:
            base = fit
        assert gu.relmax(fit.pmean, base.pmean) < 1e-6 and abs(fit.chi2 / base.chi2 - 1) < 1e-6
        print('%-10s nit=%d nfev=%d time=%.1f ms' % (alg, fit.nit, fit.fitter_results.summary.nfev, 1e3 * fit.time_fit))
