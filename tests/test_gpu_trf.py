"""-m gpu: bounded fits -- scipy_least_squares' Trust Region Reflective method
(src/lsqfit/_scipy.py:115-181, bounds src/lsqfit/__init__.py:641-655) on the
device, against oracle/trf.py (itself pinned on scipy: tests/test_oracle_trf.py) and the
reference's own assertions (tests/test_lsqfit.py:1754-1808,:1811-1838).  Fit point, chi2 and
covariance to 1e-6; the evaluation count is compared too (same iterates up to rounding)."""
import numpy as np
import pytest

from oracle import fit as ofit
from oracle import gvar_lite, trf
from tests import gpu_util as gu
from tests.helpers import load, nist_problem

pytestmark = pytest.mark.gpu
NIST = load('nist.json')


@pytest.fixture(scope='module')
def amd():
    import lsqfit_amd
    from lsqfit_amd import _lib
    _lib.load()
    return lsqfit_amd


def multiexp_case(seed, K, N=200):
    rng = np.random.default_rng(seed)
    x = np.linspace(0.05, 3.0, N)
    a = rng.uniform(0.5, 1.5, K)
    E = 0.6 * np.arange(1, K + 1) + rng.uniform(-0.05, 0.05, K)
    truth = np.concatenate([a, E])
    ybar = (a[:, None] * np.exp(-E[:, None] * x)).sum(0)
    ysd = 0.01 * np.abs(ybar)
    y = ybar + ysd * rng.standard_normal(N)
    pm = np.concatenate([np.ones(K), 0.6 * np.arange(1, K + 1)])
    psd = np.concatenate([0.5 * np.ones(K), 0.2 * np.ones(K)])

    def fcn(xx, p):
        return (p[:K, None] * np.exp(-p[K:, None] * xx)).sum(0)

    def jac(xx, p):
        ex = np.exp(-p[K:, None] * xx)
        return np.concatenate([ex, -p[:K, None] * xx * ex]).T
    return x, y, ysd, pm, psd, truth, fcn, jac


def bounds_for(kind, truth, p0):
    P = truth.size
    lo = np.minimum(truth, p0) - 0.5
    hi = np.maximum(truth, p0) + 0.5
    if kind == 'free':
        return None
    if kind == 'loose':
        return lo, hi
    hi2, lo2 = hi.copy(), lo.copy()
    for j in (0, P - 1):                      # a wall 30% of the way from p0 to the optimum
        wall = p0[j] + 0.3 * (truth[j] - p0[j])
        if truth[j] > p0[j]:
            hi2[j] = wall
        else:
            lo2[j] = wall
    if kind == 'active':
        return lo2, hi2
    return lo2, np.where(hi2 < hi, hi2, np.inf)                                   # 'semi': open elsewhere


@pytest.mark.parametrize('method', ['trf', 'dogbox'])
@pytest.mark.parametrize('x_scale', [1.0, 'jac'])
@pytest.mark.parametrize('kind', ['free', 'loose', 'active', 'semi'])
@pytest.mark.parametrize('seed,K', [(1, 1), (2, 2), (3, 3)])
def test_trf_matches_oracle(amd, seed, K, kind, x_scale, method):
    x, y, ysd, pm, psd, truth, fcn, jac = multiexp_case(seed, K)
    p0 = pm * (1.0 + 0.2 * np.cos(np.arange(2 * K) + seed))
    b = bounds_for(kind, truth, p0)
    kw = dict(tol=(1e-8, 1e-8, 1e-8), maxit=300, x_scale=x_scale, method=method)
    ref = ofit.nonlinear_fit(x, y, ysd, fcn, prior_mean=pm, prior_err=psd, p0=p0, jac=jac,
                             fitter='scipy_least_squares', bounds=b, **kw)
    fit = amd.nonlinear_fit(data=(x, y, ysd), model=amd.multiexp(K), prior=(pm, psd), p0=p0,
                            fitter='mi355x_trf', bounds=b, **kw)
    assert fit.error is None
    assert fit.description == 'method = ' + method
    assert gu.relmax(fit.pmean, ref.pmean) < 1e-6
    assert abs(fit.chi2 / ref.chi2 - 1) < 1e-6
    assert gu.relmax(fit.cov, ref.cov) < 1e-6
    assert fit.stopping_criterion == ref.stopping_criterion
    assert abs(fit.nit - ref.nit) <= max(1, ref.nit // 10), (fit.nit, ref.nit)
    if b is not None:
        if method == 'trf':
            assert np.all(fit.pmean > b[0]) and np.all(fit.pmean < b[1])      # strictly feasible
        else:
            assert np.all(fit.pmean >= b[0]) and np.all(fit.pmean <= b[1])
            on = (fit.pmean == b[0]) | (fit.pmean == b[1])                    # exactly on the walls
            assert np.array_equal(on, (ref.pmean == b[0]) | (ref.pmean == b[1]))
        if kind in ('active', 'semi'):
            assert np.any(np.minimum(fit.pmean - b[0], b[1] - fit.pmean) < 1e-6)


def test_reference_bounds_case_on_device(amd):
    """tests/test_lsqfit.py:1780-1808 (array p0): data 0.9(1), 2.2(2); fcn(p) = p; the fit ends on
    the upper bounds 0.5 and 1.0 (assertAlmostEqual: 7 places)."""
    ym, ys = gvar_lite.parse_array(['0.9(1)', '2.2(2)'])
    fit = amd.nonlinear_fit(data=(np.zeros(2), ym, ys), model=amd.identity(2), p0=[0.25, 0.5],
                            fitter='mi355x_trf', bounds=([0.0, 0.0], [0.5, 1.0]))
    assert round(abs(fit.pmean[0] - 0.5), 7) == 0 and round(abs(fit.pmean[1] - 1.0), 7) == 0
    ref = ofit.nonlinear_fit(False, ym, ys, lambda p: p, p0=[0.25, 0.5], jac=lambda p: np.eye(2),
                             fitter='scipy_least_squares', bounds=([0.0, 0.0], [0.5, 1.0]))
    assert gu.relmax(fit.pmean, ref.pmean) < 1e-9 and fit.nit == ref.nit


def test_reference_fitters_case_on_device(amd):
    """tests/test_lsqfit.py:1811-1838: str(fit.p) == '[0.904(98) 2.17(19)]' for method='trf'."""
    ym, ys = gvar_lite.parse_array(['0.9(1)', '2.2(2)'])
    pm, ps = gvar_lite.parse_array(['1.0(5)', '2.0(5)'])
    for method in ('trf', 'dogbox', 'lm'):
        fit = amd.nonlinear_fit(data=(np.zeros(2), ym, ys), model=amd.identity(2), prior=(pm, ps),
                                fitter='mi355x_trf', method=method)
        assert gvar_lite.fmt_array(fit.pmean, fit.psdev) == '[0.904(98) 2.17(19)]'


def test_reference_scipy_least_squares_case_on_device(amd):
    """tests/test_lsqfit.py:1754-1767: f = (x - xans)^2 + (x - xans)^4, tol (1e-15, 1e-8, 1e-15),
    method 'trf' -> stopping_criterion 2 (gtol)."""
    xans = np.arange(3) + 1.0
    terms = ' + '.join('s%d*((p%d - %r)**2 + (p%d - %r)**4)' % (i, i, float(a), i, float(a))
                       for i, a in enumerate(xans))
    model = amd.expr(terms, ['p0', 'p1', 'p2'], xnames=('s0', 's1', 's2'))
    pr = amd.DeviceProblem(model, np.eye(3), amd.Whitening(np.zeros(3), np.ones(3)))
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('error')                   # no "covariance undefined": the reference returns one
        ans = amd.mi355x_trf(np.ones(3), 3, None, tol=(1e-15, 1e-8, 1e-15), method='trf', problem=pr)
    np.testing.assert_allclose(ans.x, xans, rtol=1e-3)
    assert ans.stopping_criterion == 2
    f = lambda x: (x - xans) ** 2 + (x - xans) ** 4
    df = lambda x: np.diag(2 * (x - xans) + 4 * (x - xans) ** 3)
    # covariance: the thresholded-SVD pseudo-inverse of src/lsqfit/_scipy.py:170-175 at the end point; x0[0] sits
    # on its optimum, so the first column of the Jacobian is zero and that direction is dropped
    _, sv, VT = np.linalg.svd(df(ans.x), full_matrices=False)
    keep = sv > np.finfo(float).eps * 3 * sv[0]
    want = (VT[keep].T / sv[keep] ** 2) @ VT[keep]
    assert ans.error is None and ans.cov_dropped == 1 and int(np.sum(~keep)) == 1
    assert gu.relmax(ans.cov, want) < 1e-6
    ref = trf.scipy_least_squares(np.ones(3), 3, f, df, tol=(1e-15, 1e-8, 1e-15))
    # x0[0] sits exactly on its optimum: a zero Jacobian column, where scipy's SVD iteration takes
    # negative shifts; the device keeps B + alpha positive definite, so only the end point is compared
    assert gu.relmax(ans.x, ref.x) < 2e-3
    ans = amd.mi355x_trf(np.zeros(3), 3, None, tol=(1e-15, 1e-8, 1e-15), method='dogbox', problem=pr)   # :1771-1775
    np.testing.assert_allclose(ans.x, xans, rtol=1e-3)
    assert ans.stopping_criterion == 2 and ans.description == 'method = dogbox'
    ref = trf.scipy_least_squares(np.zeros(3), 3, f, df, tol=(1e-15, 1e-8, 1e-15), method='dogbox')
    assert ans.nit == ref.nit and gu.relmax(ans.x, ref.x) < 1e-9
    ans = amd.mi355x_trf(np.zeros(3), 3, None, tol=(1e-8, 1e-15, 1e-15), method='lm', problem=pr)      # :1768-1772
    np.testing.assert_allclose(ans.x, xans, rtol=1e-3)
    assert ans.stopping_criterion == 1 and ans.description == 'method = lm'
    ref = trf.scipy_least_squares(np.zeros(3), 3, f, df, tol=(1e-8, 1e-15, 1e-15), method='lm')
    assert ans.nit == ref.nit and gu.relmax(ans.x, ref.x) < 1e-9
    with pytest.raises(ValueError, match="doesn't support bounds"):
        amd.mi355x_trf(np.ones(3), 3, None, method='lm', bounds=(0.0, 5.0), problem=pr)
    with pytest.raises(ValueError, match='machine epsilon'):
        amd.mi355x_trf(np.ones(3), 3, None, method='lm', tol=(1e-8, 1e-17, 1e-8), problem=pr)
    with pytest.raises(ValueError, match='`method`'):
        amd.mi355x_trf(np.ones(3), 3, None, method='cg', problem=pr)
    with pytest.raises(ValueError, match='outside'):
        amd.mi355x_trf(np.ones(3), 3, None, bounds=(2.0, 3.0), problem=pr)
    with pytest.raises(ValueError, match='strictly less'):
        amd.mi355x_trf(np.ones(3), 3, None, bounds=(2.0, 2.0), problem=pr)
    # the bounds do not outlive the fit: the same problem runs unbounded afterwards
    again = amd.mi355x_lm(np.zeros(3), 3, None, tol=(1e-10, 0.0, 0.0), problem=pr)
    np.testing.assert_allclose(again.x, xans, rtol=1e-3)
    pr.close()


@pytest.mark.parametrize('name', ['misra1a', 'chwirut2', 'danwood', 'rat42', 'boxbod'])
def test_nist_with_positivity_bounds(amd, name):
    """NIST problems (examples/nist.py harness) with every parameter bounded to its certified sign's
    half line: same answer as the oracle, and (bounds inactive at the optimum) as the free fit."""
    pr = nist_problem(name, NIST)
    P = pr['P']
    cert = np.asarray(pr['certified'])
    lo = np.where(cert > 0, 0.0, -np.inf)
    hi = np.where(cert > 0, np.inf, 0.0)
    model = amd.expr(pr['expr'], ['b%d' % (i + 1) for i in range(P)], xnames=tuple(pr['columns'][1:]))
    x = np.column_stack([pr['x'][c] for c in pr['columns'][1:]])
    kw = dict(tol=(1e-10, 1e-10, 1e-10), maxit=2000)
    fit = amd.nonlinear_fit(data=(x, pr['y'], pr['ysd']), model=model, prior=(pr['prior_mean'], pr['prior_sd']),
                            p0=pr['p0'], fitter='mi355x_trf', bounds=(lo, hi), **kw)
    ref = ofit.nonlinear_fit(pr['x'], pr['y'], pr['ysd'], pr['fcn'], prior_mean=pr['prior_mean'],
                             prior_err=pr['prior_sd'], p0=pr['p0'], fitter='scipy_least_squares',
                             bounds=(lo, hi), **kw)
    assert gu.relmax(fit.pmean, ref.pmean) < 1e-6
    assert abs(fit.chi2 / ref.chi2 - 1) < 1e-6
    assert gu.relmax(fit.cov, ref.cov) < 1e-5
    assert gu.relmax(fit.pmean, cert) < 1e-4


@pytest.mark.parametrize('method', ['trf'])      # scipy's dogbox does not converge here within 400 evaluations
@pytest.mark.parametrize('block,prior_corr', [(0, False), (256, True)])
def test_trf_bounded_cosmix_1024x128(amd, block, prior_corr, method):
    """A bench-type problem (cosmix, P = 128; uncorrelated, and 256-row covariance blocks with a
    dense correlated prior) with the amplitudes boxed into [0.8, 1.2]: about 40% of them end on a
    wall.  Fit point, chi2, covariance and the evaluation count against the oracle."""
    from lsqfit_amd import synth
    d = synth.make_cosmix(N=1024, P=128, seed=4242, block=block, prior_corr=prior_corr)
    K = 64
    lo = np.concatenate([np.full(K, 0.8), np.full(K, -np.inf)])
    hi = np.concatenate([np.full(K, 1.2), np.full(K, np.inf)])
    kw = dict(tol=(1e-10, 1e-10, 1e-10), maxit=400, method=method)
    fit = amd.nonlinear_fit(data=(d['x'], d['ymean'], d['yerr']), model=d['model'], prior=d['prior'], p0=d['p0'],
                            fitter='mi355x_trf', bounds=(lo, hi), **kw)
    ref = ofit.nonlinear_fit(d['x'], d['ymean'], gu.dense_cov(d['yerr'], 1024), gu.cosmix_fcn,
                             prior_mean=d['prior'][0], prior_err=d['prior'][1], p0=d['p0'], jac=gu.cosmix_jac,
                             fitter='scipy_least_squares', bounds=(lo, hi), **kw)
    on_wall = np.minimum(fit.pmean[:K] - 0.8, 1.2 - fit.pmean[:K]) < 1e-6
    assert 10 < on_wall.sum() < 55
    assert gu.relmax(fit.pmean, ref.pmean) < 1e-6
    assert abs(fit.chi2 / ref.chi2 - 1) < 1e-6
    assert gu.relmax(fit.cov, ref.cov) < 1e-6
    assert fit.stopping_criterion == ref.stopping_criterion
    # ~100 reflections off the walls: the iterates drift apart at rounding level, the count a little
    assert abs(fit.nit - ref.nit) <= max(2, ref.nit // 4), (fit.nit, ref.nit)


@pytest.mark.parametrize('method', ['trf'])      # scipy's dogbox: 400 evaluations are not enough here either
def test_half_sigma_walls_cosmix_1024x128(amd, method):
    """cosmix (1024, 128), 256-row covariance blocks, dense correlated prior; eight amplitudes get a
    wall half a standard deviation short of their unconstrained optimum; the fit starts from the
    prior mean.  Against the oracle, evaluation count included."""
    from lsqfit_amd import synth
    N, P, K = 1024, 128, 64
    d = synth.make_cosmix(N=N, P=P, seed=4242, block=256, prior_corr=True)
    rng = np.random.default_rng(8)
    data = (d['x'], d['ymean'], d['yerr'])
    free = amd.nonlinear_fit(data=data, model=d['model'], prior=d['prior'], p0=d['p0'], tol=1e-12)
    lo = np.concatenate([free.pmean[:K] - 0.5, np.full(K, -np.inf)])
    hi = np.concatenate([free.pmean[:K] + 0.5, np.full(K, np.inf)])
    walled = rng.choice(K, 8, replace=False)
    hi[walled[:4]] = free.pmean[walled[:4]] - 0.5 * free.psdev[walled[:4]]
    lo[walled[4:]] = free.pmean[walled[4:]] + 0.5 * free.psdev[walled[4:]]
    p0 = d['p0'].copy()                       # the prior mean, kept 0.05 (thousands of sigma) off the walls
    p0[:K] = np.clip(p0[:K], lo[:K] + 0.05, hi[:K] - 0.05)
    kw = dict(tol=(1e-14, 1e-10, 1e-10), maxit=400, method=method)
    fit = amd.nonlinear_fit(data=data, model=d['model'], prior=d['prior'], p0=p0, fitter='mi355x_trf',
                            bounds=(lo, hi), **kw)
    ref = ofit.nonlinear_fit(d['x'], d['ymean'], gu.dense_cov(d['yerr'], N), gu.cosmix_fcn, prior_mean=d['prior'][0],
                             prior_err=d['prior'][1], p0=p0, jac=gu.cosmix_jac, fitter='scipy_least_squares',
                             bounds=(lo, hi), **kw)
    # both converged; at the end the ftol and the xtol test (scipy status 2 / 3 / 4) trip within one
    # evaluation of each other, and which one is met first is a rounding-level tie
    assert fit.stopping_criterion != 0 and ref.stopping_criterion != 0
    assert fit.stopping_criterion == ref.stopping_criterion or {fit.stopping_criterion, ref.stopping_criterion} == {1, 3}
    # 150-300 evaluations with dozens of wall reflections: the sequence is chaotic in the last bits of the inputs (the
    # walls are placed from the device's own free fit: a different summation order there moves the ORACLE's count
    # from ~150 to 268), so the counts are only required to be of the same size; the end point is the parity check
    assert 0.5 < fit.nit / ref.nit < 2.0, (fit.nit, ref.nit)
    # ftol = 1e-10 ends a run where chi2 still moves by 1e-10 of itself: end points a few 1e-5 standard deviations apart
    # (1e-5 of a sigma until the model's sincos changed its last-bit rounding in round 5, 3e-5 since); the north_star
    # tolerance on the parameters themselves is 1e-6 relative -- met with three orders to spare
    assert np.max(np.abs(fit.pmean - ref.pmean) / free.psdev) < 1e-4
    assert gu.relmax(fit.pmean, ref.pmean) < 1e-8
    assert abs(fit.chi2 / ref.chi2 - 1) < 1e-9
    assert gu.relmax(fit.cov, ref.cov) < 1e-6
    assert 0 < fit.chi2 - free.chi2 < 50


@pytest.mark.parametrize('x_scale', [1.0, 'jac'])
@pytest.mark.parametrize('start', [1.0, 1.3])
@pytest.mark.parametrize('seed,K', [(1, 1), (2, 2), (4, 2)])
def test_minpack_lm_matches_oracle(amd, seed, K, start, x_scale):
    """method='lm' (MINPACK's lmder through scipy, src/lsqfit/_scipy.py:64-67) on the device against
    oracle/minpack.py (pinned on scipy): fit point, chi2, covariance, stopping criterion and the
    evaluation count."""
    x, y, ysd, pm, psd, truth, fcn, jac = multiexp_case(seed, K)
    p0 = start * pm * (1.0 + 0.2 * np.cos(np.arange(2 * K) + seed))
    kw = dict(tol=(1e-8, 1e-8, 1e-8), maxit=300, x_scale=x_scale, method='lm')
    ref = ofit.nonlinear_fit(x, y, ysd, fcn, prior_mean=pm, prior_err=psd, p0=p0, jac=jac,
                             fitter='scipy_least_squares', **kw)
    fit = amd.nonlinear_fit(data=(x, y, ysd), model=amd.multiexp(K), prior=(pm, psd), p0=p0,
                            fitter='mi355x_trf', **kw)
    assert fit.error is None and fit.description == 'method = lm'
    assert gu.relmax(fit.pmean, ref.pmean) < 1e-6
    assert abs(fit.chi2 / ref.chi2 - 1) < 1e-6
    assert gu.relmax(fit.cov, ref.cov) < 1e-6
    assert fit.stopping_criterion == ref.stopping_criterion
    assert abs(fit.nit - ref.nit) <= 1, (fit.nit, ref.nit)


@pytest.mark.parametrize('name', ['misra1a', 'chwirut2', 'danwood', 'rat42', 'thurber', 'boxbod'])
def test_nist_minpack_lm(amd, name):
    pr = nist_problem(name, NIST)
    model = amd.expr(pr['expr'], ['b%d' % (i + 1) for i in range(pr['P'])], xnames=tuple(pr['columns'][1:]))
    x = np.column_stack([pr['x'][c] for c in pr['columns'][1:]])
    kw = dict(tol=(1e-10, 1e-10, 1e-10), maxit=2000, method='lm')
    fit = amd.nonlinear_fit(data=(x, pr['y'], pr['ysd']), model=model, prior=(pr['prior_mean'], pr['prior_sd']),
                            p0=pr['p0'], fitter='mi355x_trf', **kw)
    ref = ofit.nonlinear_fit(pr['x'], pr['y'], pr['ysd'], pr['fcn'], prior_mean=pr['prior_mean'],
                             prior_err=pr['prior_sd'], p0=pr['p0'], fitter='scipy_least_squares', **kw)
    assert gu.relmax(fit.pmean, ref.pmean) < 1e-6
    assert abs(fit.chi2 / ref.chi2 - 1) < 1e-6
    assert gu.relmax(fit.cov, ref.cov) < 1e-5
    assert gu.relmax(fit.pmean, np.asarray(pr['certified'])) < 1e-4
