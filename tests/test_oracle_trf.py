"""-m "not gpu": the TRF restatement (oracle/trf.py) pinned on scipy itself (same iterates) and on
the reference's own scipy-plugin tests (tests/test_lsqfit.py:1754-1808,:1811-1838)."""
import numpy as np
import pytest

from oracle import fit as ofit
from oracle import gvar_lite, trf

scipy_opt = pytest.importorskip('scipy.optimize')


def exp_problem(seed, m=40, k=3, noise=0.01):
    rng = np.random.default_rng(seed)
    t = np.linspace(0.0, 3.0, m)
    a = rng.uniform(0.5, 1.5, k)
    e = np.arange(1, k + 1) * 0.7 + rng.uniform(-0.1, 0.1, k)
    y = (a[:, None] * np.exp(-e[:, None] * t)).sum(0) + noise * rng.standard_normal(m)

    def fun(p):
        return (p[:k, None] * np.exp(-p[k:, None] * t)).sum(0) - y

    def jac(p):
        ex = np.exp(-p[k:, None] * t)
        return np.concatenate([ex, -p[:k, None] * t * ex]).T
    return fun, jac, np.concatenate([a, e])


CASES = []
for seed in range(6):
    fun, jac, truth = exp_problem(100 + seed, k=1 + seed % 3)
    n = truth.size
    x0 = truth * (1.0 + 0.3 * np.cos(np.arange(n) + seed))
    lo = np.minimum(truth, x0) - 0.5
    hi = np.maximum(truth, x0) + 0.5
    CASES.append(('free%d' % seed, fun, jac, x0, None))
    CASES.append(('loose%d' % seed, fun, jac, x0, (lo, hi)))
    # bounds that cut the optimum off: some parameters end on a bound
    hi2 = hi.copy()
    hi2[0] = 0.5 * (truth[0] + x0[0]) if x0[0] < truth[0] else hi[0]
    lo2 = lo.copy()
    lo2[-1] = 0.5 * (truth[-1] + x0[-1]) if x0[-1] > truth[-1] else lo[-1]
    CASES.append(('active%d' % seed, fun, jac, x0, (lo2, hi2)))
    CASES.append(('semi%d' % seed, fun, jac, x0, (lo2, np.full(n, np.inf))))


@pytest.mark.parametrize('method', ['trf', 'dogbox'])
@pytest.mark.parametrize('x_scale', [1.0, 'jac'])
@pytest.mark.parametrize('name,fun,jac,x0,bounds', CASES, ids=[c[0] for c in CASES])
def test_same_iterates_as_scipy(name, fun, jac, x0, bounds, x_scale, method):
    kw = dict(xtol=1e-8, gtol=1e-8, ftol=1e-8, max_nfev=200, x_scale=x_scale)   # stops above the rounding floor
    ref = scipy_opt.least_squares(fun, x0, jac=jac, method=method,
                                  bounds=(-np.inf, np.inf) if bounds is None else bounds, **kw)
    got = getattr(trf, method)(fun, jac, x0, bounds=bounds, **kw)
    assert got.nfev == ref.nfev and got.njev == ref.njev and got.status == ref.status
    np.testing.assert_allclose(got.x, ref.x, rtol=1e-9, atol=1e-12)
    assert abs(got.cost - ref.cost) <= 1e-9 * ref.cost
    assert abs(got.optimality - ref.optimality) <= 1e-4 * ref.optimality + 1e-12
    if method == 'dogbox':
        assert np.array_equal(got.active_mask, ref.active_mask)


def test_max_nfev_and_start_on_bound():
    fun, jac, truth = exp_problem(7, k=2)
    lo, hi = truth - 1.0, truth + 1.0
    x0 = truth.copy()
    x0[0] = lo[0]                       # on the bound: moved 1e-10 inside
    x0[1] = hi[1]
    for nmax in (1, 3, 50):
        for method in ('trf', 'dogbox'):
            ref = scipy_opt.least_squares(fun, x0, jac=jac, bounds=(lo, hi), max_nfev=nmax, method=method)
            got = getattr(trf, method)(fun, jac, x0, bounds=(lo, hi), max_nfev=nmax)
            assert (got.nfev, got.status) == (ref.nfev, ref.status)
            np.testing.assert_allclose(got.x, ref.x, rtol=1e-9)
    with pytest.raises(ValueError, match='outside'):
        trf.trf(fun, jac, truth + 5.0, bounds=(lo, hi))
    with pytest.raises(ValueError, match='strictly less'):
        trf.trf(fun, jac, truth, bounds=(hi, lo))


def test_reference_scipy_least_squares_case():
    """tests/test_lsqfit.py:1754-1767: f = (x - xans)^2 + (x - xans)^4, method trf stops on gtol."""
    xans = np.arange(3) + 1.0
    f = lambda x: (x - xans) ** 2 + (x - xans) ** 4
    df = lambda x: np.diag(2 * (x - xans) + 4 * (x - xans) ** 3)
    ans = trf.scipy_least_squares(np.ones(3), 3, f, df, tol=(1e-15, 1e-8, 1e-15), method='trf')
    np.testing.assert_allclose(ans.x, xans, rtol=1e-3)
    assert ans.stopping_criterion == 2
    assert ans.description == 'method = trf'
    ans = trf.scipy_least_squares(np.zeros(3), 3, f, df, tol=(1e-15, 1e-8, 1e-15), method='dogbox')   # :1771-1775
    np.testing.assert_allclose(ans.x, xans, rtol=1e-3)
    assert ans.stopping_criterion == 2
    assert ans.description == 'method = dogbox'


def test_reference_bounds_case():
    """tests/test_lsqfit.py:1780-1808: data 0.9(1), 2.2(2), fcn(p) = p, bounds [0, 0.5] x [0, 1]:
    the fit ends on the upper bounds (assertAlmostEqual, 7 places)."""
    ym, ys = gvar_lite.parse_array(['0.9(1)', '2.2(2)'])
    fit = ofit.nonlinear_fit(False, ym, ys, lambda p: p, p0=[0.25, 0.5], jac=lambda p: np.eye(2),
                             fitter='scipy_least_squares', bounds=([0.0, 0.0], [0.5, 1.0]))
    assert abs(fit.pmean[0] - 0.5) < 5e-8 and abs(fit.pmean[1] - 1.0) < 5e-8


def test_reference_fitters_case():
    """tests/test_lsqfit.py:1811-1838: every fitter prints fit.p = [0.904(98) 2.17(19)]."""
    ym, ys = gvar_lite.parse_array(['0.9(1)', '2.2(2)'])
    pm, ps = gvar_lite.parse_array(['1.0(5)', '2.0(5)'])
    for method in ('trf', 'dogbox'):
        fit = ofit.nonlinear_fit(False, ym, ys, lambda p: p, prior_mean=pm, prior_err=ps, jac=lambda p: np.eye(2),
                                 fitter='scipy_least_squares', method=method)
        assert gvar_lite.fmt_array(fit.pmean, fit.psdev) == '[0.904(98) 2.17(19)]'


# ---- method='lm': MINPACK's lmder (oracle/minpack.py) ----------------------------------------------
from oracle import minpack   # noqa: E402

# (the three-exponential problems collapse onto nearly equal exponents with amplitudes of +-500:
# the counts still agree there, the end points only to a few per cent, so they pin nothing)
LM_CASES = [(c[0] + s, c[1], c[2], c[3] * sc) for c in CASES if c[4] is None and c[3].size < 6
            for s, sc in (('', 1.0), ('_far', 1.6), ('_near', 1.02))]


@pytest.mark.parametrize('x_scale', [1.0, 'jac'])
@pytest.mark.parametrize('name,fun,jac,x0', LM_CASES, ids=[c[0] for c in LM_CASES])
def test_minpack_lm_same_iterates_as_scipy(name, fun, jac, x0, x_scale):
    kw = dict(xtol=1e-8, gtol=1e-8, ftol=1e-8, max_nfev=300, x_scale=x_scale)
    ref = scipy_opt.least_squares(fun, x0, jac=jac, method='lm', **kw)
    got = minpack.least_squares_lm(fun, jac, x0, **kw)
    assert (got.nfev, got.njev, got.status) == (ref.nfev, ref.njev, ref.status)
    # (the restatement goes through J^T J, MINPACK through a pivoted QR: the three-exponential
    # problems are ill-conditioned enough to show that at the 1e-8 level)
    np.testing.assert_allclose(got.x, ref.x, rtol=1e-7, atol=1e-12)
    assert abs(got.cost - ref.cost) <= 1e-9 * ref.cost


def test_minpack_lm_reference_case_and_limits():
    """tests/test_lsqfit.py:1768-1772: method 'lm', tol (1e-8, 1e-15, 1e-15) -> criterion 1 (xtol)."""
    xans = np.arange(3) + 1.0
    f = lambda x: (x - xans) ** 2 + (x - xans) ** 4
    df = lambda x: np.diag(2 * (x - xans) + 4 * (x - xans) ** 3)
    ref = scipy_opt.least_squares(f, np.zeros(3), jac=df, method='lm', xtol=1e-8, gtol=1e-15, ftol=1e-15)
    got = minpack.least_squares_lm(f, df, np.zeros(3), xtol=1e-8, gtol=1e-15, ftol=1e-15)
    assert (got.nfev, got.status) == (ref.nfev, ref.status) and ref.status == 3
    np.testing.assert_allclose(got.x, ref.x, rtol=1e-9)
    ans = trf.scipy_least_squares(np.zeros(3), 3, f, df, tol=(1e-8, 1e-15, 1e-15), method='lm')
    np.testing.assert_allclose(ans.x, xans, rtol=1e-3)
    assert ans.stopping_criterion == 1 and ans.description == 'method = lm'
    for nmax in (1, 2, 7):
        ref = scipy_opt.least_squares(f, np.zeros(3), jac=df, method='lm', max_nfev=nmax)
        got = minpack.least_squares_lm(f, df, np.zeros(3), max_nfev=nmax)
        assert (got.nfev, got.status) == (ref.nfev, ref.status)
    with pytest.raises(ValueError, match='machine epsilon'):
        minpack.least_squares_lm(f, df, np.zeros(3), xtol=1e-17)


def outlier_problem(seed, k=2, m=60):
    """exp_problem with a tenth of the points knocked far off: what a robust loss is for."""
    fun0, jac, truth = exp_problem(300 + seed, m=m, k=k)
    rng = np.random.default_rng(seed)
    bad = rng.choice(m, m // 10, replace=False)
    shift = np.zeros(m)
    shift[bad] = rng.choice([-1.0, 1.0], bad.size) * rng.uniform(0.2, 0.6, bad.size)

    def fun(p):
        return (fun0(p) - shift) / 0.02          # residuals in units of a 0.02 error bar: outliers at 10 .. 30 sigma
    return fun, (lambda p: jac(p) / 0.02), truth


@pytest.mark.parametrize('method', ['trf', 'dogbox'])
@pytest.mark.parametrize('loss', ['soft_l1', 'huber', 'cauchy', 'arctan'])
@pytest.mark.parametrize('f_scale', [1.0, 2.5])
@pytest.mark.parametrize('seed', range(3))
def test_robust_losses_same_iterates_as_scipy(seed, f_scale, loss, method):
    """loss / f_scale (src/lsqfit/_scipy.py:76-79 documents them, :147-153 forwards them): the restated losses and
    scale_for_robust_loss_function against scipy itself, with an array x_scale and with bounds."""
    fun, jac, truth = outlier_problem(seed, k=1 + seed % 2)
    n = truth.size
    x0 = truth * (1.0 + 0.2 * np.cos(np.arange(n) + seed))
    bounds = None if seed != 1 else (np.minimum(truth, x0) - 0.3, np.maximum(truth, x0) + 0.3)
    x_scale = 1.0 if seed == 0 else np.linspace(0.5, 2.0, n)
    kw = dict(xtol=1e-8, gtol=1e-8, ftol=1e-8, max_nfev=300, x_scale=x_scale, loss=loss, f_scale=f_scale)
    ref = scipy_opt.least_squares(fun, x0, jac=jac, method=method, bounds=(-np.inf, np.inf) if bounds is None else bounds, **kw)
    got = getattr(trf, method)(fun, jac, x0, bounds=bounds, **kw)
    assert got.nfev == ref.nfev and got.njev == ref.njev and got.status == ref.status
    # (two look-alike exponentials under a non-convex loss: the SVD-based sub-problem amplifies rounding differences to
    #  ~1e-8 of x over a few dozen iterations; the iteration counts above are identical)
    np.testing.assert_allclose(got.x, ref.x, rtol=2e-7, atol=1e-12)
    assert abs(got.cost - ref.cost) <= 1e-9 * ref.cost
    np.testing.assert_allclose(got.fun, ref.fun, rtol=1e-6, atol=1e-6)        # the TRUE residuals ...
    np.testing.assert_allclose(got.jac, ref.jac, rtol=1e-6, atol=1e-6)        # ... and the SCALED Jacobian (the plugin's covariance, :165-169)
    # the outliers do not drag the fit: closer to the truth than the plain least-squares answer
    plain = scipy_opt.least_squares(fun, x0, jac=jac, method=method)
    assert np.linalg.norm(got.x - truth) < np.linalg.norm(plain.x - truth)


def test_lm_takes_only_the_linear_loss():
    fun, jac, truth = outlier_problem(0)
    with pytest.raises(ValueError):
        trf.scipy_least_squares(truth, 60, fun, jac, method='lm', loss='huber')
    with pytest.raises(ValueError):
        trf.trf(fun, jac, truth, loss='tukey')
