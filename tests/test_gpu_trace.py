"""-m gpu: fits whose fit function is a PYTHON callable, traced into the device tape (lsqfit_amd.trace) -- the reference's
call form ``nonlinear_fit(data=(x, y), fcn=fcn, prior=prior)`` (src/lsqfit/__init__.py:1997-2042, _gsl.pyx:742-760) --
against the formula-string front end (bit for bit), the reference's golden values and the oracle."""
import numpy as np
import pytest

from oracle import fit as ofit
from oracle import dual
from oracle.dual import Dual
from tests import gpu_util as gu
from tests.helpers import load, nist_problem
from tests.nist_lambdas import MODELS

pytestmark = pytest.mark.gpu
NIST = load('nist.json')


@pytest.fixture(scope='module')
def amd():
    import lsqfit_amd
    from lsqfit_amd import _lib
    _lib.load()
    return lsqfit_amd


@pytest.mark.parametrize('name', sorted(NIST))
def test_nist_numpy_function_fits_like_the_formula_string(amd, name):
    pr = nist_problem(name, NIST)
    cols = pr['columns'][1:]
    x = pr['x'] if len(cols) > 1 else pr['x'][cols[0]]
    kw = dict(prior=(pr['prior_mean'], pr['prior_sd']), p0=pr['p0'], tol=pr['tol'], solver='qr')
    ref = amd.nonlinear_fit(data=(np.stack([pr['x'][c] for c in cols], axis=1), pr['y'], pr['ysd']),
                            model=amd.expr(pr['expr'], ['b%d' % (i + 1) for i in range(pr['P'])], cols), **kw)
    # operation for operation the string's tape (fold=False): the same compiled code, the same bits
    tr = amd.trace(MODELS[name], x, pr['p0'], fold=False)
    same = amd.nonlinear_fit(data=(tr.x, pr['y'], pr['ysd']), model=tr.model, **kw)
    assert np.array_equal(same.pmean, ref.pmean) and np.array_equal(same.cov, ref.cov)
    assert same.chi2 == ref.chi2 and same.nit == ref.nit and same.logGBF == ref.logGBF
    # the call a user of the reference makes: fcn=, nothing else (parameter-free arithmetic folded into predictor columns)
    fit = amd.nonlinear_fit(data=(x, pr['y'], pr['ysd']), fcn=MODELS[name], **kw)
    assert np.all(np.abs(fit.pmean - pr['certified']) <= 1e-2 * pr['certified_sd'] + 1e-9 * np.abs(pr['certified']))
    np.testing.assert_allclose(fit.psdev, pr['certified_sd'], rtol=2e-3)
    assert np.all(np.abs(fit.pmean - ref.pmean) <= 1e-6 * np.abs(ref.pmean) + 1e-4 * ref.psdev)
    assert fit.chi2 == pytest.approx(ref.chi2, rel=2e-3 if name == 'lanczos1' else 1e-6)
    assert fit.dof == pr['out']['dof'] and fit.stopping_criterion == 1


def test_dictionary_parameter_multiexponential_matches_oracle(amd):
    """p['a'], p['E'] on correlated data with a correlated prior block: the canonical lsqfit fit (examples/y-vs-x.py)"""
    rng = np.random.default_rng(20265)
    K, N = 3, 40
    x = np.linspace(0.2, 4.0, N)
    ptrue = dict(a=np.array([1.0, 0.6, 0.3]), E=np.array([0.5, 1.1, 1.9]))

    def fcn(x, p):
        return np.sum(p['a'][:, None] * np.exp(-p['E'][:, None] * x[None, :]), axis=0)

    def flat_fcn(x, p):           # the oracle's view: flat parameters, Dual-capable
        a, E = p[:K], p[K:]
        if isinstance(p, Dual):
            return dual.stack_sum([a[k] * dual.exp(-(E[k] * x)) for k in range(K)])
        return np.exp(-np.outer(x, E)) @ a

    f0 = fcn(x, ptrue)
    sd = 0.01 * np.abs(f0)
    U = rng.uniform(0.1, 0.9, (N, 2 * N))
    corr = U @ U.T
    corr /= np.sqrt(np.outer(np.diag(corr), np.diag(corr)))
    ycov = corr * np.outer(sd, sd)
    y = f0 + np.linalg.cholesky(ycov) @ rng.standard_normal(N)
    pa_cov = np.diag([0.5, 0.5, 0.5]) ** 2
    pa_cov[0, 1] = pa_cov[1, 0] = 0.3 * 0.25
    prior = (dict(a=np.array([1.0, 0.5, 0.5]), E=np.array([0.5, 1.0, 2.0])), dict(a=pa_cov, E=np.array([0.2, 0.3, 0.4])))
    fit = amd.nonlinear_fit(data=(x, y, ycov), fcn=fcn, prior=prior, tol=1e-10)
    pm = np.concatenate([prior[0]['a'], prior[0]['E']])
    pcov = np.zeros((2 * K, 2 * K))
    pcov[:K, :K] = pa_cov
    pcov[K:, K:] = np.diag(prior[1]['E'] ** 2)
    ref = ofit.nonlinear_fit(x, y, ycov, flat_fcn, prior_mean=pm, prior_err=pcov, tol=1e-10, solver='cholesky')
    assert gu.relmax(fit.pmean, ref.pmean) < 1e-6
    assert fit.chi2 / fit.dof == pytest.approx(ref.chi2 / ref.dof, rel=1e-6)
    assert gu.relmax(fit.cov, ref.cov) < 1e-6
    assert fit.logGBF == pytest.approx(ref.logGBF, rel=1e-8, abs=1e-6)
    assert set(fit.p) == {'a', 'E'} and fit.p['E'].shape == (K,) and np.array_equal(fit.p['E'], fit.pmean[K:])


def test_dictionary_data_and_output(amd):
    """examples/simple.py's shape: y is a dictionary with a covariance for one entry, the function returns a dictionary"""
    xa = np.array([1., 2., 3., 4.])
    ptrue = np.array([0.25, 0.5])
    ya = np.exp(ptrue[0] + xa * ptrue[1]) * np.array([1.01, 0.98, 1.02, 0.99])
    cov_a = np.diag((0.05 * ya) ** 2)
    cov_a[0, 1] = cov_a[1, 0] = 0.5 * 0.05 * ya[0] * 0.05 * ya[1]
    y = dict(a=ya, b=np.array(2.05))
    yerr = dict(a=cov_a, b=np.array(0.1))

    def fcn(x, p):
        return dict(a=np.exp(p[0] + x['a'] * p[1]), b=p[1] / p[0])

    fit = amd.nonlinear_fit(data=(dict(a=xa), y, yerr), fcn=fcn, prior=(np.array([0.3, 0.4]), np.array([0.5, 0.5])), tol=1e-10)
    full = np.zeros((5, 5))
    full[:4, :4] = cov_a
    full[4, 4] = 0.01

    def flat_fcn(x, p):
        if isinstance(p, Dual):
            return dual.concatenate([dual.exp(p[0] + p[1] * xa), (p[1] / p[0]).reshape(1)])
        return np.concatenate([np.exp(p[0] + xa * p[1]), [p[1] / p[0]]])

    ref = ofit.nonlinear_fit(None, np.concatenate([ya, [2.05]]), full, flat_fcn, prior_mean=np.array([0.3, 0.4]),
                             prior_err=np.array([0.5, 0.5]), tol=1e-10, solver='cholesky')
    assert gu.relmax(fit.pmean, ref.pmean) < 1e-6 and gu.relmax(fit.cov, ref.cov) < 1e-6
    assert fit.chi2 == pytest.approx(ref.chi2, rel=1e-6) and fit.nblocks == ref.nblocks


def test_control_flow_on_parameters_is_refused(amd):
    x = np.linspace(0, 1, 6)
    with pytest.raises(amd.TraceError, match='control flow'):
        amd.nonlinear_fit(data=(x, x, 0.1 * np.ones(6)), fcn=lambda x, p: p[0] * x if p[0] > 0 else p[1], p0=np.ones(2))


def test_plugin_called_the_way_lsqfit_calls_it(amd):
    """``FITTERS[name](p0, nf, chiv, tol=, maxit=)`` (src/lsqfit/__init__.py:662-664) with NOTHING else: the plugin records the
    residual function it is handed and fits on the device; same minimum, covariance and chi2 as the fit set up from parts"""
    rng = np.random.default_rng(77)
    x = np.linspace(0.2, 3.0, 30)
    y = 1.4 * np.exp(-0.7 * x) + 0.3 + 0.01 * rng.standard_normal(30)
    sd = np.full(30, 0.01)
    pm, psd = np.array([1.0, 1.0, 0.0]), np.array([1.0, 1.0, 1.0])
    mean = np.concatenate([y, pm])
    w = 1.0 / np.concatenate([sd, psd])

    def chiv(p, mixed=False):
        delta = np.concatenate(((p[0] * np.exp(-p[1] * x) + p[2]).flat, p)) - mean
        ans = np.zeros(33, object if mixed else float)
        ans[:] = np.multiply(w, delta)
        return ans

    got = amd.mi355x_lm(pm.copy(), 33, chiv, tol=(1e-10, 1e-10, 1e-10), maxit=200)
    ref = amd.nonlinear_fit(data=(x, y, sd), model=amd.expr('a*exp(-b*x)+c', ['a', 'b', 'c']), prior=(pm, psd), p0=pm, tol=1e-10)
    assert gu.relmax(got.x, ref.pmean) < 1e-8 and gu.relmax(got.cov, ref.cov) < 1e-7
    assert got.chi2 == pytest.approx(ref.chi2, rel=1e-8)
    assert got.f.shape == (33,) and got.J.shape == (33, 3)
    np.testing.assert_allclose(got.f, chiv(got.x), rtol=1e-9, atol=1e-12)
    assert got.stopping_criterion in (1, 2) and got.error is None


def test_bounded_fit_with_dictionary_parameters(amd):
    """fitter='mi355x_trf' with bounds given in the parameters' own shape (dictionaries, src/lsqfit/__init__.py:641-655) on a traced
    function.  Against the oracle's restated scipy method at 1e-6."""
    rng = np.random.default_rng(31)
    x = np.linspace(0.1, 3.0, 60)

    def fcn(x, p):
        return p['a'][0] * np.exp(-p['E'][0] * x) + p['a'][1] * np.exp(-p['E'][1] * x) + np.where(x > 2.0, p['c'], 0.0)

    def flat(x, p):
        step = (x > 2.0).astype(float)
        if isinstance(p, Dual):
            return p[0] * dual.exp(-(p[2] * x)) + p[1] * dual.exp(-(p[3] * x)) + p[4] * step
        return p[0] * np.exp(-p[2] * x) + p[1] * np.exp(-p[3] * x) + p[4] * step

    truth = np.array([1.0, 0.4, 0.5, 1.6, 0.05])
    sd = 0.01 * np.ones(60)
    y = flat(x, truth) + sd * rng.standard_normal(60)
    p0 = dict(a=np.array([0.8, 0.3]), E=np.array([0.4, 1.2]), c=0.0)
    lower = dict(a=np.array([0.0, 0.0]), E=np.array([0.0, 0.0]), c=-1.0)
    upper = dict(a=np.array([0.9, 5.0]), E=np.array([5.0, 5.0]), c=1.0)
    fit = amd.nonlinear_fit(data=(x, y, sd), fcn=fcn, p0=p0, fitter='mi355x_trf', bounds=(lower, upper), tol=1e-10)
    lo = np.array([0.0, 0.0, 0.0, 0.0, -1.0])
    hi = np.array([0.9, 5.0, 5.0, 5.0, 1.0])
    ref = ofit.nonlinear_fit(x, y, sd, flat, p0=np.array([0.8, 0.3, 0.4, 1.2, 0.0]), tol=1e-10, fitter='scipy_least_squares',
                             bounds=(lo, hi))
    assert fit.model.programs is None          # the step at x > 2 makes two formulas (39 + 21 rows): a fit this small records them as ONE (trace.merge_small_programs)
    assert gu.relmax(fit.pmean, ref.pmean) < 1e-6 and fit.chi2 == pytest.approx(ref.chi2, rel=1e-6)
    assert np.all(fit.pmean >= lo) and np.all(fit.pmean <= hi) and fit.p['c'].shape == () and fit.p['a'].shape == (2,)


@pytest.mark.parametrize('kind', ['lognormal', 'sqrtnormal'])
def test_reference_distribution_priors_through_a_python_function(amd, kind):
    """tests/test_lsqfit.py:1594-1640 as the reference writes them -- ``prior['log(a)']`` / ``prior['sqrt(a)']``, a fit function of
    ``p['a']`` -- traced and fitted on the device: the printed ``fit.p['a']`` strings '0.012(11)' and '0.010(13)'."""
    from oracle import gvar_lite
    from tests.test_oracle_kat import NORMAL_Y, TRANSFORMED_PRIOR_CASES
    a_of, um, us, da, want = TRANSFORMED_PRIOR_CASES[kind]
    ym, ys = gvar_lite.parse_array(NORMAL_Y)
    key = {'lognormal': 'log(a)', 'sqrtnormal': 'sqrt(a)'}[kind]

    def fcn(p, N=ym.size):
        return N * [p['a']]

    fit = amd.nonlinear_fit(data=(ym, ys), fcn=fcn, prior=({key: um}, {key: us}))
    u = fit.pmean[0]
    assert fit.p['a'] == pytest.approx(float(a_of(u)), rel=1e-14)
    assert gvar_lite.fmt(float(fit.p['a']), abs(da(u)) * fit.psdev[0]) == want


def test_fit_function_written_with_products_norms_and_einsum(amd):
    """the wider numpy vocabulary of the tracer (prod, linalg.norm, einsum, arctan2, hypot, subtract.outer) in ONE fit, against the
    oracle differentiating an independently written version of the same function with its dual numbers"""
    rng = np.random.default_rng(20266)
    N = 60
    x = np.linspace(0.1, 3.0, N)
    A = rng.uniform(0.2, 1.0, (N, 2))
    ptrue = np.array([0.8, 1.4, 0.7, 0.5, 0.3, 0.9])

    def fcn(x, p):
        amp = np.prod(p[2:4])
        rate = np.linalg.norm(p[2:4])
        bumps = np.exp(-np.subtract.outer(x, p[:2]) ** 2).sum(axis=1)
        return np.arctan2(p[0] * x, 1.0 + x) + amp * np.exp(-rate * x) + np.einsum('ij,j->i', A, p[4:6]) + 0.1 * np.hypot(p[5], x) + 0.05 * bumps

    def flat_fcn(x, p):
        if isinstance(p, Dual):
            amp, rate = p[2] * p[3], dual.sqrt(p[2] * p[2] + p[3] * p[3])
            t = p[0] * x
            r = dual.sqrt(t * t + (1.0 + x) * (1.0 + x))
            at2 = 2.0 * dual.arctan(t / (r + (1.0 + x)))        # (the abscissa is data: arctan2's branch is known row by row)
            lin = p[4] * A[:, 0] + p[5] * A[:, 1]
            bumps = dual.exp(-((x - p[0]) * (x - p[0]))) + dual.exp(-((x - p[1]) * (x - p[1])))
            return at2 + amp * dual.exp(-(rate * x)) + lin + 0.1 * dual.sqrt(p[5] * p[5] + x * x) + 0.05 * bumps
        return fcn(x, p)

    f0 = fcn(x, ptrue)
    sd = 0.02 * (0.2 + np.abs(f0))
    y = f0 + sd * rng.standard_normal(N)
    prior = (np.array([1.0, 1.0, 0.5, 0.5, 0.5, 0.5]), np.full(6, 0.6))
    fit = amd.nonlinear_fit(data=(x, y, sd), fcn=fcn, prior=prior, tol=1e-10)
    ref = ofit.nonlinear_fit(x, y, sd, flat_fcn, prior_mean=prior[0], prior_err=prior[1], tol=1e-10, solver='cholesky')
    assert gu.relmax(fit.pmean, ref.pmean) < 1e-6
    assert fit.chi2 / fit.dof == pytest.approx(ref.chi2 / ref.dof, rel=1e-6)
    assert gu.relmax(fit.cov, ref.cov) < 1e-6
    assert fit.logGBF == pytest.approx(ref.logGBF, rel=1e-8, abs=1e-6)
    assert fit.chi2 / fit.dof < 2.0


def test_empbayes_example_written_the_reference_way(amd):
    """examples/empbayes.py:22-36 as it stands there -- ``fitargs(z)`` returns ``dict(data=(x, y), fcn=fcn, prior=prior)`` with a
    Python fit function -- through lsqfit_amd.empbayes_fit: the function is recorded once, the candidate fits of the z search
    run as lockstep batches, the answer is examples/empbayes.out's (logGBF 21.274 at prior width 5.3).  Then the same with
    dictionary parameters."""
    from tests.helpers import load
    from oracle import gvar_lite
    k = load('kat.json')['empbayes']
    x = np.array(k['src_inputs']['x'])
    ym, ys = gvar_lite.parse_array(k['src_inputs']['y'])

    def fcn(x, p):
        return np.exp(-p[0] - p[1] * x - p[2] * x ** 2 - p[3] * x ** 3)

    def fitargs(z):
        return dict(data=(x, ym, ys), fcn=fcn, prior=(np.zeros(4), np.full(4, abs(z))))

    from lsqfit_amd import sweep
    calls = []
    orig = sweep.EvidenceSurface._batch

    def spy(self, items):
        r = orig(self, items)
        calls.append(r is not None)
        return r
    sweep.EvidenceSurface._batch = spy
    try:
        fit, z = amd.empbayes_fit(1.0, fitargs)
    finally:
        sweep.EvidenceSurface._batch = orig
    assert calls and all(calls)                      # every simplex move was one lockstep batch
    assert '%.1f' % abs(z) == '5.3'
    assert '%.5g' % fit.logGBF == '21.274'
    assert gvar_lite.fmt_array(fit.pmean, fit.psdev) == '[2.5904(22) -6.530(22) 7.832(65) -1.688(55)]'
    assert fit.traced is not None and fit.p.shape == (4,)

    def fcn_d(x, p):
        c = p['c']
        return np.exp(-p['c0'] - c[0] * x - c[1] * x ** 2 - c[2] * x ** 3)

    def fitargs_d(z):
        return dict(data=(x, ym, ys), fcn=fcn_d, prior=(dict(c0=0.0, c=np.zeros(3)), dict(c0=abs(z), c=np.full(3, abs(z)))))
    fit_d, z_d = amd.empbayes_fit(1.0, fitargs_d)
    assert '%.1f' % abs(z_d) == '5.3' and '%.5g' % fit_d.logGBF == '21.274'
    assert set(fit_d.p) == {'c0', 'c'} and np.allclose(np.concatenate([[fit_d.p['c0']], fit_d.p['c']]), fit.pmean, rtol=1e-7)


def test_resampled_copies_of_a_traced_fit(amd):
    """simulated / bootstrapped copies (src/lsqfit/__init__.py:1391-1642) of a fit whose function was a Python callable: the
    same copies, bit for bit, as those of the fit given the formula string (the recording made with fold=False IS the string's
    tape), and a dictionary-parameter fit resamples in the batch engine too"""
    rng = np.random.default_rng(5)
    x = np.linspace(0.1, 3.0, 48)
    y = 1.7 * np.exp(-0.8 * x) + 0.3 + 0.02 * rng.standard_normal(x.size)
    sd = np.full(x.size, 0.02)
    prior = (np.array([1.0, 1.0, 0.0]), np.array([2.0, 2.0, 1.0]))
    m = amd.expr('a*exp(-b*x)+c', ['a', 'b', 'c'])
    f_str = amd.nonlinear_fit(data=(x, y, sd), model=m, prior=prior)
    tr = amd.trace(lambda x, p: p[0] * np.exp(-p[1] * x) + p[2], x, prior[0], fold=False)
    f_tr = amd.nonlinear_fit(data=(tr.x, y, sd), model=tr.model, prior=prior)
    assert np.array_equal(f_str.pmean, f_tr.pmean)
    a, b = f_str.simulated_fits(5, seed=11), f_tr.simulated_fits(5, seed=11)
    assert a.engine == b.engine == 'batched' and np.array_equal(a.pmean, b.pmean) and np.array_equal(a.chi2, b.chi2)
    a, b = f_str.bootstrapped_fits(5, seed=12), f_tr.bootstrapped_fits(5, seed=12)
    assert np.array_equal(a.pmean, b.pmean)

    def fcn(x, p):
        return p['amp'] * np.exp(-p['rate'] * x) + p['off']
    pd = (dict(amp=1.0, rate=1.0, off=0.0), dict(amp=2.0, rate=2.0, off=1.0))
    f_d = amd.nonlinear_fit(data=(x, y, sd), fcn=fcn, prior=pd)
    assert np.allclose(f_d.pmean, f_str.pmean, rtol=1e-9, atol=1e-12)
    r = f_d.simulated_fits(8, seed=13)
    assert r.engine == 'batched' and r.pmean.shape == (8, 3)
    assert np.all(np.abs(r.pmean.mean(axis=0) - f_d.pmean) < 4 * f_d.psdev / np.sqrt(8) + 1e-12)


def test_errors_in_variables_with_hundreds_of_points(amd):
    """examples/x-err.py's construction at 300 points: every abscissa is a fit parameter with its own prior (the measured
    x +- its error), row i of the function reads p['x'][i] -- 304 parameters, one formula with indicator columns -- against the
    oracle differentiating the same function with dual numbers"""
    rng = np.random.default_rng(20267)
    n = 300
    xt = np.linspace(2.0, 12.0, n)
    bt = np.array([10.0, 6.0, 1.2, 0.9])

    def curve(b, x, exp=np.exp):
        return b[0] / ((1. + exp(b[1] - b[2] * x)) ** (1. / b[3]))

    xs, ys = 0.05 + 0.01 * xt, 0.02 * curve(bt, xt) + 0.01
    xm = xt + xs * rng.standard_normal(n)
    ym = curve(bt, xt) + ys * rng.standard_normal(n)

    def fcn(p):
        return curve(p['b'], p['x'])

    prior = (dict(b=np.array([9.0, 5.0, 1.0, 1.0]), x=xm), dict(b=np.array([5.0, 5.0, 1.0, 0.5]), x=xs))
    fit = amd.nonlinear_fit(data=(ym, ys), fcn=fcn, prior=prior, tol=1e-10)
    assert fit.traced.model.programs is None and fit.traced.x.shape == (n, n)

    def flat(x, p):
        if isinstance(p, Dual):
            return curve([p[0], p[1], p[2], p[3]], p[4:], exp=dual.exp)
        return curve(p[:4], p[4:])
    pm = np.concatenate([prior[0]['b'], xm])
    pe = np.concatenate([prior[1]['b'], xs])
    ref = ofit.nonlinear_fit(None, ym, ys, flat, prior_mean=pm, prior_err=pe, tol=1e-10, solver='cholesky')
    assert gu.relmax(fit.pmean, ref.pmean) < 1e-6
    assert fit.chi2 / fit.dof == pytest.approx(ref.chi2 / ref.dof, rel=1e-6)
    assert gu.relmax(fit.cov, ref.cov) < 1e-6
    assert fit.logGBF == pytest.approx(ref.logGBF, rel=1e-8, abs=1e-6)
    assert fit.p['x'].shape == (n,) and np.all(np.abs(fit.p['b'] - bt) < 5 * fit.psdev[:4])


def test_interleaved_groups_share_one_formula(amd):
    """p['norm'][group] with the groups interleaved row by row: one formula, two indicator columns; equals the fit of the
    same data sorted by group (two programs)"""
    rng = np.random.default_rng(20268)
    n = 1200
    x = np.linspace(0.0, 2.0, n)
    group = np.arange(n) % 2
    truth = dict(norm=np.array([2.0, 3.0]), E=0.7)
    sd = np.full(n, 0.02)
    y = truth['norm'][group] * np.exp(-truth['E'] * x) + sd * rng.standard_normal(n)
    prior = (dict(norm=np.array([1.0, 1.0]), E=1.0), dict(norm=np.array([5.0, 5.0]), E=2.0))
    fit = amd.nonlinear_fit(data=(x, y, sd), fcn=lambda x, p: p['norm'][group] * np.exp(-p['E'] * x), prior=prior, tol=1e-10)
    assert fit.traced.model.programs is None
    o = np.argsort(group, kind='stable')
    gs = group[o]
    srt = amd.nonlinear_fit(data=(x[o], y[o], sd[o]), fcn=lambda x, p: p['norm'][gs] * np.exp(-p['E'] * x), prior=prior, tol=1e-10)
    assert len(srt.traced.model.programs) == 2
    assert gu.relmax(fit.pmean, srt.pmean) < 1e-9 and gu.relmax(fit.cov, srt.cov) < 1e-8
    assert fit.chi2 == pytest.approx(srt.chi2, rel=1e-10)
