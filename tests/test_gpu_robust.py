"""-m gpu: scipy_least_squares' pass-through options on the device (src/lsqfit/_scipy.py:76-79 names them, :147-153 forwards
them): the robust losses soft_l1 / huber / cauchy / arctan with f_scale, and an array x_scale -- methods trf and dogbox --
against the oracle (oracle/trf.py, pinned on scipy ITSELF for every loss in tests/test_oracle_trf.py) at 1e-6: fit point, the
covariance of the loss-scaled Jacobian (:165-169), chi2 of the TRUE residuals (src/lsqfit/__init__.py:667), logGBF from the
true Jacobian (:719), stopping criterion, evaluation count."""
import numpy as np
import pytest

from oracle import fit as ofit
from tests import gpu_util as gu
from tests.test_gpu_trf import multiexp_case

pytestmark = pytest.mark.gpu
LOSSES = ['soft_l1', 'huber', 'cauchy', 'arctan']


@pytest.fixture(scope='module')
def amd():
    import lsqfit_amd
    from lsqfit_amd import _lib
    _lib.load()
    return lsqfit_amd


def with_outliers(seed, K, N=200):
    x, y, ysd, pm, psd, truth, fcn, jac = multiexp_case(seed, K, N)
    rng = np.random.default_rng(1000 + seed)
    bad = rng.choice(N, N // 12, replace=False)
    y = y.copy()
    y[bad] += rng.choice([-1.0, 1.0], bad.size) * rng.uniform(8.0, 30.0, bad.size) * ysd[bad]
    return x, y, ysd, pm, psd, truth, fcn, jac


def compare(fit, ref, method):
    assert fit.error is None and fit.description == 'method = ' + method
    assert np.all(np.abs(fit.pmean - ref.pmean) <= 1e-6 * np.abs(ref.pmean) + 1e-5 * ref.psdev), (fit.pmean, ref.pmean)
    assert abs(fit.chi2 / ref.chi2 - 1) < 1e-6
    assert gu.relmax(fit.cov, ref.cov) < 1e-6
    assert fit.logGBF == pytest.approx(ref.logGBF, rel=1e-6, abs=1e-6)
    assert fit.stopping_criterion == ref.stopping_criterion
    assert abs(fit.nit - ref.nit) <= max(1, ref.nit // 10), (fit.nit, ref.nit)


@pytest.mark.parametrize('method', ['trf', 'dogbox'])
@pytest.mark.parametrize('loss', LOSSES)
@pytest.mark.parametrize('seed,K,f_scale', [(1, 1, 1.0), (2, 2, 2.5), (3, 2, 1.0)])
def test_robust_losses_match_the_oracle(amd, seed, K, f_scale, loss, method):
    x, y, ysd, pm, psd, truth, fcn, jac = with_outliers(seed, K)
    p0 = pm * (1.0 + 0.1 * np.cos(np.arange(2 * K) + seed))
    b = None if seed != 3 else (np.minimum(truth, p0) - 0.4, np.maximum(truth, p0) + 0.4)
    kw = dict(tol=(1e-8, 1e-8, 1e-8), maxit=400, method=method, loss=loss, f_scale=f_scale)
    ref = ofit.nonlinear_fit(x, y, ysd, fcn, prior_mean=pm, prior_err=psd, p0=p0, jac=jac, fitter='scipy_least_squares', bounds=b, **kw)
    fit = amd.nonlinear_fit(data=(x, y, ysd), model=amd.multiexp(K), prior=(pm, psd), p0=p0, fitter='mi355x_trf', bounds=b, **kw)
    compare(fit, ref, method)
    # the point of it: the outliers do not drag the fit -- closer to the truth than plain least squares, in units of the errors
    plain = amd.nonlinear_fit(data=(x, y, ysd), model=amd.multiexp(K), prior=(pm, psd), p0=p0, fitter='mi355x_trf', bounds=b,
                              tol=(1e-8, 1e-8, 1e-8), maxit=400, method=method)
    if loss != 'arctan':         # (arctan is bounded: a bad start can leave it at a far stationary point, in scipy as here)
        assert np.linalg.norm((fit.pmean - truth) / plain.psdev) < np.linalg.norm((plain.pmean - truth) / plain.psdev)
    # fit.residuals / fit.J are the TRUE whitened residuals and Jacobian (_scipy.py:160-161): chi2 = |f|^2, J^T J = what logGBF used
    assert float(fit.residuals @ fit.residuals) == pytest.approx(fit.chi2, rel=1e-9)
    sign, ld = np.linalg.slogdet(fit.J.T @ fit.J)
    assert ld == pytest.approx(ref.logdet_JtJ, rel=1e-6, abs=1e-6)


@pytest.mark.parametrize('loss', ['linear', 'huber'])
@pytest.mark.parametrize('method', ['trf', 'dogbox', 'lm'])
def test_array_x_scale(amd, method, loss):
    if method == 'lm' and loss != 'linear':
        with pytest.raises(ValueError, match="supports only 'linear' loss"):
            x, y, ysd, pm, psd, truth, fcn, jac = with_outliers(5, 2)
            amd.nonlinear_fit(data=(x, y, ysd), model=amd.multiexp(2), prior=(pm, psd), fitter='mi355x_trf', method='lm', loss=loss)
        return
    x, y, ysd, pm, psd, truth, fcn, jac = with_outliers(5, 2) if loss != 'linear' else multiexp_case(5, 2)
    p0 = pm * 1.15
    xs = np.array([0.5, 2.0, 1.0, 0.25])
    kw = dict(tol=(1e-8, 1e-8, 1e-8), maxit=400, method=method, x_scale=xs, loss=loss)
    ref = ofit.nonlinear_fit(x, y, ysd, fcn, prior_mean=pm, prior_err=psd, p0=p0, jac=jac, fitter='scipy_least_squares', **kw)
    fit = amd.nonlinear_fit(data=(x, y, ysd), model=amd.multiexp(2), prior=(pm, psd), p0=p0, fitter='mi355x_trf', **kw)
    compare(fit, ref, method)
    # a scalar x_scale other than 1 is scipy's broadcast
    kw['x_scale'] = 0.5
    ref = ofit.nonlinear_fit(x, y, ysd, fcn, prior_mean=pm, prior_err=psd, p0=p0, jac=jac, fitter='scipy_least_squares', **kw)
    fit = amd.nonlinear_fit(data=(x, y, ysd), model=amd.multiexp(2), prior=(pm, psd), p0=p0, fitter='mi355x_trf', **kw)
    compare(fit, ref, method)


@pytest.mark.parametrize('loss', ['soft_l1', 'cauchy'])
def test_robust_loss_with_correlated_data_and_a_correlated_prior(amd, loss):
    """Blocks are whitened in their eigen basis (rows = modes, as gvar.PDF does): a robust loss is not invariant under a rotation
    of a block's whitened rows, so the basis is part of the answer."""
    x, y, ysd, pm, psd, truth, fcn, jac = with_outliers(7, 2, N=96)
    N = x.size
    cov = np.diag(ysd ** 2)
    for r0 in (0, 40):
        idx = np.arange(r0, r0 + 24)
        cov[np.ix_(idx, idx)] = np.outer(ysd[idx], ysd[idx]) * 0.6 ** np.abs(np.subtract.outer(idx, idx))
    rng = np.random.default_rng(3)
    L = np.tril(0.05 * rng.standard_normal((4, 4)), -1) + np.diag(psd)
    pcov = L @ L.T
    kw = dict(tol=(1e-8, 1e-8, 1e-8), maxit=400, method='trf', loss=loss, f_scale=1.5)
    ref = ofit.nonlinear_fit(x, y, cov, fcn, prior_mean=pm, prior_err=pcov, p0=pm * 1.05, jac=jac, fitter='scipy_least_squares', **kw)
    fit = amd.nonlinear_fit(data=(x, y, cov), model=amd.multiexp(2), prior=(pm, pcov), p0=pm * 1.05, fitter='mi355x_trf', **kw)
    compare(fit, ref, 'trf')
    assert fit.dof == ref.dof and fit.svdn == ref.svdn


def test_tape_model_and_no_prior(amd):
    """A compiled formula (its fused routes must step aside for the row rescaling) and a fit without a prior."""
    rng = np.random.default_rng(11)
    N = 400
    x = np.sort(rng.uniform(0.0, 4.0, N))
    pt = np.array([1.2, 0.8, 0.3])
    sd = np.full(N, 0.02)
    y = pt[0] * np.exp(-pt[1] * x) + pt[2] + sd * rng.standard_normal(N)
    bad = rng.choice(N, 30, replace=False)
    y[bad] += rng.uniform(0.3, 1.0, 30)

    def fcn(xx, p):
        from oracle import dual
        return p[0] * dual.exp(-p[1] * xx) + p[2]
    kw = dict(tol=(1e-8, 1e-8, 1e-8), maxit=400, method='trf', loss='soft_l1', f_scale=1.0)
    ref = ofit.nonlinear_fit(x, y, sd, fcn, p0=pt * 1.2, fitter='scipy_least_squares', **kw)
    fit = amd.nonlinear_fit(data=(x, y, sd), model=amd.expr('a*exp(-b*x) + c', ['a', 'b', 'c']), p0=pt * 1.2, fitter='mi355x_trf', **kw)
    assert gu.relmax(fit.pmean, ref.pmean) < 1e-6 and abs(fit.chi2 / ref.chi2 - 1) < 1e-6 and gu.relmax(fit.cov, ref.cov) < 1e-6
    assert fit.stopping_criterion == ref.stopping_criterion and fit.logGBF is None
    assert np.all(np.abs(fit.pmean - pt) < 6 * fit.psdev)
    # the same handle again with the linear loss: nothing of the robust run lingers
    lin = amd.nonlinear_fit(data=(x, y, sd), model=amd.expr('a*exp(-b*x) + c', ['a', 'b', 'c']), p0=pt * 1.2, fitter='mi355x_trf',
                            tol=(1e-8, 1e-8, 1e-8))
    ref2 = ofit.nonlinear_fit(x, y, sd, fcn, p0=pt * 1.2, fitter='scipy_least_squares', tol=(1e-8, 1e-8, 1e-8))
    assert gu.relmax(lin.pmean, ref2.pmean) < 1e-6 and gu.relmax(lin.cov, ref2.cov) < 1e-6


def test_c_abi_refusals(amd):
    """The C ABI's own guards (a host in another language sees these, not the Python checks): a loss outside the five, a
    non-positive f_scale or x_scale; a robust loss with lsqamd_set_prior (the prior must travel as rows), with MINPACK's lm."""
    import ctypes as C
    from lsqfit_amd import _lib
    x, y, ysd, pm, psd, truth, fcn, jac = with_outliers(9, 1)
    pr = amd.DeviceProblem(amd.multiexp(1), x, amd.Whitening(y, ysd, pm, psd))           # prior through lsqamd_set_prior
    lib, h = pr.lib, pr.h
    assert lib.lsqamd_set_loss(h, 7, 1.0) == -1 and b'loss' in lib.lsqamd_last_error(h)
    assert lib.lsqamd_set_loss(h, 2, 0.0) == -1 and lib.lsqamd_set_loss(h, 2, float('nan')) == -1
    bad = np.array([1.0, -2.0])
    assert lib.lsqamd_set_x_scale(h, _lib.dptr(bad)) == -1 and b'x_scale' in lib.lsqamd_last_error(h)
    assert lib.lsqamd_set_loss(h, 2, 1.5) == 0
    s = _lib.Summary()
    p0 = np.ascontiguousarray(pm, np.float64)
    pr.set_options((1e-8, 1e-8, 1e-8), 100, 'levenberg', alg='trf')
    rc = lib.lsqamd_run(h, _lib.dptr(p0), C.byref(s))
    assert _lib.ERRORS.get(rc) == 'EUNSUPPORTED' and b'rows' in lib.lsqamd_last_error(h)
    pr.set_options((1e-8, 1e-8, 1e-8), 100, 'levenberg', alg='minpack_lm')
    rc = lib.lsqamd_run(h, _lib.dptr(p0), C.byref(s))
    assert rc == -1 and b"supports only 'linear' loss" in lib.lsqamd_last_error(h)
    # the plain lm method ignores a loss that was left set (it is an option of the scipy methods): same fit as without
    pr.set_options((1e-8, 1e-10, 1e-10), 100)
    assert lib.lsqamd_run(h, _lib.dptr(p0), C.byref(s)) == 0
    chi2_with = s.chi2
    assert lib.lsqamd_set_loss(h, 0, 1.0) == 0 and lib.lsqamd_run(h, _lib.dptr(p0), C.byref(s)) == 0
    assert s.chi2 == chi2_with
    pr.close()
