"""-m gpu: f2 of SURVEY.md 8 -- chi**2(p) / pdf(p) at many parameter points in one device pass
(``_fit_dchi2`` / ``_fit_pdf``, src/lsqfit/__init__.py:1648-1816; lbatch layout of
``vegas_fit._chiv``, src/lsqfit/_extras.py:2467-2486) through ``lsqamd_chi2_points``."""
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


def test_pdf_dchi2_on_device(amd):
    """tests/test_lsqfit.py:415-454."""
    ymean, ysd = np.array([1.5, 0.8, 12.0]), np.array([1.0, 0.5, 13.0])
    model = amd.expr('p + 0*x', ['p', 'q'])
    fit = amd.nonlinear_fit(data=(np.zeros(3), ymean, ysd), model=model, prior=([0.0, 0.0], [2.0, 5.0]))
    assert abs(fit.pdf(fit.pmean) - 1.0) < 1e-7
    assert abs(fit.dchi2(fit.pmean)) < 1e-7
    p = fit.pmean + fit.psdev
    assert abs(fit.dchi2(p) - 2) < 1e-7
    assert abs(fit.pdf(p) - np.exp(-fit.dchi2(p) / 2)) < 1e-12
    p[0] = fit.pmean[0]
    assert abs(fit.dchi2(p) - 1) < 1e-7
    model1 = amd.expr('p + 0*x', ['p'])
    fit = amd.nonlinear_fit(data=(np.zeros(3), ymean, ysd), model=model1, prior=([0.0], [2.0]))
    assert abs(fit.dchi2(fit.pmean + fit.psdev) - 1) < 1e-7
    pts = fit.pmean + np.linspace(-2, 2, 7)[:, None] * fit.psdev
    np.testing.assert_allclose(fit.dchi2(pts), np.linspace(-2, 2, 7) ** 2, atol=1e-7)
    ref = ofit.nonlinear_fit(False, ymean, ysd, lambda p: np.full(3, p[0]) if not hasattr(p, 'der') else None,
                             prior_mean=[0.0], prior_err=[2.0], jac=lambda p: np.ones((3, 1)))
    assert abs(fit.pdf_lognorm - ofit.pdf_lognorm(ref)) < 1e-9 * abs(fit.pdf_lognorm)


CASES = {
    'diag': dict(N=300, P=16, seed=51, block=0, prior_corr=False),
    'blocks': dict(N=512, P=32, seed=52, block=64, prior_corr=True),
    'ragged': dict(N=333, P=10, seed=53, block=100, prior_corr=True),
    'wide': dict(N=700, P=300, seed=54, block=0, prior_corr=True),
}


@pytest.mark.parametrize('case', sorted(CASES))
def test_chi2_points_match_oracle(amd, case):
    from lsqfit_amd import synth
    d = synth.make_cosmix(**CASES[case])
    N, P = d['ymean'].size, d['p0'].size
    wh = amd.Whitening(d['ymean'], d['yerr'], *d['prior'])
    pr = amd.DeviceProblem(d['model'], d['x'], wh)
    pdf = ofit.build_pdf(d['ymean'], gu.dense_cov(d['yerr'], N), d['prior'][0], d['prior'][1])
    rng = np.random.default_rng(8)
    pts = d['p_true'] * (1 + 1e-3 * rng.standard_normal((37, P)))
    want = np.array([np.sum(gu.pdf_residual(pdf, d['x'], q, gu.cosmix_fcn) ** 2) for q in pts])
    got = pr.chi2_points(pts)
    assert gu.relmax(got, want) < 1e-9
    # single-point entry point agrees, and so does a chunked pass (tiny scratch)
    assert abs(pr.chi2(pts[5]) / got[5] - 1) < 1e-12
    small = pr.chi2_points(pts, max_scratch_bytes=1)
    assert np.array_equal(small, got)
    pr.close()


def test_many_points_throughput(amd):
    """10^4 points (one vegas iteration, src/lsqfit/__init__.py:1706-1709 `neval = 10_000`) on a
    (4096, 16) correlated fit; linearity property: chi2 along a line through pmean is a parabola
    up to O(t^3) terms, and the minimum over the cloud is at pmean."""
    import time
    from lsqfit_amd import synth
    d = synth.make_cosmix(N=4096, P=16, seed=55, block=256, prior_corr=False)
    fit = amd.nonlinear_fit(data=(d['x'], d['ymean'], d['yerr']), model=d['model'], prior=d['prior'])
    rng = np.random.default_rng(9)
    M = 10000
    L = np.linalg.cholesky(fit.cov)
    pts = fit.pmean + rng.standard_normal((M, 16)) @ L.T
    t0 = time.perf_counter()
    dc = fit.dchi2(pts)
    dt = time.perf_counter() - t0
    print('chi2 at %d points of a (4096, 16) block-correlated fit: %.1f ms (%.2e point-rows/s)'
          % (M, dt * 1e3, M * 4096 / dt))
    assert dc.min() > -1e-6 * fit.chi2
    # Gaussian limit: dchi2 of draws from N(pmean, cov) is chi^2 with P dof
    assert abs(dc.mean() / 16 - 1) < 0.1
    k = int(np.argmax(dc))
    ref = gu.oracle_fit(d)
    assert abs(dc[k] - ofit.dchi2(ref, pts[k])) < 1e-6 * (abs(dc[k]) + fit.chi2)


def test_dchi2_with_data_prior_cross_correlations(amd):
    """examples/y-noerr.py's shape: prior entries whitened together with the data (parameter rows).  chi2 at many points
    in one pass -- parameter rows take the batch dimension too -- against one evaluation per point, and 0 at the minimum."""
    from tests.helpers import load, y_noerr_joint
    k = load('kat.json')['y_noerr']
    nexp = 2
    x, mean, cov = y_noerr_joint(k, nexp)
    P, n = 2 * nexp, len(k['x'])
    fit = amd.nonlinear_fit(data=(x, mean[:n], cov[:n, :n]), model=amd.multiexp(nexp), prior=(mean[n:], cov[n:, n:]),
                            cross=cov[:n, n:], tol=k['tol'], svdcut=k['svdcut'], solver='qr')
    rng = np.random.default_rng(4)
    z = rng.standard_normal((9, P))
    pts = fit.pmean + 0.1 * z @ np.linalg.cholesky(fit.cov).T          # 0.1-sigma steps with the fit's own correlations
    got = fit.dchi2(pts)
    one = np.array([fit.problem.chi2(p) for p in pts]) - fit.chi2
    assert np.allclose(got, one, rtol=1e-10, atol=1e-10)
    assert abs(fit.dchi2(fit.pmean)) < 1e-6 * max(1.0, fit.chi2)
    # (no quadratic-form check here: the data of this example carry no errors, so a tenth of a prior-dominated sigma
    # leaves the manifold the data pin down and chi2 is dominated by second-order terms)
    assert np.all(got > 0)
