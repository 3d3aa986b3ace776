"""-m gpu: mid-size problems (P = 128 .. 768, tile-aligned and not) device vs oracle.  These shapes are where
the tile-aligned machinery switches on one piece at a time -- the chained back substitution (P >= 256, a
multiple of 128), fused trailing-update launches with the pivot-wave diagonal kernel (P >= 640), the fused
whitening kernel in its 32- and 64-term forms, captured LM steps (second fit on the same handle) -- and the
small fuzz cases never reach them."""
import numpy as np
import pytest

from tests import gpu_util as gu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def amd():
    import lsqfit_amd
    from lsqfit_amd import _lib
    _lib.load()
    return lsqfit_amd


CASES = [dict(N=1024, P=256, block=0, prior_corr=False),
         dict(N=1536, P=384, block=256, prior_corr=True),
         dict(N=768, P=640, block=128, prior_corr=True),
         dict(N=1500, P=300, block=100, prior_corr=False),       # nothing tile-aligned
         dict(N=1024, P=256, block=512, prior_corr=False)]       # two large blocks: paired tile rows, 32-term tiles


@pytest.mark.parametrize('shape', CASES, ids=lambda s: 'N%d_P%d_B%d' % (s['N'], s['P'], s['block']))
def test_midsize_fit_matches_oracle(amd, shape):
    from lsqfit_amd import synth
    d = synth.make_cosmix(seed=1000 + shape['P'], **shape)
    P = shape['P']
    p0 = d['p0'] * (1 + 3e-4 * np.random.default_rng(shape['N']).standard_normal(P))
    wh = amd.Whitening(d['ymean'], d['yerr'], *d['prior'])
    pr = amd.DeviceProblem(d['model'], d['x'], wh)
    kw = dict(data=(d['x'], d['ymean'], d['yerr']), model=d['model'], prior=d['prior'], p0=p0, problem=pr, tol=1e-10)
    fit = amd.nonlinear_fit(**kw)
    again = amd.nonlinear_fit(**kw)                        # same handle: the captured steps replay from the first iteration
    assert fit.error is None and fit.nit >= 3
    assert np.array_equal(fit.pmean, again.pmean) and np.array_equal(fit.cov, again.cov) and fit.nit == again.nit
    ref = gu.oracle_fit(d, solver='cholesky', tol=1e-10, p0=p0)
    assert fit.dof == ref.dof
    assert fit.chi2 == pytest.approx(ref.chi2, rel=1e-8)
    A, g = pr.get_jtj(), pr.get_grad()
    dd = np.sqrt(np.diag(A))
    # both covariances come from normal equations: they agree to cond(J^T J) eps, 1e-6 when that is smaller
    cond = np.linalg.cond(A / np.outer(dd, dd))
    assert gu.relmax(fit.pmean, ref.pmean) < 1e-6
    assert gu.relmax(fit.cov, ref.cov) < max(1e-6, 50 * cond * 2.2e-16), cond
    assert fit.logGBF == pytest.approx(ref.logGBF, rel=1e-8, abs=1e-6)
    # the solve by itself against LAPACK (factorisation + chained / grouped back substitution)
    v = pr.solve_damped(1e-3, dd)
    vref = np.linalg.solve(A + 1e-3 * np.diag(dd ** 2), g)
    assert np.max(np.abs(np.abs(v) - np.abs(vref))) < 1e-8 * np.max(np.abs(vref))
    pr.close()
