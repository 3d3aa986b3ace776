"""-m gpu: f1 of SURVEY.md 8 -- ``fit.p`` with its input correlations.

The device computes D = dp/d[y, prior] = cov [J_f ; I]^T inv(C_reg)
(``_getp``, src/lsqfit/__init__.py:897-911; ``chivw``, _utilities.pyx:96-139)
through the C ABI (``lsqamd_dpdy``); checked against the oracle's restatement,
against the reference's own error-budget table (examples/simple.out:24-32) and
test_partialerr1 (tests/test_lsqfit.py:1474-1510).  Tolerance 1e-6 relative."""
import numpy as np
import pytest

from oracle import fit as ofit
from tests import gpu_util as gu
from tests.helpers import load
from tests.test_oracle_kat import _simple_fit, parse_errorbudget

pytestmark = pytest.mark.gpu
KAT = load('kat.json')


@pytest.fixture(scope='module')
def amd():
    import lsqfit_amd
    from lsqfit_amd import _lib
    _lib.load()
    return lsqfit_amd


def test_simple_example_error_budget_on_device(amd):
    ymean = np.array([1.376, 2.010, 1.329, 1.582, 2.0])
    ycov = np.zeros((5, 5))
    ycov[:2, :2] = [[0.0047, 0.01], [0.01, 0.056]]
    ycov[2:4, 2:4] = [[0.0047, 0.0067], [0.0067, 0.0136]]
    ycov[4, 4] = 0.25
    # the dict-valued fit function of examples/simple.py, flattened: rows 0-3 exp(a + x b), row 4 b/a --
    # each range runs its own formula (lsqamd_set_tape_programs)
    x = np.array([0.1, 1.0, 0.1, 0.5, 0.0])
    model = amd.piecewise([(4, 'exp(a + x*b)'), (1, 'b/a')], ['a', 'b'])
    fit = amd.nonlinear_fit(data=(x, ymean, ycov), model=model, prior=([0.5, 0.5], [0.5, 0.5]))
    ref, cov_in = _simple_fit()
    assert gu.relmax(fit.pmean, ref.pmean) < 1e-6
    D = fit.dp_dinputs()
    assert D.shape == (2, 7)
    assert gu.relmax(D, ofit.dp_dinputs(ref)) < 1e-6
    assert gu.relmax(D @ cov_in @ D.T, fit.cov) < 1e-8
    a, b = fit.pmean
    grads = {'a': [1.0, 0.0], 'b/a': [-b / a ** 2, 1.0 / a], 'b': [0.0, 1.0]}
    vals = {'a': a, 'b/a': b / a, 'b': b}
    groups = {'y': [0, 1, 2, 3, 4], 'prior': [5, 6], 'total': list(range(7))}
    err = fit.partial_sdev(grads, groups, cov_in)
    want = parse_errorbudget(KAT['simple']['out'])
    for (g, name), pct in want.items():
        got = 100.0 * err[g, name] / abs(vals[g])
        assert '%.2f' % got == '%.2f' % pct, (g, name, got, pct)


def test_partialerr_weighted_average_on_device(amd):
    ny = 3
    model = amd.expr('py + 0*x', ['py', 'pn'])
    fit = amd.nonlinear_fit(data=(np.zeros(ny), np.full(ny, 2.0), np.full(ny, 0.125)), model=model,
                            prior=([0.1, 3.0], [1e4, 0.125]))
    D = fit.dp_dinputs()
    np.testing.assert_allclose(D[0, :ny], 1.0 / ny, rtol=1e-6)
    np.testing.assert_allclose(D[1, ny + 1], 1.0, rtol=1e-12)
    var = np.array([0.125] * ny + [1e4, 0.125]) ** 2
    err = fit.partial_sdev({'y': [1, 0], 'not y': [0, 1]},
                           {'y': [0, 1, 2], 'not y': [4], 'other prior': [3]}, var)
    assert abs(err['y', 'y'] - 0.125 / np.sqrt(ny)) < 1e-7
    assert err['y', 'not y'] == 0.0 and abs(err['y', 'other prior']) < 1e-5
    assert abs(err['not y', 'not y'] - 0.125) < 1e-12
    assert err['not y', 'y'] == 0.0 and err['not y', 'other prior'] == 0.0


CASES = {
    'diag': dict(N=300, P=16, seed=31, block=0, prior_corr=False),
    'blocks': dict(N=512, P=32, seed=32, block=64, prior_corr=True),
    'ragged': dict(N=333, P=10, seed=33, block=100, prior_corr=True),
    'tiles': dict(N=1024, P=128, seed=34, block=128, prior_corr=True),
    'wide': dict(N=700, P=300, seed=35, block=0, prior_corr=True),
}


@pytest.mark.parametrize('case', sorted(CASES))
def test_dp_dinputs_matches_oracle(amd, case):
    from lsqfit_amd import synth
    d = synth.make_cosmix(**CASES[case])
    fit = amd.nonlinear_fit(data=(d['x'], d['ymean'], d['yerr']), model=d['model'], prior=d['prior'])
    ref = gu.oracle_fit(d)
    N, P = d['ymean'].size, d['p0'].size
    D = fit.dp_dinputs()
    Dref = ofit.dp_dinputs(ref)
    assert D.shape == (P, N + P)
    # data and prior columns have very different scales: compare them separately
    assert gu.relmax(D[:, :N], Dref[:, :N]) < 1e-6
    assert gu.relmax(D[:, N:], Dref[:, N:]) < 1e-6
    # cov_p = D C D^T with C the (regulated) input covariance (doc/source/lsqfit.rst:112-113)
    C = np.zeros((N + P, N + P))
    C[:N, :N] = gu.dense_cov(d['yerr'], N) if not np.ndim(gu.dense_cov(d['yerr'], N)) == 1 else np.diag(np.asarray(d['yerr']) ** 2)
    perr = np.asarray(d['prior'][1])
    C[N:, N:] = perr if perr.ndim == 2 else np.diag(perr ** 2)
    assert gu.relmax(D @ C @ D.T, fit.cov) < 1e-6
    # directional variant: G @ D without forming D
    rng = np.random.default_rng(5)
    G = rng.standard_normal((3, P))
    assert gu.relmax(fit.dp_dinputs(G), G @ D) < 1e-9


def test_dp_dinputs_no_prior_and_svd_modes(amd):
    from lsqfit_amd import synth
    d = synth.make_cosmix(N=240, P=8, seed=36, block=80, prior_corr=False)
    # make one block nearly singular so that svdcut floors a mode (eigen whitening, modes kept)
    r0, cov = d['yerr']['blocks'][1]
    s = np.sqrt(np.diag(cov))
    corr = cov / np.outer(s, s)
    corr[1, :] = corr[0, :]
    corr[:, 1] = corr[:, 0]
    corr[1, 1] = 1.0
    corr[0, 1] = corr[1, 0] = 1.0 - 1e-13
    d['yerr']['blocks'][1] = (r0, corr * np.outer(s, s))
    fit = amd.nonlinear_fit(data=(d['x'], d['ymean'], d['yerr']), model=d['model'], p0=d['p_true'],
                            svdcut=1e-8)
    assert fit.svdn >= 1
    ref = ofit.nonlinear_fit(d['x'], d['ymean'], gu.dense_cov(d['yerr'], 240), gu.cosmix_fcn,
                             p0=d['p_true'], jac=gu.cosmix_jac, svdcut=1e-8, solver='cholesky')
    D = fit.dp_dinputs()
    assert D.shape == (8, 240)
    assert gu.relmax(D, ofit.dp_dinputs(ref)) < 1e-6


def test_dp_dinputs_c3_like(amd):
    """2048 x 256 with 256-row blocks and a dense prior: the interior (full-tile) GEMM path."""
    from lsqfit_amd import synth
    d = synth.make_cosmix(N=2048, P=256, seed=37, block=256, prior_corr=True)
    fit = amd.nonlinear_fit(data=(d['x'], d['ymean'], d['yerr']), model=d['model'], prior=d['prior'])
    ref = gu.oracle_fit(d)
    D = fit.dp_dinputs()
    Dref = ofit.dp_dinputs(ref)
    assert gu.relmax(D[:, :2048], Dref[:, :2048]) < 1e-6
    assert gu.relmax(D[:, 2048:], Dref[:, 2048:]) < 1e-6


def test_partialerr2_on_device(amd):
    """tests/test_lsqfit.py:1513-1549 (test_partialerr2): three measurements of p['y'] (sdev 0.125), priors
    y = 0.1(1e4) and 'not y' = 3.0(0.125).  d p_y / d y_0 = 1/3, d p_noty / d prior_noty = 1; the error
    budget puts all of p_y's error on the data (= the weighted average's) and all of p_noty's on its prior."""
    rng = np.random.default_rng(17)
    ny, sd = 3, 0.125
    y = 2.0 + sd * rng.standard_normal(ny)
    model = amd.expr('py + 0*pn + 0*x', ['py', 'pn'])
    fit = amd.nonlinear_fit(data=(np.zeros(ny), y, np.full(ny, sd)), model=model, prior=([0.1, 3.0], [1e4, sd]))
    assert abs(fit.pmean[0] / np.mean(y) - 1) < 1e-6 and abs(fit.pmean[1] / 3.0 - 1) < 1e-6
    D = fit.dp_dinputs()                                    # columns: y_0..y_2, prior_y, prior_noty
    assert abs(D[0, 0] - 1. / ny) < 5e-8 and abs(D[1, ny + 1] - 1.0) < 5e-8
    cov_in = np.array([sd ** 2] * ny + [1e8, sd ** 2])
    err = fit.partial_sdev({'y': [1.0, 0.0], 'not y': [0.0, 1.0]},
                           {'y': [0, 1, 2], 'not y': [ny + 1], 'other prior': [ny]}, cov_in)
    assert abs(err['y', 'y'] - sd / np.sqrt(ny)) < 5e-8
    assert abs(err['y', 'not y']) < 5e-8 and abs(err['y', 'other prior']) < 5e-6
    assert abs(err['not y', 'not y'] - sd) < 5e-8
    assert abs(err['not y', 'y']) < 5e-8 and abs(err['not y', 'other prior']) < 5e-8
