"""-m gpu: edge cases of the path, as the reference tests them: maxit = 0
(tests/test_lsqfit.py:405-413), scalar data / prior and x-less two-point data (:455-470), one
parameter, one data row, more parameters than data, ragged block layouts with 1 x 1 rows between
blocks, P just off the tile sizes, zero-width errors rejected, block sizes that are not multiples
of anything."""
import numpy as np
import pytest

from oracle import fit as ofit
from oracle import gvar_lite
from tests import gpu_util as gu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def amd():
    import lsqfit_amd
    from lsqfit_amd import _lib
    _lib.load()
    return lsqfit_amd


def wavg(m, s):
    w = 1.0 / np.asarray(s, float) ** 2
    return float(np.sum(w * m) / np.sum(w)), float(np.sum(w) ** -0.5)


def test_maxit0_on_device(amd):
    model = amd.expr('p + 0*x', ['p'])
    fit = amd.nonlinear_fit(data=(np.zeros(2), [1.5, 0.8], [1.0, 0.5]), model=model, prior=([0.0], [2.0]), maxit=0)
    np.testing.assert_allclose(fit.pmean, [0.0]); np.testing.assert_allclose(fit.psdev, [2.0])
    assert fit.nit == 0 and fit.error is None and fit.stopping_criterion == 0
    assert abs(fit.chi2 - (1.5 ** 2 + (0.8 / 0.5) ** 2)) < 1e-12
    fit = amd.nonlinear_fit(data=(np.zeros(2), [1.5, 0.8], [1.0, 0.5]), model=model, p0=[0.0], maxit=0)
    np.testing.assert_allclose(fit.pmean, [0.0])
    assert np.all(np.isinf(fit.psdev)) and fit.logGBF is None


def test_unusual_cases_on_device(amd):
    model = amd.expr('p + 0*x', ['p'])
    fit = amd.nonlinear_fit(data=(np.zeros(1), [1.5], [0.1]), model=model, prior=([2.0], [0.5]))
    m, s = wavg([1.5, 2.0], [0.1, 0.5])
    assert gvar_lite.fmt(fit.pmean[0], fit.psdev[0]) == gvar_lite.fmt(m, s)
    assert abs(fit.pmean[0] - m) < 1e-9 and abs(fit.psdev[0] - s) < 1e-12      # LM stops on xtol = 1e-8
    fit = amd.nonlinear_fit(data=(np.zeros(2), [1.5, 1.7], [0.1, 0.2]), model=model, prior=([2.0], [0.5]), tol=1e-8)
    m, s = wavg([1.5, 1.7, 2.0], [0.1, 0.2, 0.5])
    assert abs(fit.pmean[0] - m) < 1e-9 and abs(fit.psdev[0] - s) < 1e-12
    assert fit.dof == 2


def test_more_parameters_than_data(amd):
    """P > N is legal with a prior (dof = N): 3 data rows, 10 parameters."""
    from lsqfit_amd import synth
    d = synth.make_cosmix(N=3, P=10, seed=71, block=0, prior_corr=True)
    fit = amd.nonlinear_fit(data=(d['x'], d['ymean'], d['yerr']), model=d['model'], prior=d['prior'])
    ref = gu.oracle_fit(d)
    assert fit.dof == 3 and fit.error is None
    assert gu.relmax(fit.pmean, ref.pmean) < 1e-6 and gu.relmax(fit.cov, ref.cov) < 1e-6
    assert abs(fit.chi2 - ref.chi2) < 1e-6 * max(1.0, ref.chi2)


@pytest.mark.parametrize('P', [2, 126, 130, 254, 258])
def test_parameter_counts_around_tile_edges(amd, P):
    from lsqfit_amd import synth
    d = synth.make_cosmix(N=400, P=P, seed=72 + P, block=0, prior_corr=(P < 200))
    fit = amd.nonlinear_fit(data=(d['x'], d['ymean'], d['yerr']), model=d['model'], prior=d['prior'], p0=d['p_true'])
    ref = gu.oracle_fit(d, p0=d['p_true'])
    assert gu.relmax(fit.pmean, ref.pmean) < 1e-6
    assert gu.relmax(fit.cov, ref.cov) < 1e-6
    assert abs(fit.chi2 / ref.chi2 - 1) < 1e-6


def test_ragged_blocks_with_gaps(amd):
    """Blocks of sizes 3, 17, 130 and 64 separated by 1 x 1 rows, the last one ending the data."""
    from lsqfit_amd import synth
    N, P = 300, 12
    d = synth.make_cosmix(N=N, P=P, seed=80, block=0, prior_corr=True)
    rng = np.random.default_rng(81)
    sd = np.asarray(d['yerr'])
    blocks = []
    for r0, B in ((5, 3), (20, 17), (60, 130), (236, 64)):
        U = rng.uniform(0.1, 0.9, (B, 2 * B))
        c = U @ U.T
        dd = 1.0 / np.sqrt(np.diag(c))
        c *= np.outer(dd, dd)
        blocks.append((r0, c * np.outer(sd[r0:r0 + B], sd[r0:r0 + B])))
    d['yerr'] = dict(sdev=sd, blocks=blocks)
    fit = amd.nonlinear_fit(data=(d['x'], d['ymean'], d['yerr']), model=d['model'], prior=d['prior'], p0=d['p_true'])
    ref = gu.oracle_fit(d, p0=d['p_true'])
    assert fit.nblocks == ref.nblocks
    assert gu.relmax(fit.pmean, ref.pmean) < 1e-6 and gu.relmax(fit.cov, ref.cov) < 1e-6
    assert abs(fit.chi2 / ref.chi2 - 1) < 1e-6
    D = fit.dp_dinputs()
    assert gu.relmax(D[:, :N], ofit.dp_dinputs(ref)[:, :N]) < 1e-6


def test_bad_inputs_are_rejected(amd):
    model = amd.expr('p + 0*x', ['p'])
    with pytest.raises(ValueError):
        amd.nonlinear_fit(data=(np.zeros(2), [1.0, 2.0], [1.0, 0.0]), model=model, prior=([0.0], [1.0]))
    with pytest.raises(ValueError):
        amd.nonlinear_fit(data=(np.zeros(2), [1.0, 2.0], [1.0, 1.0]), model=model, prior=([0.0], [0.0]))
    with pytest.raises(ValueError):
        amd.nonlinear_fit(data=(np.zeros(2), [1.0, 2.0], [1.0, 1.0]), model=model)        # neither p0 nor prior
    with pytest.raises(ValueError):
        amd.nonlinear_fit(data=(np.zeros(2), [1.0, 2.0], [1.0, 1.0]), model=model, prior=([0.0, 1.0], [1.0, 1.0]))
    # non-finite model values surface as an error, not as a crash
    bad = amd.expr('log(p)*x', ['p'])
    fit = None
    try:
        fit = amd.nonlinear_fit(data=(np.ones(3), [1.0, 2.0, 3.0], [1.0, 1.0, 1.0]), model=bad, p0=[-1.0])
    except RuntimeError as e:
        assert 'finite' in str(e)
    if fit is not None:
        assert fit.error is not None


def test_expression_model_with_many_parameters(amd):
    """Expression-tape models are differentiated 16 parameters per pass: a 20-term multi-exponential
    written out as an expression (P = 40, three passes) must reproduce the built-in multiexp model
    (same J^T J, J^T f to rounding) and the oracle fit."""
    K = 20
    rng = np.random.default_rng(90)
    x = np.linspace(0.1, 3.0, 150)
    a = rng.uniform(0.5, 1.5, K)
    e = 0.3 + 0.25 * np.arange(K)
    p_true = np.concatenate([a, e])
    y = gu.multiexp_fcn(x, p_true)
    ysd = 1e-3 * np.abs(y)
    ymean = y + ysd * rng.standard_normal(x.size)
    names = ['a%d' % k for k in range(K)] + ['e%d' % k for k in range(K)]
    text = ' + '.join('a%d*exp(-e%d*x)' % (k, k) for k in range(K))
    tape = amd.expr(text, names)
    pm, ps = p_true * (1 + 0.01 * rng.standard_normal(2 * K)), 0.05 * p_true
    wh = amd.Whitening(ymean, ysd, pm, ps)
    pa = amd.DeviceProblem(tape, x, wh)
    pb = amd.DeviceProblem(amd.multiexp(K), x, wh)
    ca, cb = pa.normal(pm), pb.normal(pm)
    assert abs(ca / cb - 1) < 1e-12
    assert gu.relmax(pa.get_jtj(), pb.get_jtj()) < 1e-12
    assert gu.relmax(pa.get_grad(), pb.get_grad()) < 1e-11
    fit = amd.nonlinear_fit(data=(x, ymean, ysd), model=tape, prior=(pm, ps), problem=pa)
    ref = ofit.nonlinear_fit(x, ymean, ysd, gu.multiexp_fcn, prior_mean=pm, prior_err=ps, jac=gu.multiexp_jac,
                             solver='cholesky')
    assert gu.relmax(fit.pmean, ref.pmean) < 1e-6
    assert gu.relmax(fit.cov, ref.cov) < 1e-6
    assert abs(fit.chi2 / ref.chi2 - 1) < 1e-6
    pa.close(); pb.close()


def test_noprior_zero_dof_on_device(amd):
    """tests/test_lsqfit.py:671-713 on the device: f_i = p_i**2 (row i selects its own parameter),
    8 correlated data, 8 parameters, no prior."""
    from tests.test_oracle_kat import _noprior_data
    ymean, ycov = _noprior_data()
    model = amd.expr(' + '.join('s%d*p%d**2' % (i, i) for i in range(8)), ['p%d' % i for i in range(8)],
                     xnames=tuple('s%d' % i for i in range(8)))
    fit = amd.nonlinear_fit(data=(np.eye(8), ymean, ycov), model=model, p0=np.full(8, 0.1), tol=1e-14)
    assert fit.logGBF is None and fit.dof == 0
    assert abs(fit.chi2) < 1e-4
    np.testing.assert_allclose(fit.pmean ** 2, ymean, rtol=1e-4)
    g = np.diag(2 * fit.pmean)
    np.testing.assert_allclose(g @ fit.cov @ g.T, ycov, rtol=1e-4, atol=1e-4 * np.abs(ycov).max())


def test_uncorrelated_data_flag_on_device(amd):
    """tests/test_lsqfit.py:1152-1167: udata=... drops the data correlations."""
    from tests.test_oracle_kat import _udata_inputs
    m, cov = _udata_inputs()
    model = amd.expr('p + 0*x', ['p'])
    for kw in (dict(prior=([1.0], [1.0])), dict(p0=[1.0])):
        f1 = amd.nonlinear_fit(udata=(np.zeros(4), m, cov), model=model, **kw)
        f2 = amd.nonlinear_fit(data=(np.zeros(4), m, cov), model=model, **kw)
        assert abs(f1.pmean[0] - f2.pmean[0]) < 5e-4
        assert abs(2 * f1.psdev[0] - f2.psdev[0]) < 5e-4


@pytest.mark.parametrize('svdcut', [1e-20, 1e-2])
def test_svd_cut_floor_on_device(amd, svdcut):
    """tests/test_lsqfit.py:773-826 on the device (eigen-mode whitening of the floored blocks)."""
    from tests.test_oracle_kat import _svd_case, check_svd_fit
    y, ycov, pm, pcov, sig1, sig2 = _svd_case()
    model = amd.expr('s0*p0**2 + s1*p1**2', ['p0', 'p1'], xnames=('s0', 's1'))
    fit = amd.nonlinear_fit(data=(np.eye(2), y, ycov), model=model, prior=(pm, pcov), svdcut=svdcut)
    check_svd_fit(fit, svdcut, sig1, sig2)
    ref = ofit.nonlinear_fit(False, y, ycov, lambda p: p * p, prior_mean=pm, prior_err=pcov, svdcut=svdcut)
    assert gu.relmax(fit.pmean, ref.pmean) < 1e-6 and gu.relmax(fit.cov, ref.cov) < 1e-6
    assert abs(fit.chi2 - ref.chi2) < 1e-6 * max(1.0, ref.chi2)
    assert abs(fit.logGBF - ref.logGBF) < 1e-6 * abs(ref.logGBF)


def test_query_devices_reports_the_mi355x():
    import ctypes
    from lsqfit_amd import _lib
    lib = _lib.load()
    n, mem = ctypes.c_int32(0), ctypes.c_int64(0)
    buf = ctypes.create_string_buffer(64)
    assert lib.lsqamd_query_devices(ctypes.byref(n), 0, buf, 64, ctypes.byref(mem)) == 0
    assert n.value >= 1 and buf.value.startswith(b'gfx950') and mem.value > 200 * (1 << 30)


@pytest.mark.parametrize('scale', [1.0, 1e14])
def test_cosine_model_range_flag_at_size(amd, scale):
    """From a few million cosines per evaluation on, the model kernels run as a pair -- one without far-range trig
    code, one with -- and a device flag computed from the frequencies leaves exactly one of them with work
    (api.hip residual_vector_launch, model.hip trig_range_kernel).  Both sides of the flag against numpy."""
    from lsqfit_amd import synth
    d = synth.make_cosmix(N=8192, P=1024, seed=12, block=256, prior_corr=False)
    wh = amd.Whitening(d['ymean'], d['yerr'], *d['prior'])
    pr = amd.DeviceProblem(d['model'], d['x'], wh)
    p = d['p_true'].copy()
    p[512:] *= scale
    chi2 = pr.normal(p)
    assert pr.lib.lsqamd_debug_flags(pr.h) & 2                      # the fused whitening kernel ran
    A = pr.get_jtj()
    Jr = gu.cosmix_jac(d['x'], p)
    r = gu.cosmix_fcn(d['x'], p) - d['ymean']
    Aref = np.diag(1.0 / np.asarray(d['prior'][1]) ** 2)
    c2 = float(np.sum(((p - d['prior'][0]) / np.asarray(d['prior'][1])) ** 2))
    for r0, c in d['yerr']['blocks']:
        B = c.shape[0]
        sol = np.linalg.solve(c, np.column_stack([Jr[r0:r0 + B], r[r0:r0 + B]]))
        Aref = Aref + Jr[r0:r0 + B].T @ sol[:, :-1]
        c2 += float(r[r0:r0 + B] @ sol[:, -1])
    assert np.abs(A - Aref).max() <= 1e-9 * np.abs(Aref).max()
    assert chi2 == pytest.approx(c2, rel=1e-9)
    pr.close()


@pytest.mark.parametrize('scale', [1.0, 1e4, 1e9, 1e14, 1e17, -1.0, -3e12])
def test_cosine_model_at_large_arguments(amd, scale):
    """cos / sin of w x in the model kernels: a two-term Cody-Waite reduction carried by FMAs up to |w x| = 1e13
    (1.1e-16 absolute; the quadrant read off the mantissa of w x 2/pi + 1.5 2^52, negative arguments included), the
    library routine beyond -- model values and Jacobian (analytic kernel, the kernel fused
    into the whitening product, and the residual kernel's cosine-only path) against numpy at every magnitude."""
    from lsqfit_amd import synth
    for block in (0, 128):
        d = synth.make_cosmix(N=512, P=128, seed=8, block=block, prior_corr=False)
        wh = amd.Whitening(d['ymean'], d['yerr'], *d['prior'])
        pr = amd.DeviceProblem(d['model'], d['x'], wh)
        p = d['p_true'].copy()
        p[64:] *= scale
        f = pr.fcn(p)
        ref = gu.cosmix_fcn(d['x'], p)
        assert np.abs(f - ref).max() < 1e-13 * 64
        pr.normal(p)
        if block == 0:
            J = pr.get_J_data()
            Jref = gu.cosmix_jac(d['x'], p) * wh.wdiag[:, None]
            assert np.abs(J - Jref).max() <= 1e-14 * np.abs(Jref).max()
        else:       # whitened rows differ from numpy's by the rotation inside a block: compare J^T J
            A = pr.get_jtj()
            Jr = gu.cosmix_jac(d['x'], p)
            Aref = np.diag(1.0 / np.asarray(d['prior'][1]) ** 2)
            for r0, c in d['yerr']['blocks']:
                B = c.shape[0]
                Aref = Aref + Jr[r0:r0 + B].T @ np.linalg.solve(c, Jr[r0:r0 + B])
            assert np.abs(A - Aref).max() <= 1e-9 * np.abs(Aref).max()
        pr.close()
