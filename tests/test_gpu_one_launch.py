"""-m gpu: a whole small fit in ONE launch (jit.hip lsqamd_jit_lm, api.hip run_one_launch) -- compiled formula, at most a
few dozen parameters, a few thousand uncorrelated rows (or a few hundred correlated ones), plain lm.  Every case is checked
(a) against the ORACLE (oracle/fit.py: the restated gsl_multifit driver, src/lsqfit/_gsl.pyx:676-706, dual-number Jacobians)
at the north-star tolerance of 1e-6 -- parameters, chi2, covariance, logGBF, iteration count and stopping criterion -- with
the route flag asserted, so that the kernel is known to be what produced the numbers; and (b) against the general path
(half a dozen launches per iteration): same solve, same decision, same stopping test, sums in a different order.  Anything
irregular must fall back to the general path.  The batched kernel (lsqamd_jit_lmb) likewise: copies vs oracle fits of the
copies' data.  The hand-off through polled pinned memory is audited (LSQAMD_VERIFY_HANDOFF) and run in its copy mode
(LSQAMD_ZERO_COPY=0)."""
import ctypes

import numpy as np
import pytest

from oracle import dual
from oracle import fit as ofit
from tests import gpu_util as gu
from tests.helpers import load, nist_problem

pytestmark = pytest.mark.gpu
NIST = load('nist.json')
ONE = 32     # lsqamd_debug_flags bit 5


@pytest.fixture(scope='module')
def amd():
    import lsqfit_amd
    from lsqfit_amd import _lib
    _lib.load()
    return lsqfit_amd


@pytest.fixture(autouse=True)
def audited_handoff(monkeypatch):
    """Every fit of this file runs with the hand-off audit on: after the host has acted on a polled block, the library waits for
    the stream and compares the block with the device's own copy, word for word; the count of differences must stay 0."""
    monkeypatch.setenv('LSQAMD_VERIFY_HANDOFF', '1')
    yield
    from lsqfit_amd import _lib
    st = (ctypes.c_int64 * 3)()
    assert _lib.load().lsqamd_handoff_stats(st) == 0
    assert st[2] == 0, 'the host acted on %d words that differed from the device copy' % st[2]


def flags(fit):
    pr = fit.problem
    return pr.lib.lsqamd_debug_flags(pr.h)


def expr_fcn(text, names):
    """The formula as a function of (x, p) the oracle can differentiate (dual numbers)."""
    code = compile(text, '<expr>', 'eval')

    def fcn(x, p):
        ns = dict(dual.NAMESPACE)
        ns['x'] = x
        for k, n in enumerate(names):
            ns[n] = p[k]
        return eval(code, {'__builtins__': {}}, ns) + 0.0 * x        # (a constant formula still has one value per row)
    return fcn


def oracle_of(text, names, data, prior=None, p0=None, **kw):
    x, y, yerr = data
    N = np.asarray(y).size
    pm, pe = (None, None) if prior is None else prior
    kw.pop('model', None)
    kw.setdefault('solver', 'cholesky')            # (lsqfit_amd's default route; the reference's 'qr' is asked for by name)
    return ofit.nonlinear_fit(np.asarray(x, float), np.asarray(y, float), gu.dense_cov(yerr, N), expr_fcn(text, names),
                              prior_mean=pm, prior_err=pe, p0=p0, **kw)


def vs_oracle(one, ref, nit_slack=None, psig=1e-5):
    """North-star parity of a one-launch fit with the oracle: 1e-6 relative on p (+ psig sigma for parameters near zero and
    end points that xtol leaves a hair apart), chi2, cov, logGBF; the same stopping criterion; the iteration count to within one (the
    last, xtol-sized step may fall either side of the test)."""
    if nit_slack is None:          # (32 parameters, flat directions: the xtol test passes an iteration or two apart)
        nit_slack = max(2, ref.nit // 8) if one.pmean.size > 12 else 1
    assert abs(one.nit - ref.nit) <= nit_slack, (one.nit, ref.nit)
    assert one.stopping_criterion == ref.stopping_criterion
    assert one.dof == ref.dof
    assert np.all(np.abs(one.pmean - ref.pmean) <= 1e-6 * np.abs(ref.pmean) + psig * ref.psdev), (one.pmean, ref.pmean)
    assert one.chi2 == pytest.approx(ref.chi2, rel=1e-6, abs=1e-12)
    assert gu.relmax(one.cov, ref.cov) < 1e-6
    if ref.logGBF is not None:
        assert one.logGBF == pytest.approx(ref.logGBF, rel=1e-6, abs=1e-6)


def both(amd, monkeypatch, **kw):
    monkeypatch.setenv('LSQAMD_ONE_LAUNCH_FIT', '1')
    one = amd.nonlinear_fit(**kw)
    f1 = flags(one)
    monkeypatch.setenv('LSQAMD_ONE_LAUNCH_FIT', '0')
    gen = amd.nonlinear_fit(**kw)
    assert not flags(gen) & ONE
    return one, f1, gen


def agree(one, gen, rtol=1e-8):
    # (the two routes add their sums in different orders: at a stopping tolerance near the rounding floor -- the NIST
    #  problems run at xtol = 1e-10 -- the last, rounding-sized step may be accepted by one and rejected sixteen times by
    #  the other; both stop at the same point on the same criterion)
    assert abs(one.nit - gen.nit) <= max(2, gen.nit // 8)
    s1, s0 = one.fitter_results.summary, gen.fitter_results.summary
    assert s1.stopping_criterion == s0.stopping_criterion
    assert s1.nit == one.nit and s1.njev in (s1.nit, s1.nit + 1) and s1.nfev > s1.nit
    assert np.all(np.abs(one.pmean - gen.pmean) <= rtol * np.abs(gen.pmean) + 5e-6 * gen.psdev)
    assert abs(one.chi2 - gen.chi2) <= 1e-8 * max(gen.chi2, 1e-12) + 1e-20
    assert np.allclose(one.cov, gen.cov, rtol=1e-5, atol=1e-9 * np.max(np.abs(gen.cov)))
    assert one.logGBF == pytest.approx(gen.logGBF, rel=1e-8, abs=1e-8)


@pytest.mark.parametrize('name', ['misra1a', 'chwirut2', 'thurber', 'gauss1', 'mgh09', 'eckerle4', 'hahn1', 'lanczos3', 'boxbod'])
def test_nist_fits_agree_with_the_general_path(amd, name, monkeypatch):
    pr = nist_problem(name, NIST)
    model = amd.expr(pr['expr'], ['b%d' % (i + 1) for i in range(pr['P'])], pr['columns'][1:])
    x = np.stack([pr['x'][c] for c in pr['columns'][1:]], axis=1)
    one, f1, gen = both(amd, monkeypatch, data=(x, pr['y'], pr['ysd']), model=model, prior=(pr['prior_mean'], pr['prior_sd']),
                        p0=pr['p0'], tol=pr['tol'])
    assert f1 & ONE, 'the fit did not take the one-launch route'
    agree(one, gen)
    # ... and the oracle (the examples/nist.py harness: priors 0 +- 200|b|, start 2, tol 1e-10) and the certified values
    ref = ofit.nonlinear_fit(pr['x'], pr['y'], pr['ysd'], pr['fcn'], prior_mean=pr['prior_mean'], prior_err=pr['prior_sd'],
                             p0=pr['p0'], tol=pr['tol'], solver='cholesky')
    vs_oracle(one, ref, nit_slack=max(2, ref.nit // 8), psig=1e-4)
    assert np.all(np.abs(one.pmean - pr['certified']) <= 1e-2 * pr['certified_sd'] + 1e-9 * np.abs(pr['certified']))
    np.testing.assert_allclose(one.psdev, pr['certified_sd'], rtol=2e-3)
    # what reads the Jacobian afterwards finds it (never written by the kernel)
    assert np.allclose(one.J, gen.J, rtol=1e-5, atol=1e-7 * np.max(np.abs(gen.J)))      # (at end points up to 2e-6 sigma apart)
    assert np.allclose(one.residuals, gen.residuals, rtol=1e-5, atol=1e-7 * (1 + np.max(np.abs(gen.residuals))))


def curve(N=600, seed=3):
    rng = np.random.default_rng(seed)
    x = np.sort(rng.uniform(0.0, 5.0, N))
    pt = np.array([1.5, 0.7, 0.4, 2.0])
    f = pt[0] * np.exp(-pt[1] * x) + pt[2] * np.cos(pt[3] * x)
    sd = 0.02 + 0.01 * rng.random(N)
    return x, f + sd * rng.standard_normal(N), sd, pt


@pytest.mark.parametrize('prior', ['diag', 'dense', 'none'])
@pytest.mark.parametrize('scaler', ['more', 'levenberg', 'marquardt'])
def test_priors_and_scalers(amd, prior, scaler, monkeypatch):
    x, y, sd, pt = curve()
    model = amd.expr('a*exp(-b*x) + c*cos(d*x)', ['a', 'b', 'c', 'd'])
    kw = dict(data=(x, y, sd), model=model, p0=pt * 1.2, scaler=scaler)
    if prior == 'diag':
        kw['prior'] = (pt * 1.1, np.full(4, 0.5))
    elif prior == 'dense':
        L = np.tril(0.2 * np.random.default_rng(1).standard_normal((4, 4))) + 0.6 * np.eye(4)
        kw['prior'] = (pt * 1.1, L @ L.T)
    one, f1, gen = both(amd, monkeypatch, **kw)
    assert f1 & ONE
    agree(one, gen)
    vs_oracle(one, oracle_of('a*exp(-b*x) + c*cos(d*x)', ['a', 'b', 'c', 'd'], **kw))


def test_iteration_limit_and_resident_problem(amd, monkeypatch):
    x, y, sd, pt = curve(N=3000, seed=8)
    model = amd.expr('a*exp(-b*x) + c*cos(d*x)', ['a', 'b', 'c', 'd'])
    kw = dict(data=(x, y, sd), model=model, prior=(pt, np.full(4, 1.0)), p0=pt * 1.4, maxit=3)
    one, f1, gen = both(amd, monkeypatch, **kw)
    assert f1 & ONE and one.nit == 3
    agree(one, gen)
    assert one.stopping_criterion == gen.stopping_criterion == 0 and one.error == gen.error
    # three iterations of the oracle's driver from the same start end at the same point
    ref3 = oracle_of('a*exp(-b*x) + c*cos(d*x)', ['a', 'b', 'c', 'd'], **kw)
    assert ref3.nit == 3 and ref3.stopping_criterion == 0
    assert np.allclose(one.pmean, ref3.pmean, rtol=1e-9, atol=1e-9 * np.max(ref3.psdev)) and one.chi2 == pytest.approx(ref3.chi2, rel=1e-9)
    # the same handle again, to convergence, twice: nothing of the first run lingers
    monkeypatch.setenv('LSQAMD_ONE_LAUNCH_FIT', '1')
    kw['maxit'] = 200
    a = amd.nonlinear_fit(problem=one.problem, **kw)
    b = amd.nonlinear_fit(problem=one.problem, **kw)
    assert flags(a) & ONE and a.nit == b.nit and np.array_equal(a.pmean, b.pmean) and np.array_equal(a.cov, b.cov)
    vs_oracle(a, oracle_of('a*exp(-b*x) + c*cos(d*x)', ['a', 'b', 'c', 'd'], **kw))
    monkeypatch.setenv('LSQAMD_ONE_LAUNCH_FIT', '0')
    agree(a, amd.nonlinear_fit(**kw))


def test_irregular_fits_fall_back(amd, monkeypatch):
    """No prior and two parameters the data cannot tell apart: the damped matrix loses its pivot as mu shrinks, or the fit
    stalls -- whatever the general path does with it, the one-launch route must hand the fit over, not improvise."""
    rng = np.random.default_rng(5)
    x = np.linspace(0.0, 1.0, 200)
    y = 2.0 * x + 0.01 * rng.standard_normal(200)
    model = amd.expr('(a + b)*x + 0*c', ['a', 'b', 'c'])
    kw = dict(data=(x, y, np.full(200, 0.01)), model=model, p0=[0.5, 0.5, 0.1])
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        monkeypatch.setenv('LSQAMD_ONE_LAUNCH_FIT', '1')
        one = amd.nonlinear_fit(**kw)
        monkeypatch.setenv('LSQAMD_ONE_LAUNCH_FIT', '0')
        gen = amd.nonlinear_fit(**kw)
    assert abs(one.nit - gen.nit) <= 1 and np.allclose(one.pmean, gen.pmean, rtol=1e-7, atol=1e-10)
    assert (one.error is None) == (gen.error is None)
    # too many rows for one workgroup: the general path, silently
    xx = np.linspace(0.0, 5.0, 9000)
    big = amd.nonlinear_fit(data=(xx, np.exp(-0.5 * xx) + 0.01 * rng.standard_normal(9000), np.full(9000, 0.01)),
                            model=amd.expr('a*exp(-b*x)', ['a', 'b']), p0=[1.0, 1.0])
    assert not flags(big) & ONE and big.error is None


def correlated(N, seed, nblocks, rho=0.6):
    """A decaying curve with correlated errors: `nblocks` dense covariance blocks over the first rows, the rest uncorrelated."""
    rng = np.random.default_rng(seed)
    x = np.linspace(0.1, 4.0, N)
    pt = np.array([2.0, 0.9, 0.5, 0.25])
    f = pt[0] * np.exp(-pt[1] * x) + pt[2] * np.exp(-pt[3] * x)
    sd = 0.01 * (1.0 + x)
    bs = (N * 3 // 4) // nblocks
    blocks, cov = [], np.diag(sd ** 2)
    for b in range(nblocks):
        r0 = b * bs
        idx = np.arange(r0, r0 + bs)
        c = np.outer(sd[idx], sd[idx]) * rho ** np.abs(np.subtract.outer(idx, idx))
        blocks.append((r0, c))
        cov[np.ix_(idx, idx)] = c
    y = f + np.linalg.cholesky(cov) @ rng.standard_normal(N)
    return x, y, dict(sdev=sd, blocks=blocks), pt


@pytest.mark.parametrize('N,nblocks', [(40, 1), (96, 3), (256, 2), (17, 1)])
@pytest.mark.parametrize('solver', ['cholesky', 'qr'])
def test_correlated_data_is_whitened_inside_the_kernel(amd, N, nblocks, solver, monkeypatch):
    """The everyday lsqfit shape: a few dozen correlated points, a handful of parameters.  The workgroup files the raw rows
    in LDS and every thread forms its whitened row from them (W^T from the handle's whitening, as the general path's
    whitening product does)."""
    x, y, yerr, pt = correlated(N, seed=N + nblocks, nblocks=nblocks)
    model = amd.expr('a*exp(-b*x) + c*exp(-d*x)', ['a', 'b', 'c', 'd'])
    kw = dict(data=(x, y, yerr), model=model, prior=(pt, np.array([1.0, 0.5, 0.5, 0.2])), p0=pt * 1.1, solver=solver)
    one, f1, gen = both(amd, monkeypatch, **kw)
    assert f1 & ONE, 'the fit did not take the one-launch route'
    agree(one, gen)
    vs_oracle(one, oracle_of('a*exp(-b*x) + c*exp(-d*x)', ['a', 'b', 'c', 'd'], **kw))
    assert one.dof == gen.dof and one.Q == pytest.approx(gen.Q, rel=1e-7, abs=1e-12)
    assert np.allclose(one.J, gen.J, rtol=1e-6, atol=1e-8 * np.max(np.abs(gen.J)))          # (ensure_J: whitened rows)
    assert np.allclose(one.residuals, gen.residuals, rtol=1e-6, atol=1e-8)


def test_correlated_data_with_a_binding_svdcut(amd, monkeypatch):
    """A nearly singular block: the svdcut floor binds, the block is whitened through its eigen-modes (W^T with fewer
    modes than rows) -- the kernel reads the same factor."""
    rng = np.random.default_rng(11)
    N = 24
    x = np.linspace(0.2, 3.0, N)
    pt = np.array([1.5, 0.8])
    sd = np.full(N, 0.02)
    u = rng.standard_normal((N, 3))
    cov = 1e-4 * (u @ u.T) + np.diag(sd ** 2) * 1e-9            # rank 3 + a sliver
    y = pt[0] * np.exp(-pt[1] * x) + np.linalg.cholesky(cov + 1e-12 * np.eye(N)) @ rng.standard_normal(N)
    model = amd.expr('a*exp(-b*x)', ['a', 'b'])
    kw = dict(data=(x, y, dict(sdev=np.sqrt(np.diag(cov)), blocks=[(0, cov)])), model=model, prior=(pt, np.array([1.0, 1.0])),
              svdcut=1e-3)
    one, f1, gen = both(amd, monkeypatch, **kw)
    assert f1 & ONE and one.svdn == gen.svdn and one.svdn > 0
    agree(one, gen)
    assert one.dof == gen.dof
    ref = oracle_of('a*exp(-b*x)', ['a', 'b'], **kw)
    assert ref.svdn == one.svdn
    vs_oracle(one, ref)


def test_bootstrap_copies_are_one_launch_too(amd, monkeypatch):
    """Simulated / bootstrap copies of a small fit: one workgroup per copy in ONE launch (lsqamd_jit_lmb) instead of the
    lockstep engine's rounds; both engines must agree copy by copy, and a copy equals the single fit of its data."""
    pr = nist_problem('gauss1', NIST)
    model = amd.expr(pr['expr'], ['b%d' % (i + 1) for i in range(pr['P'])], pr['columns'][1:])
    x = np.stack([pr['x'][c] for c in pr['columns'][1:]], axis=1)
    kw = dict(model=model, prior=(pr['prior_mean'], pr['prior_sd']), p0=pr['p0'], tol=1e-8)
    monkeypatch.setenv('LSQAMD_ONE_LAUNCH_FIT', '1')
    fit = amd.nonlinear_fit(data=(x, pr['y'], pr['ysd']), **kw)
    one = fit.bootstrapped_fits(64, seed=3)
    assert one['rounds'] == 1
    monkeypatch.setenv('LSQAMD_ONE_LAUNCH_FIT', '0')
    lock = fit.bootstrapped_fits(64, seed=3)
    assert lock['rounds'] > 1
    assert np.array_equal(one['ymeans'], lock['ymeans'])
    assert np.max(np.abs(one['pmean'] - lock['pmean']) / lock['psdev']) < 1e-5
    assert np.allclose(one['chi2'], lock['chi2'], rtol=1e-8) and np.allclose(one['psdev'], lock['psdev'], rtol=1e-5)
    assert np.allclose(one['logGBF'], lock['logGBF'], rtol=1e-8, atol=1e-7)
    assert np.all(np.abs(one['nit'] - lock['nit']) <= 1) and np.array_equal(one['stopping_criterion'], lock['stopping_criterion'])
    monkeypatch.setenv('LSQAMD_ONE_LAUNCH_FIT', '1')
    for k in (0, 17, 63):
        single = amd.nonlinear_fit(data=(x, one['ymeans'][k], pr['ysd']), model=model, prior=(one['prior_means'][k], pr['prior_sd']),
                                   p0=fit.pmean, tol=1e-8)
        assert single.nit == one['nit'][k] and np.array_equal(single.pmean, one['pmean'][k])      # (the same kernel body)
        assert np.array_equal(single.psdev, one['psdev'][k]) and single.chi2 == one['chi2'][k]
        # ... and the oracle's fit of that copy's data and prior
        ref = ofit.nonlinear_fit(pr['x'], one['ymeans'][k], pr['ysd'], pr['fcn'], prior_mean=one['prior_means'][k],
                                 prior_err=pr['prior_sd'], p0=fit.pmean, tol=1e-8, solver='cholesky')
        assert abs(int(one['nit'][k]) - ref.nit) <= 1 and int(one['stopping_criterion'][k]) == ref.stopping_criterion
        assert np.all(np.abs(one['pmean'][k] - ref.pmean) <= 1e-6 * np.abs(ref.pmean) + 1e-5 * ref.psdev)
        assert one['chi2'][k] == pytest.approx(ref.chi2, rel=1e-6) and np.allclose(one['psdev'][k], ref.psdev, rtol=1e-6)
        assert one['logGBF'][k] == pytest.approx(ref.logGBF, rel=1e-6, abs=1e-6)


def test_batch_with_an_irregular_copy_takes_the_lockstep_engine(amd, monkeypatch):
    """One copy whose start makes the damped matrix lose a pivot sends the whole batch through the lockstep engine."""
    from lsqfit_amd import BatchedFits
    x, y, sd, pt = curve(N=300, seed=21)
    model = amd.expr('a*exp(-b*x) + c*cos(d*x) + 0*e', ['a', 'b', 'c', 'd', 'e'])
    B = 8
    pm = np.tile(np.append(pt, 0.0), (B, 1))
    ps = np.tile(np.array([1.0, 1.0, 1.0, 1.0, 1.0]), (B, 1))
    monkeypatch.setenv('LSQAMD_ONE_LAUNCH_FIT', '1')
    eng = BatchedFits(model, x, y, sd, pm, ps)
    good = eng.run(p0=pm * 1.05)
    assert good['rounds'] == 1 and np.all(good['stopping_criterion'] > 0)
    p0 = pm * 1.05
    p0[3] = np.nan                                  # a start that is not a number: chi2 is not finite at the first evaluation
    bad = eng.run(p0=p0)
    assert bad['rounds'] != 1 or not np.all(np.isfinite(bad['chi2']))
    ok = [b for b in range(B) if b != 3]
    assert np.allclose(bad['pmean'][ok], good['pmean'][ok], rtol=1e-6, atol=1e-9)
    eng.close()


def bumps(K, N, seed, correlated, background=False):
    """sum_k a_k exp(-b_k (x - c_k)^2) with fixed centres (+ a linear background): 2 K (+ 2) parameters, well conditioned."""
    rng = np.random.default_rng(seed)
    x = np.linspace(0.0, 10.0, N)
    cs = np.linspace(0.7, 9.3, K)
    a = 1.0 + 0.5 * rng.random(K)
    b = 2.0 + rng.random(K)
    f = sum(a[k] * np.exp(-b[k] * (x - cs[k]) ** 2) for k in range(K))
    sd = np.full(N, 0.02)
    if correlated:
        cov = np.outer(sd, sd) * 0.5 ** np.abs(np.subtract.outer(np.arange(N), np.arange(N)))
        y = f + np.linalg.cholesky(cov) @ rng.standard_normal(N)
        yerr = dict(sdev=sd, blocks=[(0, cov)])
    else:
        y, yerr = f + sd * rng.standard_normal(N), sd
    names = ['a%d' % k for k in range(K)] + ['b%d' % k for k in range(K)]
    text = ' + '.join('a%d*exp(-b%d*(x - %r)**2)' % (k, k, float(cs[k])) for k in range(K))
    pt = np.concatenate([a, b])
    if background:
        y = y + 0.3 + 0.05 * x
        names, text, pt = names + ['c0', 'c1'], text + ' + c0 + c1*x', np.concatenate([pt, [0.3, 0.05]])
    return x, y, yerr, text, names, pt


@pytest.mark.parametrize('K,N,correlated,bg', [(7, 100, False, False), (7, 64, True, False), (15, 128, False, False), (15, 120, True, True),
                                               (15, 90, False, True)])
def test_up_to_32_parameters_with_the_rows_in_lds(amd, K, N, correlated, bg, monkeypatch):
    """13 .. 32 parameters: the sums of J^T J no longer fit a thread's registers -- the (whitened) rows go to LDS, at most
    128 of them, and thread q adds up product q; the solve is the same one-wave elimination with up to 32 steps."""
    x, y, yerr, text, names, pt = bumps(K, N, seed=K + N, correlated=correlated, background=bg)
    model = amd.expr(text, names)
    kw = dict(data=(x, y, yerr), model=model, prior=(pt, np.full(pt.size, 0.5)), p0=pt * 1.05)
    one, f1, gen = both(amd, monkeypatch, **kw)
    assert f1 & ONE, 'the fit did not take the one-launch route'
    agree(one, gen)
    vs_oracle(one, oracle_of(text, names, **kw))
    assert np.allclose(one.J, gen.J, rtol=1e-6, atol=1e-8 * np.max(np.abs(gen.J)))
    # more CORRELATED rows than one chunk of the kernel's LDS rows: the general path, silently
    x2, y2, yerr2, _, _, _ = bumps(K, 300, seed=1, correlated=True, background=bg)
    big = amd.nonlinear_fit(data=(x2, y2, yerr2), model=model, prior=(pt, np.full(pt.size, 0.5)), p0=pt * 1.05)
    assert not flags(big) & ONE and big.error is None


@pytest.mark.parametrize('K,N', [(7, 1000), (7, 2600), (10, 700), (15, 400), (15, 700), (6, 2000)])
def test_wide_fits_with_many_uncorrelated_rows(amd, K, N, monkeypatch):
    """13 .. 32 parameters on up to a few thousand uncorrelated points (a spectrum with a dozen peaks): the rows pass through LDS a
    chunk at a time, thread q keeps the running sum of product q."""
    x, y, yerr, text, names, pt = bumps(K, N, seed=K + N, correlated=False, background=True)
    model = amd.expr(text, names)
    kw = dict(data=(x, y, yerr), model=model, prior=(pt, np.full(pt.size, 0.5)), p0=pt * 1.05)
    one, f1, gen = both(amd, monkeypatch, **kw)
    assert f1 & ONE, 'the fit did not take the one-launch route'
    agree(one, gen)
    vs_oracle(one, oracle_of(text, names, **kw))
    print('P = %d, N = %d: device run %.3f ms (general path %.3f ms), %d iterations' % (
        pt.size, N, one.fitter_results.summary.t_run_ms, gen.fitter_results.summary.t_run_ms, one.nit))
    # beyond ~400 000 row products per evaluation one workgroup loses to the general path's many: not taken
    x2, y2, yerr2, _, _, _ = bumps(K, 4000, seed=2, correlated=False, background=True)
    big = amd.nonlinear_fit(data=(x2, y2, yerr2), model=model, prior=(pt, np.full(pt.size, 0.5)), p0=pt * 1.05)
    assert not flags(big) & ONE and big.error is None


@pytest.mark.parametrize('N,P,maxit', [(1, 1, 5), (3, 2, 1), (5, 1, 50), (8192, 3, 30)])
def test_smallest_and_largest_shapes(amd, N, P, maxit, monkeypatch):
    """One data point, one parameter, one iteration; and the largest row count the kernel takes."""
    rng = np.random.default_rng(N + P)
    x = np.linspace(0.5, 3.0, N)
    names = ['a', 'b', 'c'][:P]
    text = {1: 'a*x', 2: 'a*exp(-b*x)', 3: 'a*exp(-b*x) + c'}[P]
    pt = np.array([1.3, 0.6, 0.2])[:P]
    f = {1: pt[0] * x, 2: pt[0] * np.exp(-pt[min(1, P - 1)] * x), 3: pt[0] * np.exp(-pt[min(1, P - 1)] * x) + pt[P - 1]}[P]
    sd = np.full(N, 0.05)
    y = f + sd * rng.standard_normal(N)
    kw = dict(data=(x, y, sd), model=amd.expr(text, names), prior=(pt, np.full(P, 0.5)), p0=pt * 1.2, maxit=maxit)
    one, f1, gen = both(amd, monkeypatch, **kw)
    assert f1 & ONE
    agree(one, gen)
    assert one.stopping_criterion == gen.stopping_criterion and (one.error is None) == (gen.error is None)
    vs_oracle(one, oracle_of(text, names, **kw), nit_slack=1 if maxit >= 30 else 0)


@pytest.mark.parametrize('wide', [False, True])
def test_bootstrap_copies_of_a_correlated_fit(amd, wide, monkeypatch):
    """Copies of a fit with covariance blocks (and, `wide`, with more than a dozen parameters): the batched kernel whitens
    each copy's rows itself; against the lockstep engine and against single fits of the copies' data."""
    if wide:
        x, y, yerr, text, names, pt = bumps(7, 96, seed=5, correlated=True)
        psd = np.full(pt.size, 0.5)
    else:
        x, y, yerr, pt = correlated(96, seed=4, nblocks=3)
        text, names, psd = 'a*exp(-b*x) + c*exp(-d*x)', ['a', 'b', 'c', 'd'], np.array([1.0, 0.5, 0.5, 0.2])
    model = amd.expr(text, names)
    monkeypatch.setenv('LSQAMD_ONE_LAUNCH_FIT', '1')
    fit = amd.nonlinear_fit(data=(x, y, yerr), model=model, prior=(pt, psd), p0=pt * 1.05)
    one = fit.bootstrapped_fits(24, seed=9)
    assert one['rounds'] == 1
    monkeypatch.setenv('LSQAMD_ONE_LAUNCH_FIT', '0')
    lock = fit.bootstrapped_fits(24, seed=9)
    assert lock['rounds'] > 1 and np.array_equal(one['ymeans'], lock['ymeans'])
    assert np.max(np.abs(one['pmean'] - lock['pmean']) / lock['psdev']) < 1e-5
    assert np.allclose(one['chi2'], lock['chi2'], rtol=1e-8) and np.allclose(one['psdev'], lock['psdev'], rtol=1e-5)
    assert np.allclose(one['logGBF'], lock['logGBF'], rtol=1e-8, atol=1e-7)
    monkeypatch.setenv('LSQAMD_ONE_LAUNCH_FIT', '1')
    for k in (0, 23):
        single = amd.nonlinear_fit(data=(x, one['ymeans'][k], yerr), model=model, prior=(one['prior_means'][k], psd), p0=fit.pmean)
        assert single.nit == one['nit'][k] and np.array_equal(single.pmean, one['pmean'][k])
        ref = oracle_of(text, names, data=(x, one['ymeans'][k], yerr), prior=(one['prior_means'][k], psd), p0=fit.pmean)
        assert abs(int(one['nit'][k]) - ref.nit) <= 1
        assert np.all(np.abs(one['pmean'][k] - ref.pmean) <= 1e-6 * np.abs(ref.pmean) + 1e-5 * ref.psdev)
        assert one['chi2'][k] == pytest.approx(ref.chi2, rel=1e-6) and np.allclose(one['psdev'][k], ref.psdev, rtol=1e-6)
        assert one['logGBF'][k] == pytest.approx(ref.logGBF, rel=1e-6, abs=1e-6)


@pytest.mark.parametrize('K', [9, 10, 14])
def test_dense_prior_beyond_sixteen_parameters(amd, K, monkeypatch):
    """A correlated prior on 18 .. 28 parameters: the kernel stages P x P prior precisions in LDS (more entries than the
    workgroup has threads).  Single fit and bootstrap copies, against the oracle and the general path."""
    x, y, yerr, text, names, pt = bumps(K, 160, seed=40 + K, correlated=False)
    P = pt.size
    rng = np.random.default_rng(K)
    L = np.tril(0.05 * rng.standard_normal((P, P)), -1) + np.diag(0.3 + 0.2 * rng.random(P))
    pcov = L @ L.T
    model = amd.expr(text, names)
    kw = dict(data=(x, y, yerr), model=model, prior=(pt * 1.02, pcov), p0=pt * 1.05)
    one, f1, gen = both(amd, monkeypatch, **kw)
    assert f1 & ONE, 'the fit did not take the one-launch route'
    agree(one, gen)
    vs_oracle(one, oracle_of(text, names, **kw))
    monkeypatch.setenv('LSQAMD_ONE_LAUNCH_FIT', '1')
    bs1 = one.bootstrapped_fits(6, seed=2)
    assert bs1['rounds'] == 1
    monkeypatch.setenv('LSQAMD_ONE_LAUNCH_FIT', '0')
    bs0 = one.bootstrapped_fits(6, seed=2)
    assert np.max(np.abs(bs1['pmean'] - bs0['pmean']) / bs0['psdev']) < 1e-5 and np.allclose(bs1['chi2'], bs0['chi2'], rtol=1e-8)
    ref = oracle_of(text, names, data=(x, bs1['ymeans'][3], yerr), prior=(bs1['prior_means'][3], pcov), p0=one.pmean)
    assert np.all(np.abs(bs1['pmean'][3] - ref.pmean) <= 1e-6 * np.abs(ref.pmean) + 1e-5 * ref.psdev)
    assert bs1['chi2'][3] == pytest.approx(ref.chi2, rel=1e-6) and bs1['logGBF'][3] == pytest.approx(ref.logGBF, rel=1e-6, abs=1e-6)


@pytest.mark.parametrize('zero_copy', ['0', '1'])
def test_handoff_modes_agree_bit_for_bit(amd, zero_copy, monkeypatch):
    """LSQAMD_ZERO_COPY=0: nothing is published to host memory -- the host waits for the stream and copies the kernel's device
    block (also the fallback of a published block that does not verify).  Same kernel, same numbers, either way; pinned blocks
    poisoned before every launch (LSQAMD_POISON_PINNED is read once per process: here the fits simply repeat on one handle,
    whose block still holds the previous fit's words)."""
    monkeypatch.setenv('LSQAMD_ONE_LAUNCH_FIT', '1')
    x, y, sd, pt = curve(N=900, seed=12)
    model = amd.expr('a*exp(-b*x) + c*cos(d*x)', ['a', 'b', 'c', 'd'])
    kw = dict(data=(x, y, sd), model=model, prior=(pt, np.full(4, 1.0)))
    monkeypatch.setenv('LSQAMD_ZERO_COPY', '1')
    base = amd.nonlinear_fit(p0=pt * 1.3, **kw)
    monkeypatch.setenv('LSQAMD_ZERO_COPY', zero_copy)
    fits = []
    for i, (scale, maxit) in enumerate([(1.3, 1000), (1.1, 2), (0.8, 1000), (1.3, 1000), (1.2, 1), (1.3, 1000)]):
        fits.append(amd.nonlinear_fit(problem=base.problem, p0=pt * scale, maxit=maxit, **kw))
        assert flags(fits[-1]) & ONE
    for f in (fits[0], fits[3], fits[5]):
        assert f.nit == base.nit and np.array_equal(f.pmean, base.pmean) and np.array_equal(f.cov, base.cov) and f.chi2 == base.chi2
        assert f.stopping_criterion == base.stopping_criterion and f.logGBF == base.logGBF
    assert fits[1].nit == 2 and fits[1].stopping_criterion == 0 and fits[4].nit == 1 and fits[4].stopping_criterion == 0
    vs_oracle(fits[2], oracle_of('a*exp(-b*x) + c*cos(d*x)', ['a', 'b', 'c', 'd'], p0=pt * 0.8, **kw))


def test_general_small_path_record_is_verified(amd, monkeypatch):
    """The general path's 16-word LM record is polled from pinned memory too (api.hip wait_record): two hundred half steps with the
    audit on, both modes, same trajectory."""
    monkeypatch.setenv('LSQAMD_ONE_LAUNCH_FIT', '0')
    x, y, sd, pt = curve(N=2000, seed=5)
    model = amd.expr('a*exp(-b*x) + c*cos(d*x)', ['a', 'b', 'c', 'd'])
    kw = dict(data=(x, y, sd), model=model, prior=(pt, np.full(4, 1.0)), p0=pt * 1.5)
    monkeypatch.setenv('LSQAMD_ZERO_COPY', '1')
    a = [amd.nonlinear_fit(**kw) for _ in range(6)]
    monkeypatch.setenv('LSQAMD_ZERO_COPY', '0')
    b = amd.nonlinear_fit(**kw)
    for f in a:
        assert not flags(f) & ONE
        assert f.nit == b.nit and np.array_equal(f.pmean, b.pmean) and np.array_equal(f.cov, b.cov) and f.stopping_criterion == b.stopping_criterion
    vs_oracle(b, oracle_of('a*exp(-b*x) + c*cos(d*x)', ['a', 'b', 'c', 'd'], **kw))
