"""-m gpu: f3 of SURVEY.md 8 -- simulated / bootstrap refits as one device batch.

Mirrors tests/test_lsqfit.py:1551-1577 (test_fit_iter) and :714-770 (test_bootstrap) of the
reference, plus parity of individual copies against the single-fit engine and the oracle
(1e-6 relative, BASELINE.json north_star)."""
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


def test_batched_blocks_match_single_engine_and_oracle(amd):
    """Correlated (ragged) data blocks + per-fit priors in the lockstep engine."""
    from lsqfit_amd import synth
    d = synth.make_cosmix(N=333, P=10, seed=41, block=100, prior_corr=False)
    pm, ps = d['prior']
    B = 5
    widths = np.linspace(0.5, 2.0, B)
    pms = np.tile(pm, (B, 1))
    pss = ps[None, :] * widths[:, None]
    bf = amd.BatchedFits(d['model'], d['x'], d['ymean'], d['yerr'], pms, pss)
    out = bf.run(p0=d['p0'])
    for b in range(B):
        single = amd.nonlinear_fit(data=(d['x'], d['ymean'], d['yerr']), model=d['model'], prior=(pm, pss[b]))
        assert gu.relmax(out['pmean'][b], single.pmean) < 1e-6
        assert abs(out['chi2'][b] / single.chi2 - 1) < 1e-6
        assert gu.relmax(bf.cov(b), single.cov) < 1e-6
        assert abs(out['logGBF'][b] - single.logGBF) < 1e-6 * abs(single.logGBF) + 1e-6
        assert out['dof'] == single.dof
    dd = dict(d, prior=(pm, pss[2]))
    ref = gu.oracle_fit(dd)
    assert gu.relmax(out['pmean'][2], ref.pmean) < 1e-6
    assert gu.relmax(bf.cov(2), ref.cov) < 1e-6
    assert abs(out['chi2'][2] / ref.chi2 - 1) < 1e-6
    bf.close()


def test_simulated_copies_match_oracle(amd):
    """Each simulated copy refitted in the batch equals an oracle fit of the same inputs."""
    from lsqfit_amd import synth
    d = synth.make_cosmix(N=256, P=8, seed=42, block=64, prior_corr=False)
    fit = amd.nonlinear_fit(data=(d['x'], d['ymean'], d['yerr']), model=d['model'], prior=d['prior'])
    res = fit.simulated_fits(6, add_priornoise=True, seed=3)
    assert res.engine == 'batched' and res.pmean.shape == (6, 8)
    # model values at pexact through the C ABI
    assert gu.relmax(fit.problem.fcn(fit.pmean), gu.cosmix_fcn(d['x'], fit.pmean)) < 1e-12
    for k in (0, 5):
        dd = dict(d, ymean=res.ymeans[k], prior=(res.prior_means[k], d['prior'][1]))
        ref = gu.oracle_fit(dd, p0=fit.pmean)
        assert gu.relmax(res.pmean[k], ref.pmean) < 1e-6
        assert abs(res.chi2[k] / ref.chi2 - 1) < 1e-6
        assert gu.relmax(res.psdev[k], ref.psdev) < 1e-6
    # no-prior variant of the batch (dof = N - P)
    fit2 = amd.nonlinear_fit(data=(d['x'], d['ymean'], d['yerr']), model=d['model'], p0=d['p_true'])
    res2 = fit2.bootstrapped_fits(4, seed=9)
    assert res2.dof == 256 - 8 and res2.prior_means is None
    ref2 = ofit.nonlinear_fit(d['x'], res2.ymeans[1], gu.dense_cov(d['yerr'], 256), gu.cosmix_fcn,
                              p0=fit2.pmean, jac=gu.cosmix_jac, solver='cholesky')
    assert gu.relmax(res2.pmean[1], ref2.pmean) < 1e-6
    assert abs(res2.chi2[1] / ref2.chi2 - 1) < 1e-6


@pytest.mark.parametrize('svdcut', [1e-12, 0.5])
def test_fit_iter(amd, svdcut):
    """tests/test_lsqfit.py:1551-1577: y = ['2.1(1.2)','1.7(5.2)','2.2(3)','3.2(1.5)','1.9(2)'],
    y[1:] = (y[1:] + y[:1]) / 2, fcn = constant, prior 2(1)."""
    m = np.array([2.1, 1.7, 2.2, 3.2, 1.9])
    s = np.array([1.2, 5.2, 0.3, 1.5, 0.2])
    A = np.eye(5)
    A[1:, 0] = 0.5
    A[1:, 1:] = 0.5 * np.eye(4)
    ymean, ycov = A @ m, A @ np.diag(s ** 2) @ A.T
    model = amd.expr('p0 + 0*x', ['p0'])
    fit = amd.nonlinear_fit(data=(np.zeros(5), ymean, ycov), model=model, prior=([2.0], [1.0]), svdcut=svdcut)
    N = 100
    res = fit.simulated_fits(N, seed=11)
    np.testing.assert_allclose(res.psdev[:, 0], fit.psdev[0], rtol=1e-7)
    assert np.all(res.prior_means == 2.0)
    assert abs(np.average(res.pmean[:, 0]) - fit.pmean[0]) < 5 * fit.psdev[0] / N ** 0.5
    assert np.all(res.stopping_criterion > 0)
    res = fit.simulated_fits(N, add_priornoise=True, seed=12)
    pmn = res.prior_means[:, 0]
    assert abs(np.average(pmn) - 2.0) < 5 * 1.0
    assert abs(np.std(pmn) / 1.0 - 1) < 5. / N ** 0.5
    # with prior noise chi2/dof ~ 1 +- sqrt(2/dof) on average (docstring :1427-1431)
    assert abs(np.average(res.chi2) / res.dof - 1) < 5 * np.sqrt(2.0 / res.dof / N)


def _bin(v):
    return (v[:-1] + v[1:]) / 2.


@pytest.mark.parametrize('case', ['plain', 'binned', 'binned_y', 'binned_p', 'noprior', 'noprior_binned'])
def test_bootstrap(amd, case):
    """tests/test_lsqfit.py:714-770: fcn = p**2; the bootstrap distribution of avg(p**2) agrees
    with the fit (10 sigma of the mean, as there)."""
    rng = np.random.default_rng(50)
    ny = 3
    y0 = 4.0 + 0.25 * rng.standard_normal(ny)
    p20 = 4.0 + 0.25 * rng.standard_normal(ny)
    ycov, y2cov = np.diag(np.full(ny, 0.25 ** 2)), np.diag(np.full(ny, 0.25 ** 2))
    # prior on p = sqrt(p2): mean sqrt(p2), sdev 0.25 / (2 sqrt(p2))
    pmean0 = np.sqrt(p20)
    Jp = np.diag(0.5 / pmean0)
    pcov0 = Jp @ y2cov @ Jp.T
    Bm = np.zeros((ny - 1, ny))
    for i in range(ny - 1):
        Bm[i, i] = Bm[i, i + 1] = 0.5
    Tr = np.eye(ny)[:-1]
    ysel = Bm if case in ('binned', 'binned_y', 'noprior_binned') else Tr
    psel = Bm if case in ('binned', 'binned_p') else Tr
    ymean, yc = ysel @ y0, ysel @ ycov @ ysel.T
    model = amd.expr('s0*p0**2 + s1*p1**2', ['p0', 'p1'], xnames=('s0', 's1'))
    x = np.eye(2)
    if case.startswith('noprior'):
        fit = amd.nonlinear_fit(data=(x, ymean, yc), model=model, p0=np.sqrt(ymean))
    else:
        pm, pc = psel @ pmean0, psel @ pcov0 @ psel.T
        fit = amd.nonlinear_fit(data=(x, ymean, yc), model=model, prior=(pm, pc))
    nbs = 100
    res = fit.bootstrapped_fits(nbs, seed=77)
    assert res.engine == 'batched'          # correlated (dense) priors are shared by the batch too
    ok = res.status == 0
    assert ok.sum() >= nbs - 2
    bs = np.median(np.mean(res.pmean[ok] ** 2, axis=1))
    g = 2 * fit.pmean / 2.0                                  # gradient of avg(p**2)
    fit_ans, fit_sd = np.mean(fit.pmean ** 2), np.sqrt(g @ fit.cov @ g)
    assert abs(bs - fit_ans) < 10. * fit_sd / nbs ** 0.5


def test_config5_shape_correlated_bootstrap(amd):
    """128 simulated copies (data and prior means redrawn around the fit) of a (4096, 512) fit
    with 256-row correlated blocks in one batch."""
    import time
    from lsqfit_amd import synth
    d = synth.make_cosmix(N=4096, P=512, seed=20264, block=256, prior_corr=False)
    fit = amd.nonlinear_fit(data=(d['x'], d['ymean'], d['yerr']), model=d['model'], prior=d['prior'])
    t0 = time.perf_counter()
    res = fit.simulated_fits(128, add_priornoise=True, seed=5, covariance=False)
    dt = time.perf_counter() - t0
    print('config-5-shaped correlated simulated fits: 128 copies, %d rounds, device %.1f ms, wall %.2f s'
          % (res['rounds'], res['device_ms'], dt))
    assert res.engine == 'batched'
    assert np.all(res.status == 0) and np.all(res.stopping_criterion > 0)
    # both data and prior means redrawn: chi2/dof ~ 1 (exactly 1 +- sqrt(2/dof) for a linear
    # model -- test_fit_iter holds that; the frequencies make this one mildly nonlinear)
    assert abs(np.mean(res.chi2) / res.dof - 1) < 0.05
    # spread of the copies' parameters = the fit's errors (Gaussian limit)
    z = (res.pmean - fit.pmean) / fit.psdev
    assert 0.9 < z.std() < 1.1


def test_dense_prior_batch_matches_single_engine_and_oracle(amd):
    """Correlated data blocks AND a correlated prior in the lockstep engine: every simulated copy
    equals a single-engine / oracle fit of the same inputs."""
    from lsqfit_amd import synth
    d = synth.make_cosmix(N=512, P=32, seed=95, block=128, prior_corr=True)
    fit = amd.nonlinear_fit(data=(d['x'], d['ymean'], d['yerr']), model=d['model'], prior=d['prior'])
    res = fit.simulated_fits(5, add_priornoise=True, seed=4)
    assert res.engine == 'batched' and np.all(res.status == 0)
    for k in (0, 4):
        dd = dict(d, ymean=res.ymeans[k], prior=(res.prior_means[k], d['prior'][1]))
        single = amd.nonlinear_fit(data=(d['x'], res.ymeans[k], d['yerr']), model=d['model'],
                                   prior=(res.prior_means[k], d['prior'][1]), p0=fit.pmean)
        assert gu.relmax(res.pmean[k], single.pmean) < 1e-9
        assert abs(res.chi2[k] / single.chi2 - 1) < 1e-9
        assert abs(res.logGBF[k] - single.logGBF) < 1e-8 * abs(single.logGBF)
        assert res.nit[k] == single.nit
        ref = gu.oracle_fit(dd, p0=fit.pmean)
        assert gu.relmax(res.pmean[k], ref.pmean) < 1e-6
        assert gu.relmax(res.psdev[k], ref.psdev) < 1e-6
        assert abs(res.chi2[k] / ref.chi2 - 1) < 1e-6


def _cross_problem(seed=31, N=24, P=3):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((N + P, N + P))
    full = (A @ A.T + (N + P) * np.eye(N + P)) * 1e-4
    x = np.linspace(0.1, 2.0, N)
    truth = np.array([1.0, 0.7, 0.3])
    dev = np.linalg.cholesky(full) @ rng.standard_normal(N + P)
    y = truth[0] * np.exp(-truth[1] * x) + truth[2] + dev[:N]
    return x, y, truth + dev[N:], full, N, P


def _cross_fcn(xx, p):
    from oracle import dual
    return p[0] * dual.exp(-p[1] * xx) + p[2]


def test_resampled_copies_of_a_fit_correlated_with_its_prior(amd):
    """Data correlated with the prior (examples/y-noerr.py; a declared limit of rounds 2-3).  What gvar.bootstrap_iter is handed
    decides what a copy's covariance is (src/lsqfit/__init__.py:1519-1543,:1607-1626): bootstrap copies and simulated copies with
    prior noise keep the full joint covariance -- refits of the same joint problem with new means; simulated copies without prior
    noise get data that are no longer correlated with the prior.  Copies against independent fits of their inputs and the oracle."""
    x, y, pm, full, N, P = _cross_problem()
    model = amd.expr('b1*exp(-b2*x) + b3', ['b1', 'b2', 'b3'])
    Cyy, Cpp, Cyp = full[:N, :N], full[N:, N:], full[:N, N:]
    fit = amd.nonlinear_fit(data=(x, y, Cyy), model=model, prior=(pm, Cpp), cross=Cyp, tol=1e-10)
    extra = [((i, N + j), full[i, N + j]) for i in range(N) for j in range(P)]
    for res in (fit.bootstrapped_fits(5, seed=4), fit.simulated_fits(5, add_priornoise=True, seed=5)):
        assert res.pmean.shape == (5, P) and res.ymeans.shape == (5, N) and res.prior_means.shape == (5, P)
        assert np.std(res.prior_means[:, 0]) > 0 and np.std(res.ymeans[:, 0]) > 0
        for k in (0, 4):
            single = amd.nonlinear_fit(data=(x, res.ymeans[k], Cyy), model=model, prior=(res.prior_means[k], Cpp), cross=Cyp,
                                       p0=res.get('pexact', fit.pmean), tol=1e-10)
            assert gu.relmax(res.pmean[k], single.pmean) < 1e-9 and res.chi2[k] == pytest.approx(single.chi2, rel=1e-9)
            assert res.logGBF[k] == pytest.approx(single.logGBF, rel=1e-9, abs=1e-9) and gu.relmax(res.cov[k], single.cov) < 1e-8
            ref = ofit.nonlinear_fit(x, res.ymeans[k], Cyy, _cross_fcn, prior_mean=res.prior_means[k], prior_err=Cpp, extra_cov=extra,
                                     p0=res.get('pexact', fit.pmean), tol=1e-10, solver='cholesky')
            assert np.all(np.abs(res.pmean[k] - ref.pmean) <= 1e-6 * np.abs(ref.pmean) + 1e-6 * ref.psdev)
            assert res.chi2[k] == pytest.approx(ref.chi2, rel=1e-6) and gu.relmax(res.cov[k], ref.cov) < 1e-6
    # the handle is back where it was: the original fit again, bit for bit
    again = amd.nonlinear_fit(data=(x, y, Cyy), model=model, prior=(pm, Cpp), problem=fit.problem, tol=1e-10)
    assert np.array_equal(again.pmean, fit.pmean) and again.chi2 == fit.chi2
    # simulated copies without prior noise: y alone went through bootstrap_iter -- no cross terms in the refits, prior means fixed
    res = fit.simulated_fits(4, seed=6)
    assert np.all(res.prior_means == pm[None, :]) and res.engine == 'batched'
    for k in (0, 3):
        ref = ofit.nonlinear_fit(x, res.ymeans[k], Cyy, _cross_fcn, prior_mean=pm, prior_err=Cpp, p0=fit.pmean, tol=1e-10, solver='cholesky')
        assert np.all(np.abs(res.pmean[k] - ref.pmean) <= 1e-6 * np.abs(ref.pmean) + 1e-6 * ref.psdev)
        assert res.chi2[k] == pytest.approx(ref.chi2, rel=1e-6)
    # the scatter of the copies is the fit's own covariance (loosely: 200 copies)
    many = fit.bootstrapped_fits(200, seed=9)
    assert np.all(np.abs(np.std(many.pmean, axis=0) / fit.psdev - 1) < 0.25)


def test_eps_regulation_with_data_correlated_with_the_prior(amd):
    """eps (gvar.regulate's other mode, src/lsqfit/__init__.py:240-245) on the joint vector: against the oracle's restatement."""
    x, y, pm, full, N, P = _cross_problem(seed=33)
    model = amd.expr('b1*exp(-b2*x) + b3', ['b1', 'b2', 'b3'])
    extra = [((i, N + j), full[i, N + j]) for i in range(N) for j in range(P)]
    fit = amd.nonlinear_fit(data=(x, y, full[:N, :N]), model=model, prior=(pm, full[N:, N:]), cross=full[:N, N:], svdcut=None, eps=1e-3,
                            tol=1e-10)
    ref = ofit.nonlinear_fit(x, y, full[:N, :N], _cross_fcn, prior_mean=pm, prior_err=full[N:, N:], extra_cov=extra, svdcut=None, eps=1e-3,
                             tol=1e-10, solver='cholesky')
    assert fit.dof == ref.dof and gu.relmax(fit.pmean, ref.pmean) < 1e-6 and fit.chi2 == pytest.approx(ref.chi2, rel=1e-6)
    assert gu.relmax(fit.cov, ref.cov) < 1e-6 and fit.logGBF == pytest.approx(ref.logGBF, rel=1e-6, abs=1e-6)
    plain = amd.nonlinear_fit(data=(x, y, full[:N, :N]), model=model, prior=(pm, full[N:, N:]), cross=full[:N, N:], tol=1e-10)
    assert abs(plain.chi2 - fit.chi2) > 1e-6 * fit.chi2           # (the regulation did something)


def test_noise_with_data_correlated_with_the_prior(amd):
    """noise= (src/lsqfit/__init__.py:247-256,:535-536,:1896) for the joint vector: noise[1] shifts the prior means by a draw from
    the prior's covariance before the fit, noise[0] adds a draw from what the regulation added to concat(y, prior).  gvar's random
    stream is not reproduced (unpinned): the fit with noise must BE the plain fit of the shifted means, bit for bit."""
    x, y, pm, full, N, P = _cross_problem(seed=35)
    model = amd.expr('b1*exp(-b2*x) + b3', ['b1', 'b2', 'b3'])
    Cyy, Cpp, Cyp = full[:N, :N], full[N:, N:], full[:N, N:]
    plain = amd.nonlinear_fit(data=(x, y, Cyy), model=model, prior=(pm, Cpp), cross=Cyp, tol=1e-10)
    pn = amd.nonlinear_fit(data=(x, y, Cyy), model=model, prior=(pm, Cpp), cross=Cyp, tol=1e-10, noise=(False, True), rng=7)
    shifted = pn.whitening.prior_mean_host
    assert np.all(shifted != pm) and np.all(np.abs(shifted - pm) < 6 * np.sqrt(np.diag(Cpp)))
    same = amd.nonlinear_fit(data=(x, y, Cyy), model=model, prior=(shifted, Cpp), cross=Cyp, tol=1e-10, p0=pm)
    assert np.array_equal(pn.pmean, same.pmean) and pn.chi2 == same.chi2 and pn.logGBF == same.logGBF
    assert not np.array_equal(pn.pmean, plain.pmean)
    # nothing is regulated at the default svdcut here: data noise adds nothing; with a binding svdcut it moves the joint means
    dn = amd.nonlinear_fit(data=(x, y, Cyy), model=model, prior=(pm, Cpp), cross=Cyp, tol=1e-10, noise=(True, False), rng=7)
    assert np.array_equal(dn.pmean, plain.pmean) and dn.svdn == 0
    cut = amd.nonlinear_fit(data=(x, y, Cyy), model=model, prior=(pm, Cpp), cross=Cyp, tol=1e-10, svdcut=0.3)
    cutn = amd.nonlinear_fit(data=(x, y, Cyy), model=model, prior=(pm, Cpp), cross=Cyp, tol=1e-10, svdcut=0.3, noise=(True, False), rng=7)
    assert cut.svdn > 0 and cutn.svdn == cut.svdn and not np.array_equal(cutn.pmean, cut.pmean)
    assert np.all(np.abs(cutn.pmean - cut.pmean) < 8 * cut.psdev)


def test_batched_copies_keep_the_fits_scaler_and_factors(amd):
    """The reference refits every copy with the ORIGINAL fit's fitter arguments (src/lsqfit/__init__.py:1457-1459,:1603-1604).
    scaler / factor_up / factor_down travel into the lockstep engine (same trajectories as single fits with those arguments:
    same nit, same bits to 1e-10); solver='qr' or another avmax is not what the engine does -> copy by copy with the fit's own
    arguments.  (Round-5 advisor finding: the batch ran such copies with the defaults.)"""
    from lsqfit_amd import synth
    d = synth.make_cosmix(N=300, P=8, seed=77, block=0, prior_corr=False)
    data = (d['x'], d['ymean'], d['yerr'])
    kw = dict(scaler='levenberg', factor_up=7.0, factor_down=1.5)
    fit = amd.nonlinear_fit(data=data, model=d['model'], prior=d['prior'], **kw)
    res = fit.bootstrapped_fits(4, seed=3)
    assert res.engine == 'batched'
    dflt = amd.nonlinear_fit(data=data, model=d['model'], prior=d['prior']).bootstrapped_fits(4, seed=3)
    assert np.array_equal(res.ymeans, dflt.ymeans)
    differs = False
    for k in range(4):
        single = amd.nonlinear_fit(data=(d['x'], res.ymeans[k], d['yerr']), model=d['model'], prior=(res.prior_means[k], d['prior'][1]),
                                   p0=fit.pmean, **kw)
        assert int(res.nit[k]) == single.nit
        assert gu.relmax(res.pmean[k], single.pmean) < 1e-10 and res.chi2[k] == pytest.approx(single.chi2, rel=1e-10)
        differs = differs or int(res.nit[k]) != int(dflt.nit[k]) or not np.array_equal(res.pmean[k], dflt.pmean[k])
    assert differs                      # (the arguments do change the trajectories: the check above is not vacuous)
    fq = amd.nonlinear_fit(data=data, model=d['model'], prior=d['prior'], solver='qr')
    assert fq.bootstrapped_fits(2, seed=3).engine.startswith('sequential')
    fa = amd.nonlinear_fit(data=data, model=d['model'], prior=d['prior'], avmax=0.5)
    assert fa.bootstrapped_fits(2, seed=3).engine.startswith('sequential')
