"""-m gpu: data (and prior) whose covariance components interleave -- rows 1, 4, 9, 10 correlated with
each other and with nothing between them.  gvar's block search returns index sets
(tests/test_lsqfit.py:1011-1012); the whitening makes each set contiguous by a row permutation and
the device problem follows it.  Checked against the oracle, which works on index sets directly."""
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


def _scatter_blocks(rng, n, sets, floor, deficit=0):
    assert len(set(sum(map(list, sets), []))) == sum(len(k) for k in sets)       # disjoint index sets
    cov = np.diag(rng.uniform(0.05, 0.2, n) ** 2)
    for idx in sets:
        B = len(idx)
        sd = np.sqrt(cov[idx, idx])
        A = rng.standard_normal((B, B - deficit))
        corr = A @ A.T + floor * B * np.eye(B)
        d = np.sqrt(np.diag(corr))
        cov[np.ix_(idx, idx)] = corr / np.outer(d, d) * np.outer(sd, sd)
    return cov


@pytest.mark.parametrize('svdcut', [1e-12, 1e-2])
def test_fit_with_interleaved_components_matches_oracle(amd, svdcut):
    rng = np.random.default_rng(23)
    K, N = 3, 96
    P = 2 * K
    x = np.sort(rng.uniform(0.0, 3.0, N))
    ptrue = np.concatenate([[1.0, 0.6, 0.3], [1.1, 2.3, 3.9]])
    ysets = [list(range(1, N, 7)), [2, 3, 52, 53, 90], list(range(5, 40, 11))]
    ycov = _scatter_blocks(rng, N, ysets, 0.02 if svdcut < 1e-6 else 1e-4)
    psets = [[0, 3], [1, 5]]
    pcov = _scatter_blocks(rng, P, psets, 0.5) * 25.0
    f = gu.cosmix_fcn(x, ptrue)
    ymean = f + np.linalg.cholesky(ycov) @ rng.standard_normal(N)
    pmean = ptrue + 0.05 * rng.standard_normal(P)
    model = amd.cosmix(K)
    fit = amd.nonlinear_fit(data=(x, ymean, ycov), model=model, prior=(pmean, pcov), p0=pmean, svdcut=svdcut,
                            tol=1e-10)
    wh = fit.whitening
    assert wh.perm is not None and sorted(b['size'] for b in wh.blocks) == sorted(len(s) for s in ysets)
    ref = ofit.nonlinear_fit(x, ymean, ycov, gu.cosmix_fcn, prior_mean=pmean, prior_err=pcov, p0=pmean, tol=1e-10,
                             svdcut=svdcut, jac=gu.cosmix_jac, solver='cholesky')
    assert (fit.dof, wh.nmod, wh.nblocks) == (ref.dof, ref.pdf.nmod, ref.pdf.nblocks)
    assert fit.chi2 == pytest.approx(ref.chi2, rel=1e-8)
    assert gu.relmax(fit.pmean, ref.pmean) < 1e-6 and gu.relmax(fit.cov, ref.cov) < 1e-6
    assert fit.logGBF == pytest.approx(ref.logGBF, rel=1e-8, abs=1e-6)
    # row-ordered outputs come back in the caller's order
    np.testing.assert_allclose(fit.problem.fcn(fit.pmean), gu.cosmix_fcn(x, fit.pmean), rtol=1e-12, atol=1e-13)
    D, Dref = fit.dp_dinputs(), ofit.dp_dinputs(ref)
    assert D.shape == Dref.shape and gu.relmax(D, Dref) < 1e-6
    pts = fit.pmean + 1e-3 * rng.standard_normal((4, P))
    np.testing.assert_allclose(fit.dchi2(pts), [ofit.dchi2(ref, q) for q in pts], rtol=1e-6)
    # resampled copies run in the whitening's row order; their data come back in the caller's
    for res in (fit.bootstrapped_fits(4, seed=5), fit.simulated_fits(4, seed=6)):
        assert res.ymeans.shape == (4, N)
        for k in (0, 3):
            single = amd.nonlinear_fit(data=(x, res.ymeans[k], ycov), model=model, prior=(res.prior_means[k], pcov),
                                       p0=fit.pmean, svdcut=svdcut, tol=fit.tol)
            assert gu.relmax(res.pmean[k], single.pmean) < 1e-6 and res.chi2[k] == pytest.approx(single.chi2, rel=1e-7)
    # a shard is a range of the REORDERED rows and must hold whole components (sharded fits: tests/test_gpu_dist2.py)
    with pytest.raises(ValueError, match='cuts through a covariance block'):
        amd.DeviceProblem(model, x, wh, rows=(0, 10))           # rows 1..14 of the reordered data are one component
    part = amd.DeviceProblem(model, x, wh, rows=(0, 20))        # [0], the 14-row component, the 5-row one
    assert part.N == 20
    np.testing.assert_allclose(part.fcn(fit.pmean), gu.cosmix_fcn(x[wh.perm[:20]], fit.pmean), rtol=1e-12, atol=1e-13)
    part.close()


def test_fit_with_eps_regulation_matches_oracle(amd):
    """eps instead of svdcut (src/lsqfit/__init__.py:240-245): C -> C + eps ||corr||_inf D^2 on every
    correlated block, data and prior; every block then takes the device's Cholesky route.  Unpinned in the
    reference -- the device fit is held to the oracle's restatement of gvar's documented rule."""
    rng = np.random.default_rng(29)
    K, N = 2, 64
    P = 2 * K
    x = np.sort(rng.uniform(0.0, 3.0, N))
    ptrue = np.array([1.0, 0.5, 1.3, 3.1])
    ycov = _scatter_blocks(rng, N, [list(range(0, N, 5)), [1, 2, 3], [33, 36, 39, 42, 46, 49]], 0.0, deficit=1)
    pcov = _scatter_blocks(rng, P, [[0, 2]], 0.5) * 25.0
    ymean = gu.cosmix_fcn(x, ptrue) + 0.05 * rng.standard_normal(N)
    pmean = ptrue + 0.05 * rng.standard_normal(P)
    eps = 1e-2
    fit = amd.nonlinear_fit(data=(x, ymean, ycov), model=amd.cosmix(K), prior=(pmean, pcov), p0=pmean, eps=eps, tol=1e-10)
    assert fit.svdcut is None and fit.eps == eps
    assert all(b['tri'] == 1 and 'Wt_dev' in b for b in fit.whitening.blocks)        # built on the device
    ref = ofit.nonlinear_fit(x, ymean, ycov, gu.cosmix_fcn, prior_mean=pmean, prior_err=pcov, p0=pmean, tol=1e-10,
                             svdcut=None, eps=eps, jac=gu.cosmix_jac, solver='cholesky')
    assert (fit.dof, fit.svdn) == (ref.dof, ref.svdn) and fit.svdn > 0
    assert fit.chi2 == pytest.approx(ref.chi2, rel=1e-8)
    assert gu.relmax(fit.pmean, ref.pmean) < 1e-6 and gu.relmax(fit.cov, ref.cov) < 1e-6
    assert fit.logGBF == pytest.approx(ref.logGBF, rel=1e-8, abs=1e-6)
    # the blocks are singular to rounding: without regulation they are refused
    with pytest.raises(ValueError):
        amd.nonlinear_fit(data=(x, ymean, ycov), model=amd.cosmix(K), prior=(pmean, pcov), p0=pmean, svdcut=0.0)


def test_batched_fits_built_directly_on_interleaved_data(amd):
    """A BatchedFits handed data whose covariance components interleave (a declared limit of rounds 2-3): it reorders x and every
    vector of data means itself; fits of a prior-width sweep against single fits and the oracle."""
    rng = np.random.default_rng(29)
    K, N = 2, 60
    P = 2 * K
    x = np.sort(rng.uniform(0.0, 3.0, N))
    ptrue = np.array([1.0, 0.5, 1.2, 2.7])
    ycov = _scatter_blocks(rng, N, [list(range(0, N, 9)), [4, 5, 33, 34]], 0.02)
    ymean = gu.cosmix_fcn(x, ptrue) + np.linalg.cholesky(ycov) @ rng.standard_normal(N)
    B = 4
    pms = np.tile(ptrue * 1.02, (B, 1))
    pss = np.outer(np.linspace(0.5, 2.0, B), np.array([0.5, 0.5, 0.2, 0.2]))
    ymeans = ymean[None, :] + 0.01 * rng.standard_normal((B, N))
    bf = amd.BatchedFits(amd.cosmix(K), x, ymeans, ycov, pms, pss)
    out = bf.run(p0=pms)
    for b in range(B):
        single = amd.nonlinear_fit(data=(x, ymeans[b], ycov), model=amd.cosmix(K), prior=(pms[b], pss[b]), p0=pms[b])
        assert gu.relmax(out['pmean'][b], single.pmean) < 1e-6 and abs(out['chi2'][b] / single.chi2 - 1) < 1e-6
        assert gu.relmax(bf.cov(b), single.cov) < 1e-6 and abs(out['logGBF'][b] - single.logGBF) < 1e-6 * abs(single.logGBF) + 1e-6
    ref = ofit.nonlinear_fit(x, ymeans[2], ycov, gu.cosmix_fcn, prior_mean=pms[2], prior_err=pss[2], p0=pms[2], jac=gu.cosmix_jac,
                             solver='cholesky')
    assert gu.relmax(out['pmean'][2], ref.pmean) < 1e-6 and gu.relmax(bf.cov(2), ref.cov) < 1e-6
    assert abs(out['chi2'][2] / ref.chi2 - 1) < 1e-6
    bf.close()
