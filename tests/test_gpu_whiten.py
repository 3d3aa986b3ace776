"""-m gpu: the whitening set-up on the device (lsqamd_whiten_blocks through lsqfit_amd.Whitening):
the identities tests/test_host_whitening.py checks for the LAPACK route (and the reference asserts at
tests/test_lsqfit.py:923-943: W^T W = inv(C_reg), logdet = log det C_reg), re-run against the
device-built weights; the decision whether the svdcut floor binds; parity of whole fits."""
import time

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


def spd(rng, B, cond):
    Q, _ = np.linalg.qr(rng.standard_normal((B, B)))
    lam = np.geomspace(1.0, 1.0 / cond, B)
    C = (Q * lam) @ Q.T
    s = rng.uniform(0.1, 10.0, B)
    return (C + C.T) / 2 * np.outer(s, s)


@pytest.mark.parametrize('sizes', [[8], [100, 100, 37], [128, 256], [300]])
def test_device_weights_identities(amd, sizes):
    from lsqfit_amd.whiten import regulate_blocks
    rng = np.random.default_rng(sum(sizes))
    covs = [spd(rng, B, 1e4) for B in sizes]
    dev = regulate_blocks(covs, 1e-12, want_prec=True, engine='device')
    host = regulate_blocks(covs, 1e-12, engine='host')
    for c, d, h in zip(covs, dev, host):
        B = c.shape[0]
        assert 'Wt_dev' in d and d['tri'] == 1 and d['modes'] == B and d['nmod'] == 0
        Wt = d['Wt']
        assert np.all(np.tril(Wt, -1) == 0.0)                       # upper triangular, as promised
        ic = np.linalg.inv(c)
        assert gu.relmax(Wt @ Wt.T, ic) < 1e-9                      # W^T W = inv(C)
        assert gu.relmax(d['prec'], ic) < 1e-9
        assert np.array_equal(d['prec'], d['prec'].T)
        assert d['logdet'] == pytest.approx(np.linalg.slogdet(c)[1], rel=1e-11, abs=1e-9)
        assert d['logdet'] == pytest.approx(h['logdet'], rel=1e-11, abs=1e-9)
        sd = np.sqrt(np.diag(c))
        w = np.linalg.eigvalsh(c / np.outer(sd, sd))
        lo, hi = d['lam_bounds']
        assert lo <= w[0] * (1 + 1e-9) and hi >= w[-1] * (1 - 1e-9)   # the bounds ARE bounds
        assert gu.relmax(d['S'] @ d['S'].T, c) < 1e-10


def test_floor_decision_and_fallbacks(amd):
    """cond 1e4 with svdcut 1e-2: the floor binds -> eigen route (gvar's), same numbers as the host
    engine; cond 3e9 with svdcut 1e-12: the rigorous bounds cannot decide, the inverse iteration
    does; a non-positive-definite block is an error, not garbage."""
    from lsqfit_amd.whiten import regulate_blocks
    rng = np.random.default_rng(4)
    c = spd(rng, 64, 1e4)
    d = regulate_blocks([c], 1e-2, engine='device')[0]
    h = regulate_blocks([c], 1e-2, engine='host')[0]
    assert d['tri'] == 0 and d['nmod'] == h['nmod'] > 0 and d['logdet'] == pytest.approx(h['logdet'], rel=1e-12)
    assert gu.relmax(d['Wt'] @ d['Wt'].T, h['Wt'] @ h['Wt'].T) < 1e-9
    c2 = spd(rng, 200, 3e9)
    d2 = regulate_blocks([c2], 1e-12, engine='device')[0]
    assert d2['tri'] == 1 and d2['nmod'] == 0
    lo, hi = d2['lam_bounds']
    sd = np.sqrt(np.diag(c2))
    w = np.linalg.eigvalsh(c2 / np.outer(sd, sd))
    assert lo <= w[0] * 1.001
    c3 = spd(rng, 200, 1e14)                                           # on the floor of svdcut = 1e-12
    d3 = regulate_blocks([c3], 1e-12, engine='device')[0]
    assert d3['tri'] == 0 and d3['nmod'] >= 1
    bad = c.copy()
    bad[0, 1] = bad[1, 0] = 2.0 * np.sqrt(c[0, 0] * c[1, 1])
    with pytest.raises(ValueError, match='positive definite'):
        regulate_blocks([bad], 0.0, engine='device')


@pytest.mark.parametrize('shape', [dict(N=512, P=32, block=64, prior_corr=True),
                                   dict(N=300, P=16, block=100, prior_corr=True),
                                   dict(N=256, P=64, block=256, prior_corr=False)])
def test_fits_with_device_whitening_match_host_whitening_and_oracle(amd, shape):
    from lsqfit_amd import synth
    d = synth.make_cosmix(seed=91, **shape)
    wd = amd.Whitening(d['ymean'], d['yerr'], *d['prior'], engine='device')
    whh = amd.Whitening(d['ymean'], d['yerr'], *d['prior'], engine='host')
    assert wd.nchiv == whh.nchiv and wd.nmod == whh.nmod and wd.nblocks == whh.nblocks
    assert wd.logdet == pytest.approx(whh.logdet, rel=1e-11)
    assert (wd.prior_prec_dev is not None) == bool(shape['prior_corr'])
    assert gu.relmax(wd.prior_prec, whh.prior_prec) < 1e-9
    kw = dict(data=(d['x'], d['ymean'], d['yerr']), model=d['model'], prior=d['prior'])
    fd = amd.nonlinear_fit(problem=amd.DeviceProblem(d['model'], d['x'], wd), **kw)
    fh = amd.nonlinear_fit(problem=amd.DeviceProblem(d['model'], d['x'], whh), **kw)
    assert gu.relmax(fd.pmean, fh.pmean) < 1e-9 and gu.relmax(fd.cov, fh.cov) < 1e-8
    assert fd.chi2 == pytest.approx(fh.chi2, rel=1e-9) and fd.logGBF == pytest.approx(fh.logGBF, rel=1e-9)
    ref = gu.oracle_fit(d, solver='cholesky')
    assert gu.relmax(fd.pmean, ref.pmean) < 1e-6 and gu.relmax(fd.cov, ref.cov) < 1e-6
    assert fd.logGBF == pytest.approx(ref.logGBF, rel=1e-8, abs=1e-6)
    # fit.f / fit.J assemble prior rows from the lazily fetched weights
    assert fd.residuals.size == wd.nchiv and fd.J.shape == (wd.nchiv, shape['P'])
    assert float(fd.residuals @ fd.residuals) == pytest.approx(fd.chi2, rel=1e-9)


def test_config3_setup_time(amd):
    """BASELINE.json configs[2] shape: ONE dense 8192 x 8192 data block + a dense 1024 x 1024 prior.
    The factorisations that took 24 s in LAPACK on the host run on the device."""
    from lsqfit_amd import synth
    d = gu.config3_problem()
    amd.Whitening(d['ymean'][:256], dict(sdev=d['yerr']['sdev'][:256], blocks=[(0, d['yerr']['blocks'][0][1][:256, :256])]))  # warm-up
    t0 = time.perf_counter()
    wh = amd.Whitening(d['ymean'], d['yerr'], *d['prior'])
    pr = amd.DeviceProblem(d['model'], d['x'], wh)
    dt = time.perf_counter() - t0
    print('config 3 set-up (whitening + problem): %.3f s' % dt)
    assert wh.blocks[0]['tri'] == 1 and 'Wt_dev' in wh.blocks[0]
    assert dt < 6.0            # (0.03 - 0.2 s on the pool's boxes; the host path it replaced took 23.9 s)
    # log det against LAPACK's Cholesky of the same block
    c = d['yerr']['blocks'][0][1]
    L = np.linalg.cholesky(c)
    assert wh.logdet_data == pytest.approx(2.0 * np.sum(np.log(np.diag(L))), rel=1e-10)
    pr.close()
