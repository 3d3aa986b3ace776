"""-m gpu: whitening product of ONE large dense block (BASELINE config 3's shape: a fully correlated data set).
With few long tile rows the triangular product runs both halves of every tile row's K-range as separate
workgroups (gemm_tn_f64.hip gemm_tn_wants_tri_halves; api.hip whiten_jacobian adds the scratch slab back).
Checked against the one-workgroup-per-row-pair launch (LSQAMD_TRI_HALVES=0, read per call)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def amd():
    import lsqfit_amd
    from lsqfit_amd import _lib
    _lib.load()
    return lsqfit_amd


@pytest.mark.parametrize('N,P,block', [(2048, 128, 2048), (1024, 256, 1024), (4096, 128, 2048)])
def test_whitened_jacobian_equals_the_unsplit_product(amd, N, P, block, monkeypatch):
    from lsqfit_amd import synth
    monkeypatch.setenv('LSQAMD_FUSED_JACOBIAN', '0')     # (the product under test reads the raw Jacobian)
    d = synth.make_cosmix(N=N, P=P, seed=N + P, block=block, prior_corr=False)
    wh = amd.Whitening(d['ymean'], d['yerr'], *d['prior'])
    p = d['p0'] * (1.0 + 0.03 * np.random.default_rng(2).standard_normal(P))
    got = {}
    for mode in ('1', '0'):
        monkeypatch.setenv('LSQAMD_TRI_HALVES', mode)
        pr = amd.DeviceProblem(d['model'], d['x'], wh)
        chi2 = pr.normal(p)
        got[mode] = (chi2, pr.get_grad(), pr.get_jtj(), pr.get_J_data())
        pr.close()
    c1, g1, A1, J1 = got['1']
    c0, g0, A0, J0 = got['0']
    tol = 1e-12 * np.max(np.abs(J0))
    assert np.max(np.abs(J1 - J0)) <= tol
    assert abs(c1 - c0) <= 1e-12 * c0
    assert np.max(np.abs(g1 - g0)) <= 1e-11 * np.max(np.abs(g0))
    assert np.max(np.abs(A1 - A0)) <= 1e-11 * np.max(np.abs(A0))


def test_fit_of_one_dense_block_matches(amd, monkeypatch):
    from lsqfit_amd import synth
    d = synth.make_cosmix(N=2048, P=128, seed=4, block=2048, prior_corr=False)
    kw = dict(data=(d['x'], d['ymean'], d['yerr']), model=d['model'], prior=d['prior'], p0=d['p0'])
    monkeypatch.setenv('LSQAMD_FUSED_JACOBIAN', '0')
    monkeypatch.setenv('LSQAMD_TRI_HALVES', '0')
    ref = amd.nonlinear_fit(**kw)
    monkeypatch.setenv('LSQAMD_TRI_HALVES', '1')
    fit = amd.nonlinear_fit(**kw)
    assert fit.nit == ref.nit
    assert np.allclose(fit.pmean, ref.pmean, rtol=1e-9, atol=1e-12)
    assert abs(fit.chi2 - ref.chi2) <= 1e-9 * ref.chi2
    assert np.allclose(fit.cov, ref.cov, rtol=1e-7, atol=1e-16)
