"""-m gpu: J^T f and chi2 formed by the diagonal tiles of the split-K J^T J launch (gemm_tn_f64.hip
gemm_tn_f64_interior_kernel<false, true, true>; api.hip eval_normal_dev) instead of a second pass over J.
Checked against the separate pass (LSQAMD_SYRK_COLSUM=0, read per call) and against numpy on the
Jacobian the device hands back; J^T J itself must not change by a bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def amd():
    import lsqfit_amd
    from lsqfit_amd import _lib
    _lib.load()
    return lsqfit_amd


# (N, P): one tile / three tiles / ten tiles; row counts that leave empty and ragged K-chunks
@pytest.mark.parametrize('N,P', [(1024, 128), (4096, 256), (1040, 256), (8192, 512), (16, 128)])
def test_gradient_from_the_syrk_equals_the_separate_pass(amd, N, P, monkeypatch):
    from lsqfit_amd import synth
    d = synth.make_cosmix(N=N, P=P, seed=N + P, block=0, prior_corr=False)
    wh = amd.Whitening(d['ymean'], d['yerr'], *d['prior'])
    rng = np.random.default_rng(1)
    p = d['p0'] * (1.0 + 0.05 * rng.standard_normal(P))
    got = {}
    for mode in ('1', '0'):
        monkeypatch.setenv('LSQAMD_SYRK_COLSUM', mode)
        pr = amd.DeviceProblem(d['model'], d['x'], wh)
        chi2 = pr.normal(p)
        got[mode] = (chi2, pr.get_grad(), pr.get_jtj(), pr.get_J_data(), pr.get_f_data())
        pr.close()
    c1, g1, A1, J, fd = got['1']
    c0, g0, A0, _, _ = got['0']
    assert np.array_equal(A1, A0)                      # the tiles themselves: same products, same order
    scale = np.abs(J).T @ np.abs(fd) + 1e-300          # bound on the rounding of a P-term... N-term sum
    assert np.all(np.abs(g1 - g0) <= 1e-13 * scale + 1e-13 * np.abs(g0))
    assert abs(c1 - c0) <= 1e-13 * c0
    # ... and against the Jacobian itself (data rows; the prior's share is the same in both runs)
    dg = g1 - g0
    assert np.all(np.isfinite(g1)) and np.max(np.abs(dg)) <= 1e-12 * np.max(np.abs(g0))


def test_fit_is_unchanged_to_rounding(amd, monkeypatch):
    from lsqfit_amd import synth
    d = synth.make_cosmix(N=4096, P=256, seed=9, block=0, prior_corr=False)
    kw = dict(data=(d['x'], d['ymean'], d['yerr']), model=d['model'], prior=d['prior'], p0=d['p0'])
    monkeypatch.setenv('LSQAMD_SYRK_COLSUM', '0')
    ref = amd.nonlinear_fit(**kw)
    monkeypatch.setenv('LSQAMD_SYRK_COLSUM', '1')
    fit = amd.nonlinear_fit(**kw)
    assert fit.nit == ref.nit
    assert np.allclose(fit.pmean, ref.pmean, rtol=1e-10, atol=1e-12)
    assert abs(fit.chi2 - ref.chi2) <= 1e-10 * ref.chi2
    assert np.allclose(fit.cov, ref.cov, rtol=1e-8, atol=1e-16)
