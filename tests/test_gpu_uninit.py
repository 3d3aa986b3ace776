"""-m gpu: the engines must not depend on what the caller-provided workspace contained.  Each
problem is solved on fresh memory and again after the allocator's pool has been filled with NaNs;
results must agree bit for bit (this is also what exposes races: a racy in-place product only went
wrong once workgroups outnumbered the CUs)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def amd():
    import lsqfit_amd
    from lsqfit_amd import _lib
    _lib.load()
    return lsqfit_amd


def poison():
    import torch
    t = torch.full((2 * 1024 ** 3 // 8,), float('nan'), dtype=torch.float64, device='cuda')
    del t


@pytest.mark.parametrize('shape', [(2048, 256, 256, True), (1000, 130, 100, True), (4096, 512, 0, False),
                                   (1024, 128, 1024, True)])
def test_single_fit_ignores_workspace_contents(amd, shape):
    import torch
    from lsqfit_amd import synth
    N, P, block, pc = shape
    d = synth.make_cosmix(N=N, P=P, seed=5, block=block, prior_corr=pc)
    out = []
    for dirty in (False, True):
        if dirty:
            poison()
        else:
            torch.cuda.empty_cache()
        fit = amd.nonlinear_fit(data=(d['x'], d['ymean'], d['yerr']), model=d['model'], prior=d['prior'],
                                p0=d['p_true'])
        out.append((fit.pmean.copy(), fit.cov.copy(), fit.chi2, fit.nit, fit.logGBF,
                    fit.dp_dinputs(np.ones((1, P))), fit.dchi2(fit.pmean[None, :] * 1.0001)))
        fit.problem.close()
    for x, y in zip(*out):
        assert np.array_equal(np.asarray(x), np.asarray(y))
    assert np.all(np.isfinite(out[1][0])) and np.all(np.isfinite(out[1][1]))


def test_batched_fits_ignore_workspace_contents(amd):
    import torch
    from lsqfit_amd import synth
    d = synth.make_cosmix(N=2048, P=256, seed=6, block=256, prior_corr=False)
    B = 96
    pm = np.tile(d['prior'][0], (B, 1))
    ps = np.tile(d['prior'][1], (B, 1)) * np.linspace(0.5, 2.0, B)[:, None]
    out = []
    for dirty in (False, True):
        if dirty:
            poison()
        else:
            torch.cuda.empty_cache()
        bf = amd.BatchedFits(d['model'], d['x'], d['ymean'], d['yerr'], pm, ps)
        r = bf.run(p0=d['p0'])
        out.append((r['pmean'].copy(), r['chi2'].copy(), r['nit'].copy(), r['logGBF'].copy(), r['rounds']))
        bf.close()
    for x, y in zip(*out):
        assert np.array_equal(np.asarray(x), np.asarray(y))
    assert np.all(np.isfinite(out[1][3])) and np.all(out[1][2] > 0)
