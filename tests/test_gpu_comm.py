"""-m gpu: the collective inside the library (lsqamd_comm_*: persistent RCCL communicator, sums
enqueued on the handle's stream).  A 1-GPU box admits one rank per device, so what runs here is the
one-rank communicator -- the same call sequence (id, init, reduce-scatter + all-gather + tail
all-reduce, or the single all-reduce) with nothing to add: the fit must equal the communicator-free
fit bit for bit.  The N > 1 arithmetic is covered by tests/test_dist_gloo.py / test_gpu_dist2.py."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def amd():
    import lsqfit_amd
    from lsqfit_amd import _lib
    _lib.load()
    return lsqfit_amd


@pytest.mark.parametrize('algo', ['rsag', 'allreduce'])
def test_one_rank_communicator_is_transparent(amd, algo, monkeypatch):
    from lsqfit_amd import synth
    monkeypatch.setenv('LSQAMD_COMM_ALGO', algo)
    # (single-rank fits of this size take fused single-workgroup tails that a handle with a communicator does not:
    # same arithmetic up to the order of a few sums.  Bit-for-bit transparency is a statement about the general
    # kernels, so both fits run on those.)
    monkeypatch.setenv('LSQAMD_SMALL_FUSE', '0')
    d = synth.make_cosmix(N=1024, P=256, seed=77, block=64, prior_corr=True)
    kw = dict(data=(d['x'], d['ymean'], d['yerr']), model=d['model'], prior=d['prior'])
    ref = amd.nonlinear_fit(**kw)
    wh = amd.Whitening(d['ymean'], d['yerr'], *d['prior'])
    pr = amd.DeviceProblem(d['model'], d['x'], wh)
    assert pr.comm_info() == (-1, 0)
    uid = pr.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    pr.comm_init(uid, 0, 1)
    assert pr.comm_info() == (0, 1)
    pr.timing(True)
    fit = amd.nonlinear_fit(problem=pr, **kw)
    tm = pr.timings()
    assert np.array_equal(fit.pmean, ref.pmean) and np.array_equal(fit.cov, ref.cov)
    assert fit.chi2 == ref.chi2 and fit.nit == ref.nit and fit.logGBF == ref.logGBF
    s = fit.fitter_results.summary
    assert tm['reduce'][1] == s.njev + s.nfev - 1        # one packed exchange per Jacobian, one scalar per trial
    # the many-point chi2 path reduces through the communicator as well
    pts = ref.pmean + 1e-3 * np.random.default_rng(0).standard_normal((5, 256))
    assert np.allclose(fit.dchi2(pts), ref.dchi2(pts), rtol=1e-12, atol=1e-9)
    pr.comm_destroy()
    assert pr.comm_info() == (-1, 0)
    again = amd.nonlinear_fit(problem=pr, **kw)
    assert np.array_equal(again.pmean, ref.pmean)
    pr.close()


def test_comm_argument_checks(amd):
    from lsqfit_amd import synth
    d = synth.make_cosmix(N=64, P=8, seed=3, block=0)
    wh = amd.Whitening(d['ymean'], d['yerr'], *d['prior'])
    pr = amd.DeviceProblem(d['model'], d['x'], wh)
    with pytest.raises(RuntimeError, match='EINVAL'):
        pr.comm_init(b'\0' * 16, 0, 1)
    with pytest.raises(RuntimeError, match='EINVAL'):
        pr.comm_init(pr.comm_unique_id(), 2, 2)
    pr.close()


def test_two_handles_share_one_communicator(amd):
    """communicators belong to the process: the second handle that names an id takes a reference (no second
    ncclCommInitRank), both fit through it one after the other, and it outlives the handle that created it"""
    from lsqfit_amd import synth
    d = synth.make_cosmix(N=512, P=128, seed=78, block=64, prior_corr=True)
    kw = dict(data=(d['x'], d['ymean'], d['yerr']), model=d['model'], prior=d['prior'])
    ref = amd.nonlinear_fit(**kw)
    wh = amd.Whitening(d['ymean'], d['yerr'], *d['prior'])
    a = amd.DeviceProblem(d['model'], d['x'], wh)
    b = amd.DeviceProblem(d['model'], d['x'], wh)
    assert a.comm_stats() == (0.0, 0)
    uid = a.comm_unique_id()
    a.comm_init(uid, 0, 1)
    ms, n = a.comm_stats()
    assert ms > 0.0 and n == 1
    b.comm_init(uid, 0, 1)
    assert a.comm_stats() == (ms, 2) and b.comm_stats() == (ms, 2)          # same communicator, same set-up time
    with pytest.raises(RuntimeError):
        b.comm_init(uid, 0, 2)                                                 # the id names a one-rank communicator
    b.comm_init(uid, 0, 1)
    fa = amd.nonlinear_fit(problem=a, **kw)
    fb = amd.nonlinear_fit(problem=b, **kw)
    for f in (fa, fb):
        assert np.allclose(f.pmean, ref.pmean, rtol=1e-12, atol=0) and f.nit == ref.nit
    a.close()
    assert b.comm_stats() == (ms, 1)
    fb2 = amd.nonlinear_fit(problem=b, **kw)
    assert np.array_equal(fb2.pmean, fb.pmean)
    b.close()
    from lsqfit_amd import _lib
    assert _lib.load().lsqamd_comm_shutdown() == 0
