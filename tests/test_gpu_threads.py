"""-m gpu: the boundary's concurrency contract (SURVEY.md 8(b) "threading / re-entrancy").

The reference is NOT thread-safe -- module globals _valder, _p_f, _pyerr (src/lsqfit/_gsl.pyx:397-399), nested fits only by
save / restore (:667, :723), the GIL held for a whole fit.  The C ABI promises more: handle-scoped state, distinct handles
usable from several host threads (ctypes releases the GIL around every call), nested use.  Here:

  * 4 host threads x distinct handles, each thread a different KIND of work at the same time -- a one-launch NIST fit, a
    general-path correlated fit (P = 160: blocked Cholesky, captured step graphs), a lockstep batch (lsqamdb_*), and run-time
    compiles of formulas no other thread has seen (the hiprtc cache and its mutex) -- under LSQAMD_VERIFY_HANDOFF=1 (every
    polled pinned hand-off audited against the device's copy); every result BIT-identical to the serial run of the same job;
  * a nested fit: a fit started (and finished) inside the `fitargs` callback of an evidence sweep that is itself in the
    middle of its batches;
  * an exception raised inside the library while other threads run comes back as a code and its text in THAT handle's
    last_error only.
"""
import ctypes as C
import threading

import numpy as np
import pytest

from tests.helpers import load, nist_problem

pytestmark = pytest.mark.gpu
NIST = load('nist.json')


@pytest.fixture(scope='module')
def amd():
    import lsqfit_amd
    from lsqfit_amd import _lib
    _lib.load()
    return lsqfit_amd


def _nist_job(amd, name):
    pr = nist_problem(name, NIST)
    model = amd.expr(pr['expr'], ['b%d' % (i + 1) for i in range(pr['P'])], pr['columns'][1:])
    x = np.stack([pr['x'][c] for c in pr['columns'][1:]], axis=1)

    def run():
        fit = amd.nonlinear_fit(data=(x, pr['y'], pr['ysd']), model=model, prior=(pr['prior_mean'], pr['prior_sd']),
                                p0=pr['p0'], tol=pr['tol'])
        return dict(p=fit.pmean.copy(), cov=fit.cov.copy(), chi2=fit.chi2, nit=fit.nit, logGBF=fit.logGBF)
    return run


def _general_job(amd, seed):
    from lsqfit_amd import synth
    d = synth.make_cosmix(N=1024, P=160, seed=seed, block=128, prior_corr=True)

    def run():
        fit = amd.nonlinear_fit(data=(d['x'], d['ymean'], d['yerr']), model=d['model'], prior=d['prior'])
        return dict(p=fit.pmean.copy(), cov=fit.cov.copy(), chi2=fit.chi2, nit=fit.nit, logGBF=fit.logGBF)
    return run


def _batched_job(amd, seed):
    from lsqfit_amd import synth
    d = synth.make_cosmix(N=512, P=32, seed=seed, block=0, prior_corr=False)
    pm, ps = d['prior']
    B = 6
    psb = np.tile(ps, (B, 1))
    psb[:, :16] = (0.1 * 10 ** (2.0 * np.arange(B) / (B - 1)))[:, None]
    pmb = np.tile(pm, (B, 1))

    def run():
        bf = amd.BatchedFits(d['model'], d['x'], d['ymean'], d['yerr'], pmb, psb)
        out = bf.run(p0=np.tile(d['p0'], (B, 1)), use_graph=True)
        res = dict(p=out['pmean'].copy(), chi2=out['chi2'].copy(), nit=out['nit'].copy(), logGBF=out['logGBF'].copy(),
                   cov=np.stack([bf.cov(b) for b in range(B)]))
        bf.close()
        return res
    return run


def _jit_job(amd, tag, k):
    """A formula with constants of its own: no other job (and no earlier test) compiled it."""
    rng = np.random.default_rng(1000 * tag + k)
    x = np.linspace(0.1, 2.0, 40)
    c1, c2 = 1.0 + 0.001 * tag + 0.01 * k, 0.5 + 0.002 * tag + 0.003 * k
    y = 2.0 * np.exp(-c1 * 0.7 * x) + 0.3 * np.cos(c2 * x) + 0.01 * rng.standard_normal(x.size)
    text = 'a*exp(-%.6f*b*x) + c*cos(%.6f*x) + 0*d' % (c1, c2)

    def run():
        model = amd.expr(text, ['a', 'b', 'c', 'd'])
        fit = amd.nonlinear_fit(data=(x, y, np.full(x.size, 0.01)), model=model, prior=(np.array([1., 1., 0., 0.]), np.array([5., 5., 5., 1.])))
        return dict(p=fit.pmean.copy(), cov=fit.cov.copy(), chi2=fit.chi2, nit=fit.nit, logGBF=fit.logGBF)
    return run


def _same(a, b):
    assert a.keys() == b.keys()
    for k in a:
        assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k


def test_four_threads_distinct_handles_bit_identical_to_serial(amd, monkeypatch):
    monkeypatch.setenv('LSQAMD_VERIFY_HANDOFF', '1')
    from lsqfit_amd import _lib
    lib = _lib.load()
    before = (C.c_int64 * 3)()
    lib.lsqamd_handoff_stats(before)
    rounds = 3
    jobs = {
        0: [_nist_job(amd, n) for n in ('misra1a', 'thurber', 'mgh09')][:rounds],        # one-launch fits (P <= 32)
        1: [_general_job(amd, 700 + i) for i in range(rounds)],                          # blocked Cholesky, step graphs
        2: [_batched_job(amd, 800 + i) for i in range(rounds)],                          # lsqamdb_*: hipGraph rounds
        3: [_jit_job(amd, 3, i) for i in range(rounds)],                                 # hiprtc compiles nobody has cached
    }
    # the serial run FIRST would fill the compile cache for thread 3: its serial reference uses formulas of its own (tag 4)
    serial = {t: [job() for job in js] for t, js in jobs.items() if t != 3}
    results, errors = {}, []
    go = threading.Barrier(4)

    def worker(t):
        try:
            go.wait(timeout=120)
            results[t] = [job() for job in jobs[t]]
        except BaseException as e:       # noqa: BLE001 -- reported below, with the thread's number
            errors.append((t, repr(e)))
    threads = [threading.Thread(target=worker, args=(t,)) for t in jobs]
    [t.start() for t in threads]
    [t.join(timeout=900) for t in threads]
    assert not any(t.is_alive() for t in threads), 'a worker thread is still running (deadlock?)'
    assert not errors, errors
    for t in (0, 1, 2):
        for a, b in zip(serial[t], results[t]):
            _same(a, b)
    # thread 3's formulas again, now serially (cache hits): same bits as when they were compiled under contention
    for a, job in zip(results[3], jobs[3]):
        _same(a, job())
    after = (C.c_int64 * 3)()
    lib.lsqamd_handoff_stats(after)
    assert after[2] == before[2], 'LSQAMD_VERIFY_HANDOFF found a hand-off block that differed from the device copy'


def test_same_kind_of_work_on_every_thread(amd):
    """Four threads ALL running general-path fits of the same shape (same kernels, same attribute bookkeeping, same stream pool)
    and four all compiling the SAME new formula at once (one compile, three waiters on the cache)."""
    job = _general_job(amd, 910)
    ref = job()
    out = [None] * 4
    go = threading.Barrier(4)

    def worker(i):
        go.wait(timeout=120)
        out[i] = job()
    ts = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    [t.start() for t in ts]
    [t.join(timeout=600) for t in ts]
    for o in out:
        assert o is not None
        _same(ref, o)
    jit = _jit_job(amd, 9, 0)
    out2 = [None] * 4

    def worker2(i):
        go.wait(timeout=120)
        out2[i] = jit()
    ts = [threading.Thread(target=worker2, args=(i,)) for i in range(4)]
    [t.start() for t in ts]
    [t.join(timeout=600) for t in ts]
    for o in out2[1:]:
        assert o is not None
        _same(out2[0], o)


def test_nested_fits(amd):
    """A fit started from inside a sweep's callback while the sweep's own handle is alive and mid-search (the reference saves and
    restores its module globals for this, src/lsqfit/_gsl.pyx:667,:723; here state is handle-scoped), and one from inside a
    resampling loop.  The outer results equal those of the un-nested runs, the inner ones those of the same fit run alone."""
    from oracle import gvar_lite
    k = load('kat.json')['empbayes']
    x = np.array(k['src_inputs']['x'])
    ym, ys = gvar_lite.parse_array(k['src_inputs']['y'])
    model = amd.expr('exp(-b1 - b2*x - b3*x**2 - b4*x**3)', ['b1', 'b2', 'b3', 'b4'])
    inner_job = _general_job(amd, 920)
    inner_alone = inner_job()

    def fitargs(z):
        return dict(data=(x, ym, ys), model=model, prior=(np.zeros(4), np.full(4, abs(z))))
    fit0, z0 = amd.empbayes_fit(1.0, fitargs)
    inner = []

    def fitargs_nested(z):
        inner.append(inner_job())                      # a whole general-path fit inside the sweep's callback
        small = amd.nonlinear_fit(data=(x, ym, ys), model=model, prior=(np.zeros(4), np.full(4, 2.0)))   # and a one-launch one
        inner.append(dict(p=small.pmean.copy()))
        return fitargs(z)
    fit1, z1 = amd.empbayes_fit(1.0, fitargs_nested)
    assert z1 == z0 and np.array_equal(fit1.pmean, fit0.pmean) and fit1.logGBF == fit0.logGBF
    assert '%.5g' % fit1.logGBF == '21.274'            # examples/empbayes.out
    assert len(inner) >= 4
    for r in inner[0::2]:
        _same(inner_alone, r)
    for r in inner[1::2]:
        assert np.array_equal(r['p'], inner[1]['p'])
    # and around a resampling batch: copies, a fit of another problem, the same copies again -- same bits
    fit = amd.nonlinear_fit(data=(x, ym, ys), model=model, prior=(np.zeros(4), np.full(4, 5.3)))
    b1 = fit.bootstrapped_fits(3, seed=5)
    _same(inner_alone, inner_job())
    b2 = fit.bootstrapped_fits(3, seed=5)
    assert np.array_equal(b1.pmean, b2.pmean) and np.array_equal(b1.chi2, b2.chi2)


def test_exception_inside_the_library_stays_in_its_handle(amd):
    """lsqamd_debug_throw on one handle while another thread fits: code -3 / -10, the text in THAT handle's last_error, the
    other handle's fit untouched (bit-identical), the process alive."""
    from lsqfit_amd import synth
    job = _general_job(amd, 930)
    ref = job()
    d = synth.make_cosmix(N=256, P=16, seed=931, block=0, prior_corr=False)
    wh = amd.Whitening(d['ymean'], d['yerr'], *d['prior'])
    pr = amd.DeviceProblem(d['model'], d['x'], wh)
    got = {}

    def fitter():
        got['fit'] = job()
    t = threading.Thread(target=fitter)
    t.start()
    codes = [pr.lib.lsqamd_debug_throw(pr.h, kind) for kind in (1, 2, 3, 0)]
    texts = []
    for kind in (1, 2, 3):
        pr.lib.lsqamd_debug_throw(pr.h, kind)
        texts.append(pr.lib.lsqamd_last_error(pr.h).decode())
    t.join(timeout=600)
    assert codes == [-3, -10, -10, 0]
    assert 'bad_alloc' in texts[0] and 'lsqamd_debug_throw' in texts[1] and 'unknown C++ exception' in texts[2]
    _same(ref, got['fit'])
    assert pr.chi2(d['p0']) > 0           # the handle that caught the exceptions still works
    pr.close()


def test_recovery_from_a_capture_another_thread_invalidated(amd):
    """Other code in the process may use the legacy default stream (a plain hipMemcpy; torch's default stream).  On ROCm 7 such
    a call fails with hipErrorStreamCaptureImplicit while ANY thread has a graph capture open, invalidates that capture, and
    hipStreamEndCapture then leaves the stream in the invalidated state -- every later launch on it fails
    (tools/dbg_capture_reset.hip).  The library never uses the legacy stream itself (that was the defect this file found in
    the batch engine's setters), and a handle whose capture was invalidated by somebody else resets its stream
    (csrc/common.h capture_reset: an empty begin / end pair) and queues the step eagerly.  Deterministic reproduction through
    lsqamd_debug_capture_selftest: ONE intruding hipMemcpy from another thread (a thread hammering hipMemcpy beside capturing
    threads crashes inside the HIP runtime itself -- not something a library can promise to survive)."""
    import ctypes as C
    import torch
    from lsqfit_amd import _lib
    lib = _lib.load()
    st = torch.cuda.Stream()
    rep = (C.c_int32 * 6)()
    assert lib.lsqamd_debug_capture_selftest(C.c_void_p(st.cuda_stream), rep) == 0
    intruder, end_capture, status_after_end, status_after_reset, eager, value = list(rep)
    assert status_after_reset == 0 and eager == 0 and value == 42           # the stream is usable again
    if intruder != 0:            # the runtime refused the legacy call and invalidated the capture (ROCm 7.0 behaviour)
        assert end_capture != 0
    # and fits on handles created afterwards are what they were (same bits as before the episode)
    job = _general_job(amd, 940)
    a = job()
    assert lib.lsqamd_debug_capture_selftest(C.c_void_p(st.cuda_stream), rep) == 0
    _same(a, job())
