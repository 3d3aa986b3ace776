"""Developer tool: hammer the one-launch fit's host/device hand-off.

    python tools/stress_one_launch.py [loops]

Repeats the sequence of tests/test_gpu_one_launch.py that went red on the round-3 driver box (nine prior x scaler fits
through both routes, then the maxit = 3 fit) and counts every disagreement instead of stopping at the first.  Run it
with LSQAMD_VERIFY_HANDOFF=1 (api.hip run_one_launch prints every word that changed between "flag seen" and "stream
drained") and with LSQAMD_POISON_PINNED=1."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lsqfit_amd as amd  # noqa: E402
from lsqfit_amd import _lib  # noqa: E402

_lib.load()
ONE = 32


def curve(N=600, seed=3):
    rng = np.random.default_rng(seed)
    x = np.sort(rng.uniform(0.0, 5.0, N))
    pt = np.array([1.5, 0.7, 0.4, 2.0])
    f = pt[0] * np.exp(-pt[1] * x) + pt[2] * np.cos(pt[3] * x)
    sd = 0.02 + 0.01 * rng.random(N)
    return x, f + sd * rng.standard_normal(N), sd, pt


def both(**kw):
    os.environ['LSQAMD_ONE_LAUNCH_FIT'] = '1'
    one = amd.nonlinear_fit(**kw)
    f1 = one.problem.lib.lsqamd_debug_flags(one.problem.h)
    os.environ['LSQAMD_ONE_LAUNCH_FIT'] = '0'
    gen = amd.nonlinear_fit(**kw)
    return one, f1, gen


def differ(one, gen):
    s1, s0 = one.fitter_results.summary, gen.fitter_results.summary
    out = []
    if s1.stopping_criterion != s0.stopping_criterion:
        out.append('stop %d vs %d' % (s1.stopping_criterion, s0.stopping_criterion))
    if abs(one.nit - gen.nit) > max(2, gen.nit // 8):
        out.append('nit %d vs %d' % (one.nit, gen.nit))
    if not np.all(np.abs(one.pmean - gen.pmean) <= 1e-8 * np.abs(gen.pmean) + 5e-6 * gen.psdev):
        out.append('p')
    if not abs(one.chi2 - gen.chi2) <= 1e-8 * max(gen.chi2, 1e-12) + 1e-20:
        out.append('chi2 %r vs %r' % (one.chi2, gen.chi2))
    if not np.allclose(one.cov, gen.cov, rtol=1e-5, atol=1e-9 * np.max(np.abs(gen.cov))):
        out.append('cov')
    return out


def main():
    loops = int(sys.argv[1]) if len(sys.argv) > 1 else 500
    model = amd.expr('a*exp(-b*x) + c*cos(d*x)', ['a', 'b', 'c', 'd'])
    x, y, sd, pt = curve()
    x3, y3, sd3, pt3 = curve(N=3000, seed=8)
    L = np.tril(0.2 * np.random.default_rng(1).standard_normal((4, 4))) + 0.6 * np.eye(4)
    bad = 0
    fits = 0
    t0 = time.time()
    # a wide fit (30 parameters: a record block of ~1100 words, most of it covariance) and a correlated one, every tenth loop
    rngw = np.random.default_rng(4)
    xw = np.linspace(0.0, 10.0, 128)
    cs = np.linspace(0.7, 9.3, 15)
    aw, bw = 1.0 + 0.5 * rngw.random(15), 2.0 + rngw.random(15)
    ptw = np.concatenate([aw, bw])
    yw = sum(aw[k] * np.exp(-bw[k] * (xw - cs[k]) ** 2) for k in range(15)) + 0.02 * rngw.standard_normal(128)
    namesw = ['a%d' % k for k in range(15)] + ['b%d' % k for k in range(15)]
    wide = amd.expr(' + '.join('a%d*exp(-b%d*(x - %r)**2)' % (k, k, float(cs[k])) for k in range(15)), namesw)
    for it in range(loops):
        if it % 10 == 0:
            kw = dict(data=(xw, yw * (1.0 + 1e-3 * (it % 7)), np.full(128, 0.02)), model=wide, prior=(ptw, np.full(30, 0.5)), p0=ptw * 1.05)
            one, f1, gen = both(**kw)
            fits += 1
            d = differ(one, gen)
            if d or not f1 & ONE:
                bad += 1
                print('loop %d wide: route %d %s' % (it, f1 & ONE, d), flush=True)
        for prior in ('diag', 'dense', 'none'):
            for scaler in ('more', 'levenberg', 'marquardt'):
                kw = dict(data=(x, y, sd), model=model, p0=pt * 1.2, scaler=scaler)
                if prior == 'diag':
                    kw['prior'] = (pt * 1.1, np.full(4, 0.5))
                elif prior == 'dense':
                    kw['prior'] = (pt * 1.1, L @ L.T)
                one, f1, gen = both(**kw)
                fits += 1
                d = differ(one, gen)
                if d or not f1 & ONE:
                    bad += 1
                    print('loop %d %s/%s: route %d %s' % (it, prior, scaler, f1 & ONE, d), flush=True)
        kw = dict(data=(x3, y3, sd3), model=model, prior=(pt3, np.full(4, 1.0)), p0=pt3 * 1.4, maxit=3)
        one, f1, gen = both(**kw)
        fits += 1
        d = differ(one, gen)
        if d or one.nit != 3 or one.stopping_criterion != 0 or not f1 & ONE:
            bad += 1
            print('loop %d maxit3: route %d nit %d stop %d %s' % (it, f1 & ONE, one.nit, one.stopping_criterion, d), flush=True)
        os.environ['LSQAMD_ONE_LAUNCH_FIT'] = '1'
        kw['maxit'] = 200
        a = amd.nonlinear_fit(problem=one.problem, **kw)
        b = amd.nonlinear_fit(problem=one.problem, **kw)
        fits += 2
        if not (a.nit == b.nit and np.array_equal(a.pmean, b.pmean) and np.array_equal(a.cov, b.cov)):
            bad += 1
            print('loop %d resident: nit %d vs %d' % (it, a.nit, b.nit), flush=True)
    import ctypes
    st = (ctypes.c_int64 * 3)()
    _lib.load().lsqamd_handoff_stats(st)
    print('stress_one_launch: %d loops, %d one-launch fits, %d disagreements, %.1f s (poison %s, verify %s, zero-copy %s); '
          'hand-offs: %d snapshots polled again, %d served from device memory, %d words differed from the device copy'
          % (loops, fits, bad, time.time() - t0, os.environ.get('LSQAMD_POISON_PINNED', '0'),
             os.environ.get('LSQAMD_VERIFY_HANDOFF', '0'), os.environ.get('LSQAMD_ZERO_COPY', '1'), st[0], st[1], st[2]), flush=True)
    return 1 if bad or st[2] else 0


if __name__ == '__main__':
    sys.exit(main())
