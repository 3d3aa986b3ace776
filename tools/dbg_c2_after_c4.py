"""Developer tool (GPU): why does the c2 companion of the default bench run slower than `bench.py --workload c2`?"""
import ctypes as C
import sys
import time
import numpy as np
sys.path.insert(0, '.')
import torch
import lsqfit_amd
from lsqfit_amd import _lib, synth
from lsqfit_amd.dist import sharded_problem


def make(N, P, block, dense, seed):
    d = synth.make_cosmix(N=N, P=P, seed=seed, block=block, prior_corr=dense)
    wh = lsqfit_amd.Whitening(d['ymean'], d['yerr'], *d['prior'])
    pr = sharded_problem(d['model'], d['x'], wh, 0, 1)
    pr.set_options((1e-8, 1e-10, 1e-10), 1000)
    return d, pr


def steps(d, pr, n, seed=1, trace=None):
    lib, h = pr.lib, pr.h
    rng = np.random.Generator(np.random.PCG64(seed))
    P = d['p0'].size
    ps = np.concatenate([np.full(P // 2, 0.5), np.full(P // 2, 0.1)])
    conv = True
    t0 = None
    for i in range(n + 20):
        if i == 20:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        if conv:
            ts = time.perf_counter()
            lib.lsqamd_init(h, _lib.dptr(np.ascontiguousarray(d['p0'] + 0.3 * ps * rng.standard_normal(P))))
            if trace is not None:
                trace.append((-i, 1e3 * (time.perf_counter() - ts), 'init'))
            conv = False
        info = C.c_int32(0)
        ts = time.perf_counter()
        rc = lib.lsqamd_step(h, C.byref(info))
        if trace is not None:
            trace.append((i, 1e3 * (time.perf_counter() - ts), 'restart before' if (i > 0 and trace and False) else ''))
        if rc != 0 or info.value != 0:
            conv = True
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


d2, p2 = make(4096, 256, 0, False, 20261)
tr0 = []
print('c2 alone                         %.4f ms/step' % steps(d2, p2, 200, trace=tr0))
print('   lsqamd_init calls of that run (before step, ms):', [(-i, round(ms, 2)) for i, ms, k in tr0 if k == 'init'])
print('c2 again                         %.4f' % steps(d2, p2, 200))
d4, p4 = make(65536, 4096, 256, True, 20263)
print('c2 with an idle c4 handle alive  %.4f' % steps(d2, p2, 200))
print('c4 steps                         %.3f' % steps(d4, p4, 5))
print('c2 after c4 ran (c4 alive)       %.4f' % steps(d2, p2, 200))
p4.timing(True)
steps(d4, p4, 3)
p4.timing(False)
print('c2 after c4 ran with timers      %.4f' % steps(d2, p2, 200))
d2b, p2b = make(4096, 256, 0, False, 20261)
tr = []
print('a NEW c2 handle (c4 alive)       %.4f' % steps(d2b, p2b, 200, trace=tr))
big = sorted(tr, key=lambda t: -t[1])[:12]
print('   lsqamd_init calls of that run (before step, ms):', [(-i, round(ms, 2)) for i, ms, k in tr if k == 'init'])
print('   slowest lsqamd_step calls of that run (step index, ms):', [(i, round(ms, 2)) for i, ms, _ in big], ' median %.3f' % np.median([t[1] for t in tr]))
p4.close()
torch.cuda.empty_cache()
print('c2 after c4 closed               %.4f' % steps(d2, p2, 200))
print('new c2 handle after c4 closed    %.4f' % steps(d2b, p2b, 200))
d2c, p2c = make(4096, 256, 0, False, 20261)
print('a third c2 handle                %.4f' % steps(d2c, p2c, 200))
print('debug flags', [hex(p.lib.lsqamd_debug_flags(p.h)) for p in (p2, p2b, p2c)])


def stats():
    a = (C.c_int64 * 3)()
    _lib.load().lsqamd_handoff_stats(a)
    return list(a)


print('--- a fourth handle, 50 steps at a time (hand-off counters: polled again / served from device memory / mismatches)')
d2d, p2d = make(4096, 256, 0, False, 20261)
for k in range(8):
    s0 = stats()
    t = steps(d2d, p2d, 50, seed=k)
    s1 = stats()
    print('   steps %3d-%3d: %.4f ms/step, hand-off deltas %s, graph launches flag %s' % (70 * k, 70 * k + 69, t, [b - a for a, b in zip(s0, s1)],
                                                                                      hex(p2d.lib.lsqamd_debug_flags(p2d.h))))
import os
os.environ['LSQAMD_STEP_GRAPH'] = '0'
