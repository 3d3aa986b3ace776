import sys, time
import numpy as np
sys.path.insert(0, '.')
import lsqfit_amd as amd
from lsqfit_amd import _lib, fitter
from tests.helpers import load, nist_problem
NIST = load('nist.json')
pr = nist_problem('misra1a', NIST)
model = amd.expr(pr['expr'], ['b1', 'b2'], xnames=tuple(pr['columns'][1:]))
x = np.column_stack([pr['x'][c] for c in pr['columns'][1:]])
kw = dict(data=(x, pr['y'], pr['ysd']), model=model, prior=(pr['prior_mean'], pr['prior_sd']), p0=pr['p0'], tol=pr['tol'])
fit = amd.nonlinear_fit(**kw)
lib = _lib.load()
# wrap every lsqamd_* entry point with a timer
import collections
acc = collections.defaultdict(float); cnt = collections.Counter()
class Timed:
    def __init__(self, lib): self._lib = lib
    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if not name.startswith('lsqamd'): return fn
        def w(*a):
            t0 = time.perf_counter(); r = fn(*a); acc[name] += time.perf_counter() - t0; cnt[name] += 1; return r
        return w
_lib._LIB_TIMED = Timed(lib)
orig = _lib.load
_lib.load = lambda: _lib._LIB_TIMED
fitter._lib = _lib
t0 = time.perf_counter()
for rep in range(200):
    fit = amd.nonlinear_fit(**kw)
tot = (time.perf_counter() - t0) / 200
print('whole fit %.3f ms' % (1e3 * tot))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print('%-28s %6.1f us x %d' % (k, 1e6 * v / 200, cnt[k] // 200))
print('library total %.1f us' % (1e6 * sum(acc.values()) / 200))
