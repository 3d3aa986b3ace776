"""Developer tool (GPU): bootstrap / simulated copies of small fits as one batch -- one workgroup per copy in ONE launch
(jit.hip lsqamd_jit_lmb) against the lockstep engine (LSQAMD_ONE_LAUNCH_FIT=0)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import lsqfit_amd as amd
from tests.helpers import load, nist_problem

NIST = load('nist.json')
for name, n in (('misra1a', 200), ('misra1a', 2000), ('thurber', 200), ('gauss1', 1000)):
    pr = nist_problem(name, NIST)
    model = amd.expr(pr['expr'], ['b%d' % (i + 1) for i in range(pr['P'])], xnames=tuple(pr['columns'][1:]))
    x = np.column_stack([pr['x'][c] for c in pr['columns'][1:]])
    fit = amd.nonlinear_fit(data=(x, pr['y'], pr['ysd']), model=model, prior=(pr['prior_mean'], pr['prior_sd']), p0=pr['p0'], tol=1e-8)
    line = '%-8s %5d bootstrap copies:' % (name, n)
    keep = {}
    for mode in ('1', '0'):
        os.environ['LSQAMD_ONE_LAUNCH_FIT'] = mode
        res = fit.bootstrapped_fits(n, seed=1)
        t0 = time.perf_counter()
        res = fit.bootstrapped_fits(n, seed=1)
        dt = time.perf_counter() - t0
        keep[mode] = res
        line += '   %s %.2f ms (device %.2f ms, %d round(s))' % ('one launch' if mode == '1' else 'lockstep', 1e3 * dt, res['device_ms'], res['rounds'])
    a, b = keep['1'], keep['0']
    line += '   max |dp|/sd %.1e' % np.max(np.abs(a['pmean'] - b['pmean']) / b['psdev'])
    print(line)
