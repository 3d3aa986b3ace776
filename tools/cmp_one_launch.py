"""Developer tool (GPU): the 27 NIST fits through the one-launch kernel and through the general path, side by side."""
import os
import sys

import numpy as np

sys.path.insert(0, '.')
import lsqfit_amd as amd
from tests.helpers import load, nist_problem

NIST = load('nist.json')
for name in sorted(NIST):
    pr = nist_problem(name, NIST)
    model = amd.expr(pr['expr'], ['b%d' % (i + 1) for i in range(pr['P'])], pr['columns'][1:])
    x = np.stack([pr['x'][c] for c in pr['columns'][1:]], axis=1)
    kw = dict(data=(x, pr['y'], pr['ysd']), model=model, prior=(pr['prior_mean'], pr['prior_sd']), p0=pr['p0'], tol=pr['tol'])
    out = []
    for mode in ('1', '0'):
        os.environ['LSQAMD_ONE_LAUNCH_FIT'] = mode
        fit = amd.nonlinear_fit(**kw)
        s = fit.fitter_results.summary
        out.append((fit, s, fit.problem.lib.lsqamd_debug_flags(fit.problem.h) & 32))
    (a, sa, fa), (b, sb, fb) = out
    print('%-10s one-launch %d  nit %3d/%3d nfev %3d/%3d njev %3d/%3d crit %d/%d  dp/sd %.1e  dchi2 %.1e  run ms %.3f/%.3f'
          % (name, fa != 0, a.nit, b.nit, sa.nfev, sb.nfev, sa.njev, sb.njev, sa.stopping_criterion, sb.stopping_criterion,
             np.max(np.abs(a.pmean - b.pmean) / b.psdev), abs(a.chi2 - b.chi2) / b.chi2, sa.t_run_ms, sb.t_run_ms))
