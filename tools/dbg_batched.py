import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lsqfit_amd as amd
from tests.test_gpu_batched import make
d, pmb, psb = make(256, 16, 4, 52)
bf = amd.BatchedFits(d['model'], d['x'], d['ymean'], d['yerr'], pmb, psb)
p0 = np.tile(d['p0'], (4, 1)); p0[1, 8:] += 1.0
for g in (False, True):
    out = bf.run(p0=p0, maxit=40, tol=1e-8, use_graph=g)
    print('graph', g, 'nit', out['nit'], 'nfev', out['nfev'], 'crit', out['stopping_criterion'], 'rounds', out['rounds'], 'chi2', out['chi2'])
for b in range(4):
    s = amd.nonlinear_fit(data=(d['x'], d['ymean'], d['yerr']), model=d['model'], prior=(pmb[b], psb[b]), p0=p0[b], maxit=40, tol=1e-8)
    sm = s.fitter_results.summary
    print(b, 'single nit', sm.nit, 'nfev', sm.nfev, 'ntrial', sm.ntrial, 'cholfail', sm.chol_fail, 'crit', sm.stopping_criterion, 'chi2', sm.chi2, 'maxdiff', np.abs(out['pmean'][b]-s.pmean).max())
