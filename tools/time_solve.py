"""Developer tool (GPU): time the damped solve (factorisation + substitutions) of a P = 4096 problem
through the C ABI (lsqamd_solve_damped), reporting the handle's phase timers."""
import sys
import numpy as np
sys.path.insert(0, '.')
import lsqfit_amd as amd
from lsqfit_amd import synth, _lib
_lib.load()
d = synth.make_cosmix(N=8192, P=4096, seed=20263, block=256, prior_corr=True)
wh = amd.Whitening(d['ymean'], d['yerr'], *d['prior'])
pr = amd.DeviceProblem(d['model'], d['x'], wh)
pr.normal(d['p0'])
diag = np.sqrt(np.diag(pr.get_jtj()))
pr.timing(True)
for rep in range(3):
    pr.timing_reset()
    for _ in range(10):
        pr.solve_damped(1e-3, diag)
    t = pr.timings()
    print('cholesky %.3f ms  solve %.3f ms' % (t['cholesky'][0] / t['cholesky'][1], t['solve'][0] / t['solve'][1]))
# the solution against LAPACK on the host (the chained back substitution hands 1 KiB pieces between workgroups)
A = pr.get_jtj()
g = pr.get_grad()
M = A + 1e-3 * np.diag(diag ** 2)
v = pr.solve_damped(1e-3, diag)
ref = np.linalg.solve(M, g)
print('solution vs LAPACK: rel %.2e' % (np.max(np.abs(np.abs(v) - np.abs(ref))) / np.max(np.abs(ref))))
