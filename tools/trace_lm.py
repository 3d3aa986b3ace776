"""Developer tool (GPU): step-by-step trace of the device LM driver on a synthetic problem."""
import ctypes as C
import sys
import numpy as np
import lsqfit_amd as amd
from lsqfit_amd import _lib, synth

N, P, block = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
d = synth.make_cosmix(N=N, P=P, seed=20262, block=block, prior_corr=True)
wh = amd.Whitening(d['ymean'], d['yerr'], *d['prior'])
pr = amd.DeviceProblem(d['model'], d['x'], wh)
pr.set_options((1e-8, 1e-10, 1e-10), 1000)
p = np.ascontiguousarray(d['p_true'] * (1 + 1e-3 * np.random.default_rng(3).standard_normal(P)))
assert pr.lib.lsqamd_init(pr.h, _lib.dptr(p)) == 0
xprev = pr.get_x()
for it in range(40):
    info = C.c_int32()
    rc = pr.lib.lsqamd_step(pr.h, C.byref(info))
    s = _lib.Summary()
    x = pr.get_x()
    g = pr.get_grad()
    dx = x - xprev
    xprev = x
    rel = np.max(np.abs(dx) / (1e-8 ** 2 + 1e-8 * np.abs(x)))
    gn = np.max(np.abs(np.maximum(x, 1.0) * g))
    chi2 = pr.chi2(x)
    print('it %2d rc %d info %d  chi2 %.10f  max|dx|/(xtol scale) %.3e  gnorm %.3e (gtol bound %.3e)'
          % (it, rc, info.value, chi2, rel, gn, 1e-10 * max(0.5 * chi2, 1.0)))
    if rc != 0 or info.value != 0:
        break
