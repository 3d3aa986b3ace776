"""Developer tool (GPU): accuracy of the model kernel's sincos through the Jacobian of cosmix."""
import numpy as np
import lsqfit_amd as amd
from lsqfit_amd import synth
from tests import gpu_util as gu
d = synth.make_cosmix(N=4096, P=512, seed=3, block=0, prior_corr=False)
wh = amd.Whitening(d['ymean'], d['yerr'], *d['prior'])
pr = amd.DeviceProblem(d['model'], d['x'], wh)
p = d['p_true'].copy()
p[256:] *= 37.0        # frequencies up to ~9500 -> arguments up to 6e4
pr.normal(p)
J = pr.get_J_data()
Jref = gu.cosmix_jac(d['x'], p) * wh.wdiag[:, None]
print('max abs err of J / max|J|: %.3e' % (np.abs(J - Jref).max() / np.abs(Jref).max()))
f = pr.fcn(p)
print('max abs err of f: %.3e (|f| max %.2f)' % (np.abs(f - gu.cosmix_fcn(d['x'], p)).max(), np.abs(f).max()))
