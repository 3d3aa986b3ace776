"""Developer tool (GPU): the eight-wave whitening-synthesis kernel (both tile rows of a 256-row block in one workgroup, raw rows
synthesised once) against the two-workgroup kernel it replaces: the whitened Jacobian must be the same BITS, J^T f too."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lsqfit_amd as amd  # noqa: E402
from lsqfit_amd import synth  # noqa: E402

out = {}
for model in ('cosmix', 'multiexp'):
    for N, P in ((8192, 4096), (4096, 1024)):
        d = synth.make_cosmix(N=N, P=P, seed=5, block=256, prior_corr=False)
        mdl = d['model'] if model == 'cosmix' else amd.multiexp(P // 2)
        p = d['p0'] * (1.0 + 0.01 * np.cos(np.arange(P)))
        if model == 'multiexp':
            p[P // 2:] = 0.3 + 0.5 * np.arange(P // 2) / (P // 2)
        res = []
        for knob in ('0', '1'):     # 0 (default): the two-workgroup kernel
            os.environ['LSQAMD_SYNTH_PAIR'] = knob
            wh = amd.Whitening(d['ymean'], d['yerr'], *d['prior'])
            pr = amd.DeviceProblem(mdl, d['x'], wh)
            chi2 = pr.normal(p)
            fl = pr.lib.lsqamd_debug_flags(pr.h)
            res.append((chi2, pr.get_J_data().copy(), pr.get_grad().copy() if hasattr(pr, 'get_grad') else None, fl))
            pr.close()
        same = np.array_equal(res[0][1], res[1][1])
        print('%s N=%d P=%d: flags %d/%d, chi2 %r vs %r, J bit-identical %s, max |dJ| %.3e' % (
            model, N, P, res[0][3] & 2, res[1][3] & 2, res[0][0], res[1][0], same, np.abs(res[0][1] - res[1][1]).max()))
        assert same and res[0][0] == res[1][0]
print('ok')
