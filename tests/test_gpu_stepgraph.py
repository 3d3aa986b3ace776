"""-m gpu: the device-resident LM step replayed from captured graphs (api.hip run_half): the trial
(factor, solve, trial point, residual, decision, 128-byte read-back) and the accepted branch
(Jacobian, normal equations, scaling, convergence test) are each captured once per parameter-buffer
parity and replayed.  Same kernels, same arguments: the fit must be bit-identical to the eager one."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SCRIPT = r'''
import hashlib, sys
import numpy as np
sys.path.insert(0, %(root)r)
import lsqfit_amd as amd
from lsqfit_amd import synth
out = []
for N, P, block, corr in ((2000, 64, 0, False), (1536, 128, 256, True), (4096, 256, 0, False)):
    d = synth.make_cosmix(N=N, P=P, seed=5, block=block, prior_corr=corr)
    p0 = d['p0'] * (1 + 0.002 * np.random.default_rng(1).standard_normal(P))
    pr = amd.DeviceProblem(d['model'], d['x'], amd.Whitening(d['ymean'], d['yerr'], *d['prior']))
    fits = [amd.nonlinear_fit(data=(d['x'], d['ymean'], d['yerr']), model=d['model'], prior=d['prior'], p0=p0,
                              problem=pr, tol=1e-10) for _ in range(2)]        # the second run replays from the start
    assert all(f.error is None and f.nit >= 4 for f in fits), [(f.error, f.nit) for f in fits]
    assert np.array_equal(fits[0].pmean, fits[1].pmean) and fits[0].nit == fits[1].nit
    h = hashlib.sha256()
    for a in (fits[0].pmean, fits[0].cov, np.array([fits[0].chi2, fits[0].logGBF, fits[0].nit])):
        h.update(np.ascontiguousarray(a).tobytes())
    out.append('%%s:%%d:%%d' %% (h.hexdigest(), fits[0].nit, (pr.lib.lsqamd_debug_flags(pr.h) >> 2) & 1))
    pr.close()
print(' '.join(out))
'''


def test_captured_steps_are_bit_identical_to_eager_steps(tmp_path):
    script = tmp_path / 'stepgraph.py'
    script.write_text(_SCRIPT % dict(root=ROOT))
    got = {}
    for knob in ('1', '0'):
        env = dict(os.environ, LSQAMD_STEP_GRAPH=knob)
        r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        got[knob] = [t.split(':') for t in r.stdout.strip().splitlines()[-1].split()]
    assert len(got['1']) == 3
    for g, e in zip(got['1'], got['0']):
        assert g[0] == e[0] and g[1] == e[1]              # same bits, same iteration count
        assert g[2] == '1' and e[2] == '0'                # ... and the graphs really ran / really did not
