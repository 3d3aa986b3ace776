"""-m gpu: Jacobian rows synthesised inside the whitening product (gemm_tn_f64.hip whiten_synth_kernel)
against the two-kernel route (Jacobian kernel -> raw buffer -> whitening GEMM): same tile, same staging
order, same MFMA sequence, so the whitened Jacobian and J^T J must agree BIT FOR BIT (the residual
column comes from the residual kernel instead of the Jacobian kernel: chi2 and J^T f agree to
rounding); whole fits, both sum models, a sharded handle."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import sys, numpy as np
sys.path.insert(0, %r)
import lsqfit_amd as amd
from lsqfit_amd import synth
out = {}
for name, kw in dict(cos=dict(N=2048, P=256, block=256, prior_corr=True), cos512=dict(N=1024, P=128, block=512, prior_corr=False),
                     shard=dict(N=2048, P=128, block=128, prior_corr=True)).items():
    d = synth.make_cosmix(seed=7, **kw)
    wh = amd.Whitening(d['ymean'], d['yerr'], *d['prior'])
    rows = (512, 1536) if name == 'shard' else None
    pr = amd.DeviceProblem(d['model'], d['x'], wh, rows=rows)
    p = d['p_true'] * (1 + 1e-3 * np.random.default_rng(1).standard_normal(kw['P']))
    out[name + '_chi2'] = pr.normal(p)
    out[name + '_A'] = pr.get_jtj(); out[name + '_g'] = pr.get_grad(); out[name + '_J'] = pr.get_J_data()
    if rows is None:
        fit = amd.nonlinear_fit(data=(d['x'], d['ymean'], d['yerr']), model=d['model'], prior=d['prior'], problem=pr)
        out[name + '_p'] = fit.pmean; out[name + '_cov'] = fit.cov; out[name + '_nit'] = fit.nit
    pr.close()
# multiexp through the same kernel
x = np.linspace(0.05, 3.0, 512); K = 64
model = amd.multiexp(K)
rng = np.random.default_rng(3)
pt = np.concatenate([rng.uniform(0.5, 1.5, K), 0.05 * np.arange(1, K + 1)])
y = np.exp(-np.outer(x, pt[K:])) @ pt[:K]
sd = 0.01 * np.abs(y)
blocks = [(r0, np.diag(sd[r0:r0 + 128] ** 2) + 0.3 * np.outer(sd[r0:r0 + 128], sd[r0:r0 + 128])) for r0 in range(0, 512, 128)]
wh = amd.Whitening(y, dict(sdev=sd, blocks=blocks), pt, np.full(2 * K, 0.5))
pr = amd.DeviceProblem(model, x, wh)
out['mexp_chi2'] = pr.normal(pt * 1.01); out['mexp_A'] = pr.get_jtj(); out['mexp_J'] = pr.get_J_data()
pr.close()
np.savez(sys.argv[1], **out)
'''


def run(tmp, fused):
    env = dict(os.environ, LSQAMD_FUSED_JACOBIAN='1' if fused else '0')
    path = os.path.join(tmp, 'fused%d.npz' % fused)
    r = subprocess.run([sys.executable, '-c', SCRIPT % ROOT, path], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return np.load(path)


def test_fused_route_is_bit_identical(tmp_path):
    a, b = run(str(tmp_path), True), run(str(tmp_path), False)
    assert set(a.files) == set(b.files) and len(a.files) >= 14
    for k in a.files:
        if k.endswith('_J') or k.endswith('_A'):
            assert np.array_equal(a[k], b[k]), k
        elif k.endswith('_nit'):
            assert abs(int(a[k]) - int(b[k])) <= 1, k
        else:
            scale = np.max(np.abs(b[k]))
            assert np.max(np.abs(a[k] - b[k])) <= 1e-9 * scale, k


def test_fused_route_is_taken():
    """The bench shape's structure (uniform triangular blocks covering every row, sum model) takes the
    fused kernel: the handle never launches the Jacobian kernel (phase timer) after the first evaluation."""
    import lsqfit_amd as amd
    from lsqfit_amd import synth
    d = synth.make_cosmix(N=1024, P=128, seed=9, block=256, prior_corr=True)
    wh = amd.Whitening(d['ymean'], d['yerr'], *d['prior'])
    pr = amd.DeviceProblem(d['model'], d['x'], wh)
    pr.timing(True)
    fit = amd.nonlinear_fit(data=(d['x'], d['ymean'], d['yerr']), model=d['model'], prior=d['prior'], problem=pr)
    tm = pr.timings()
    assert fit.error is None and tm['whiten'][1] == fit.fitter_results.summary.njev
    # ('jacobian' now times the residual column only: evaluated afresh just once, at the start)
    assert pr.lib.lsqamd_debug_flags(pr.h) & 2
    pr.close()
