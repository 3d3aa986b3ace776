"""-m gpu: few parameters, many rows -- the normal equations formed straight from the compiled formula's registers
(jit.hip lsqamd_jit_nrm: J^T J, J^T f, chi2 per workgroup, the Jacobian never written; api.hip eval_normal_dev).
The fit must equal the oracle's and the unfused device path's; whatever reads the Jacobian afterwards (fit.J,
fit.residuals, the QR-grade covariance, fit.p sensitivities) must find it (ensure_J)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import fit as ofit
from tests import gpu_util as gu

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TEXT = 'a*exp(-b*x) + c/(1 + d*x**2) + e*x'
NAMES = ['a', 'b', 'c', 'd', 'e', 'unused']


def problem(N=20000, seed=5):
    rng = np.random.default_rng(seed)
    x = np.sort(rng.uniform(0.0, 4.0, N))
    pt = np.array([1.2, 0.8, 0.5, 1.5, 0.1, 0.0])
    f = pt[0] * np.exp(-pt[1] * x) + pt[2] / (1 + pt[3] * x ** 2) + pt[4] * x
    sd = 0.01 + 0.01 * rng.random(N)
    y = f + sd * rng.standard_normal(N)
    prior = (np.array([1.0, 1.0, 1.0, 1.0, 0.0, 0.3]), np.array([1.0, 1.0, 1.0, 1.0, 1.0, 0.2]))
    return x, y, sd, prior, pt


def fcn(x, p):
    from oracle import dual
    X = x['x'] if isinstance(x, dict) else x
    return p[0] * dual.exp(-p[1] * X) + p[2] / (1 + p[3] * X ** 2) + p[4] * X + 0.0 * p[5]


@pytest.fixture(autouse=True)
def general_path(monkeypatch):
    # (these tests are about the many-workgroup route of larger fits; a fit of up to 8192 rows would otherwise be ONE launch,
    #  tests/test_gpu_one_launch.py)
    monkeypatch.setenv('LSQAMD_ONE_LAUNCH_FIT', '0')


@pytest.fixture(scope='module')
def amd():
    import lsqfit_amd
    from lsqfit_amd import _lib
    _lib.load()
    return lsqfit_amd


@pytest.mark.parametrize('solver', ['cholesky', 'qr'])
def test_fused_normal_equations_fit(amd, solver):
    x, y, sd, prior, pt = problem()
    model = amd.expr(TEXT, NAMES)
    fit = amd.nonlinear_fit(data=(x, y, sd), model=model, prior=prior, solver=solver, tol=1e-10)
    assert fit.problem.lib.lsqamd_debug_flags(fit.problem.h) & 16           # the fused kernel ran
    ref = ofit.nonlinear_fit(x, y, sd, fcn, prior_mean=prior[0], prior_err=prior[1], solver=solver, tol=1e-10)
    assert np.max(np.abs(fit.pmean - ref.pmean) / ref.psdev) < 1e-6
    assert gu.relmax(fit.pmean[:5], ref.pmean[:5]) < 1e-6
    assert abs(fit.chi2 / ref.chi2 - 1) < 1e-8 and fit.nit == ref.nit
    assert gu.relmax(fit.cov, ref.cov) < 1e-6
    assert fit.logGBF == pytest.approx(ref.logGBF, rel=1e-9, abs=1e-6)
    assert fit.pmean[5] == pytest.approx(prior[0][5], abs=1e-12)          # the unread parameter stays on its prior
    # the rows nobody wrote during the fit are there when asked for
    J = fit.J[:x.size]
    w = 1.0 / sd
    a, b, c, d, e, _ = fit.pmean
    want = np.column_stack([np.exp(-b * x), -a * x * np.exp(-b * x), 1 / (1 + d * x ** 2), -c * x ** 2 / (1 + d * x ** 2) ** 2, x,
                            np.zeros_like(x)]) * w[:, None]
    assert gu.relmax(J, want) < 1e-12
    r = fit.residuals[:x.size]
    assert gu.relmax(r, w * (a * np.exp(-b * x) + c / (1 + d * x ** 2) + e * x - y)) < 1e-11
    D = fit.dp_dinputs()
    assert D.shape == (6, x.size + 6) and gu.relmax(D[:, x.size:] @ np.diag(prior[1] ** 2) @ D[:, x.size:].T
                                                    + (D[:, :x.size] * sd ** 2) @ D[:, :x.size].T, fit.cov) < 1e-8


def test_fused_and_unfused_paths_agree(amd, tmp_path):
    x, y, sd, prior, pt = problem(N=5000, seed=6)
    fit = amd.nonlinear_fit(data=(x, y, sd), model=amd.expr(TEXT, NAMES), prior=prior, tol=1e-10)
    prog = ('import sys, numpy as np\n'
            'sys.path.insert(0, %r)\n'
            'import lsqfit_amd as amd\n'
            'from tests.test_gpu_fused_normal import problem, TEXT, NAMES\n'
            'x, y, sd, prior, pt = problem(N=5000, seed=6)\n'
            'fit = amd.nonlinear_fit(data=(x, y, sd), model=amd.expr(TEXT, NAMES), prior=prior, tol=1e-10)\n'
            'assert not (fit.problem.lib.lsqamd_debug_flags(fit.problem.h) & 16)\n'
            'np.savez(%r, pmean=fit.pmean, cov=fit.cov, chi2=fit.chi2, nit=fit.nit)\n' % (ROOT, str(tmp_path / 'o.npz')))
    r = subprocess.run([sys.executable, '-c', prog], env=dict(os.environ, LSQAMD_FUSED_NORMAL='0', LSQAMD_ONE_LAUNCH_FIT='0'), cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    o = np.load(str(tmp_path / 'o.npz'))
    assert gu.relmax(o['pmean'][:5], fit.pmean[:5]) < 1e-10 and gu.relmax(o['cov'], fit.cov) < 1e-9
    assert float(o['chi2']) == pytest.approx(fit.chi2, rel=1e-11) and int(o['nit']) == fit.nit
