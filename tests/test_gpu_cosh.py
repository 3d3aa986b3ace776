"""-m gpu: the functions a gvar-overloaded fit function may call beyond exp/log/sin/cos/arctan/sqrt -- tan sinh cosh tanh
arcsin arccos abs (LSQAMD_OP_TAN .. LSQAMD_OP_ABS) -- in the compiled route (jit.hip) and in the three interpreter kernels
(model.hip), values and EVERY derivative against the oracle's dual numbers (oracle/dual.py: what gvar.valder gives the
reference, src/lsqfit/_gsl.pyx:742-760; numpy.fabs on fit-function output: src/lsqfit/_extras.py:2569); and the model these
are mostly wanted for: a periodic two-state correlator, sum_n a_n cosh(E_n (t - T/2)) / cosh(E_n T/2), fitted to correlated
data on the device and by the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import dual
from oracle import fit as ofit
from tests import gpu_util as gu
from tests.test_gpu_jit_fuzz import oracle_values

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

TERMS = ['{a}*cosh({b}*(x - 1))', '{a}*sinh({b}*x)/{s}', '{a}*tanh({b}*x + {s})', '{a}*tan(0.3*{b}*x)', 'arcsin(0.2*{a}*x*{b})',
         '{a}*arccos(0.2*{b}*x)', 'abs({a}*x - {b})', '{a}*cosh({k}*x)/cosh({b})', 'fabs(sinh({a}) - {b}*x)*tanh({s})']


@pytest.fixture(scope='module')
def amd():
    import lsqfit_amd
    from lsqfit_amd import _lib
    _lib.load()
    return lsqfit_amd


def formula(seed):
    rng = np.random.default_rng(7000 + seed)
    K = int(rng.choice([2, 5, 16, 40]))
    tmpl = TERMS[seed % len(TERMS)]
    a = ['a%d' % k for k in range(K)]
    b = ['b%d' % k for k in range(K)]
    names = a + b if seed % 2 else [v for pair in zip(a, b) for v in pair]
    text = ' + '.join('(' + tmpl.format(a=a[k], b=b[k], s='sh', k=repr(0.25 * (k + 1))) + ')' for k in range(K))
    if 'sh' in text:
        names.append('sh')
    if seed % 3 == 0:
        text = 'cc*tanh(%s) + cosh(cc*x)' % text
        names.append('cc')
    return text, names, rng


def check(amd, seed):
    text, names, rng = formula(seed)
    P, N = len(names), 200
    x = np.sort(rng.uniform(0.05, 2.0, N))
    p = rng.uniform(0.3, 1.5, P)
    ysd = rng.uniform(0.5, 2.0, N)
    f0, J0 = oracle_values(text, names, x, p)
    ym = f0 + 0.1 * rng.standard_normal(N)
    pr = amd.DeviceProblem(amd.expr(text, names), x, amd.Whitening(ym, ysd))
    compiled = bool(pr.lib.lsqamd_debug_flags(pr.h) & 8)
    chi2 = pr.normal(p)
    J, f = pr.get_J_data(), pr.get_f_data()
    want_f, want_J = (f0 - ym) / ysd, J0 / ysd[:, None]
    scale = np.abs(want_J).max(axis=0) + 1e-300
    assert np.max(np.abs(J - want_J) / scale) < 1e-11, (text[:100], compiled)
    assert np.max(np.abs(f - want_f)) < 1e-11 * max(1.0, np.abs(want_f).max())
    assert chi2 == pytest.approx(float(want_f @ want_f), rel=1e-12)
    assert abs(pr.chi2(p) - chi2) < 1e-11 * max(1.0, chi2)
    pr.close()
    return compiled


@pytest.mark.parametrize('seed', range(18))
def test_new_functions_compiled(amd, seed):
    assert check(amd, seed), 'the formula was not compiled'


@pytest.mark.parametrize('knob', ['i', 'w', 'f'])
def test_new_functions_interpreted(knob):
    """The interpreter kernels -- by segment (i), whole tape (w), forward mode (f); the knob is read once per process."""
    prog = ('import sys\nsys.path.insert(0, %r)\nimport lsqfit_amd as amd\nfrom lsqfit_amd import _lib\n_lib.load()\n'
            'from tests.test_gpu_cosh import check\n'
            'for seed in range(9):\n    assert not check(amd, seed)\nprint("ok")\n' % ROOT)
    r = subprocess.run([sys.executable, '-c', prog], env=dict(os.environ, LSQAMD_TAPE=knob), cwd=ROOT, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and 'ok' in r.stdout, r.stderr[-2000:]


T = 32
TEXT = 'a0*cosh(E0*(x - 16))/cosh(E0*16) + a1*cosh((E0 + dE)*(x - 16))/cosh((E0 + dE)*16)'
NAMES = ['a0', 'E0', 'a1', 'dE']


def correlator(seed=3):
    rng = np.random.default_rng(seed)
    t = np.arange(1.0, T)
    pt = np.array([1.0, 0.25, 0.6, 0.35])
    g = pt[0] * np.cosh(pt[1] * (t - 16)) / np.cosh(pt[1] * 16) + pt[2] * np.cosh((pt[1] + pt[3]) * (t - 16)) / np.cosh((pt[1] + pt[3]) * 16)
    sd = 0.01 * g
    cov = np.outer(sd, sd) * 0.7 ** np.abs(np.subtract.outer(np.arange(t.size), np.arange(t.size)))
    y = g + np.linalg.cholesky(cov) @ rng.standard_normal(t.size)
    prior = (np.array([1.0, 0.3, 0.5, 0.4]), np.array([0.5, 0.2, 0.5, 0.3]))
    return t, y, cov, prior, pt


def fcn(x, p):
    E1 = p[1] + p[3]
    return p[0] * dual.cosh(p[1] * (x - 16)) / dual.cosh(p[1] * 16) + p[2] * dual.cosh(E1 * (x - 16)) / dual.cosh(E1 * 16)


@pytest.mark.parametrize('route', ['one_launch', 'general'])
@pytest.mark.parametrize('solver', ['cholesky', 'qr'])
def test_two_state_cosh_correlator(amd, route, solver, monkeypatch):
    monkeypatch.setenv('LSQAMD_ONE_LAUNCH_FIT', '1' if route == 'one_launch' else '0')
    t, y, cov, prior, pt = correlator()
    fit = amd.nonlinear_fit(data=(t, y, dict(sdev=np.sqrt(np.diag(cov)), blocks=[(0, cov)])), model=amd.expr(TEXT, NAMES), prior=prior,
                            solver=solver)
    fl = fit.problem.lib.lsqamd_debug_flags(fit.problem.h)
    assert fl & 8 and bool(fl & 32) == (route == 'one_launch')
    ref = ofit.nonlinear_fit(t, y, cov, fcn, prior_mean=prior[0], prior_err=prior[1], solver=solver)
    assert abs(fit.nit - ref.nit) <= 1 and fit.stopping_criterion == ref.stopping_criterion and fit.dof == ref.dof
    assert np.all(np.abs(fit.pmean - ref.pmean) <= 1e-6 * np.abs(ref.pmean) + 1e-5 * ref.psdev)
    assert fit.chi2 == pytest.approx(ref.chi2, rel=1e-6) and gu.relmax(fit.cov, ref.cov) < 1e-6
    assert fit.logGBF == pytest.approx(ref.logGBF, rel=1e-6, abs=1e-6)
    assert np.all(np.abs(fit.pmean - pt) < 5 * fit.psdev) and fit.chi2 / fit.dof < 2.0        # (and it is a sensible fit)


def test_unknown_function_is_refused(amd):
    with pytest.raises(ValueError):
        amd.expr('a*erf(b*x)', ['a', 'b'])
