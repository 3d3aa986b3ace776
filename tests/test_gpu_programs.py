"""-m gpu: one formula per range of data rows (lsqamd_set_tape_programs / lsqfit_amd.piecewise) -- the
flattened form of a fit function that returns a dictionary (src/lsqfit/__init__.py:1997-2042,
examples/simple.py).  Checked against the same fit written with a selector column (every row evaluating
every formula), against the oracle, through the compiled route and through the interpreter fallback,
sharded over two ranks, and at many parameter points (the batched residual path)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import gpu_util as gu

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def amd():
    import lsqfit_amd
    from lsqfit_amd import _lib
    _lib.load()
    return lsqfit_amd


def three_part_problem(amd, blocks=False):
    """60 rows a*exp(-b*x), 50 rows c + d*x**2, 40 rows a*cos(d*x) -- shared parameters across the parts"""
    rng = np.random.default_rng(17)
    n = (60, 50, 40)
    x = np.concatenate([np.linspace(0.1, 3, n[0]), np.linspace(-1, 1, n[1]), np.linspace(0, 2, n[2])])
    pt = np.array([1.3, 0.7, 0.4, 1.9])
    a, b, c, d = pt
    f = np.concatenate([a * np.exp(-b * x[:60]), c + d * x[60:110] ** 2, a * np.cos(d * x[110:])])
    sd = 0.01 + 0.02 * rng.random(150)
    if blocks:      # 30-row correlated blocks (they cross the 60 | 50 | 40 boundaries at 60 only: 90-120 spans two parts)
        cov = np.diag(sd ** 2)
        for r0 in range(0, 150, 30):
            u = rng.uniform(0.1, 0.9, (30, 60))
            cc = u @ u.T
            dd = np.sqrt(np.diag(cc))
            cov[r0:r0 + 30, r0:r0 + 30] = cc / np.outer(dd, dd) * np.outer(sd[r0:r0 + 30], sd[r0:r0 + 30])
        y = f + np.linalg.cholesky(cov) @ rng.standard_normal(150)
        yerr = cov
    else:
        y = f + sd * rng.standard_normal(150)
        yerr = sd
    parts = [(60, 'a*exp(-b*x)'), (50, 'c + d*x**2'), (40, 'a*cos(d*x)')]
    pw = amd.piecewise(parts, ['a', 'b', 'c', 'd'])
    sel = np.zeros((150, 4))
    sel[:, 0] = x
    sel[:60, 1] = 1
    sel[60:110, 2] = 1
    sel[110:, 3] = 1
    one = amd.expr('s1*a*exp(-b*x) + s2*(c + d*x**2) + s3*a*cos(d*x)', ['a', 'b', 'c', 'd'], xnames=('x', 's1', 's2', 's3'))
    prior = (np.array([1.0, 1.0, 0.0, 2.0]), np.array([1.0, 1.0, 1.0, 1.0]))
    return dict(x=x, y=y, yerr=yerr, pw=pw, sel=sel, one=one, prior=prior, pt=pt)


@pytest.mark.parametrize('blocks', [False, True])
def test_programs_equal_the_selector_formulation(amd, blocks):
    q = three_part_problem(amd, blocks)
    fit = amd.nonlinear_fit(data=(q['x'], q['y'], q['yerr']), model=q['pw'], prior=q['prior'])
    ref = amd.nonlinear_fit(data=(q['sel'], q['y'], q['yerr']), model=q['one'], prior=q['prior'])
    assert fit.problem.lib.lsqamd_debug_flags(fit.problem.h) & 8          # all three formulas run compiled
    assert gu.relmax(fit.pmean, ref.pmean) < 1e-10
    assert gu.relmax(fit.cov, ref.cov) < 1e-9
    assert fit.chi2 == pytest.approx(ref.chi2, rel=1e-10) and fit.nit == ref.nit
    assert fit.logGBF == pytest.approx(ref.logGBF, rel=1e-10)
    assert np.all(np.abs(fit.pmean - q['pt']) < 5 * np.sqrt(np.diag(fit.cov)))
    # chi2 at many points: the batched residual path (forward interpreter kernel per program)
    pts = fit.pmean + 1e-2 * np.random.default_rng(3).standard_normal((7, 4))
    assert np.allclose(fit.dchi2(pts), ref.dchi2(pts), rtol=1e-9, atol=1e-9)
    # unwhitened model values
    a, b, c, d = fit.pmean
    want = np.concatenate([a * np.exp(-b * q['x'][:60]), c + d * q['x'][60:110] ** 2, a * np.cos(d * q['x'][110:])])
    assert gu.relmax(fit.problem.fcn(fit.pmean), want) < 1e-13


def test_programs_through_the_interpreter_fallback(amd, tmp_path):
    """LSQAMD_TAPE=i (no hiprtc): every range runs the forward-mode interpreter kernel; same fit."""
    q = three_part_problem(amd)
    fit = amd.nonlinear_fit(data=(q['x'], q['y'], q['yerr']), model=q['pw'], prior=q['prior'])
    prog = ('import sys, numpy as np\n'
            'sys.path.insert(0, %r)\n'
            'import lsqfit_amd as amd\n'
            'from tests.test_gpu_programs import three_part_problem\n'
            'q = three_part_problem(amd)\n'
            'fit = amd.nonlinear_fit(data=(q["x"], q["y"], q["yerr"]), model=q["pw"], prior=q["prior"])\n'
            'assert not (fit.problem.lib.lsqamd_debug_flags(fit.problem.h) & 8)\n'
            'np.savez(%r, pmean=fit.pmean, cov=fit.cov, chi2=fit.chi2)\n' % (ROOT, str(tmp_path / 'out.npz')))
    r = subprocess.run([sys.executable, '-c', prog], env=dict(os.environ, LSQAMD_TAPE='i'), cwd=ROOT, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = np.load(str(tmp_path / 'out.npz'))
    assert gu.relmax(out['pmean'], fit.pmean) < 1e-9 and gu.relmax(out['cov'], fit.cov) < 1e-8
    assert float(out['chi2']) == pytest.approx(fit.chi2, rel=1e-10)


def test_programs_on_a_row_shard(amd):
    """rows (45, 120) of the three-part problem: the ranges are clipped to the shard (15 + 50 + 10 rows)"""
    q = three_part_problem(amd)
    wh = amd.Whitening(q['y'], q['yerr'], *q['prior'])
    pr = amd.DeviceProblem(q['pw'], q['x'], wh, rows=(45, 120))
    ref = amd.DeviceProblem(q['one'], q['sel'], wh, rows=(45, 120))
    p = np.array([1.1, 0.8, 0.3, 2.1])
    assert pr.chi2(p) == pytest.approx(ref.chi2(p), rel=1e-12)
    assert pr.normal(p) == pytest.approx(ref.normal(p), rel=1e-12)
    assert gu.relmax(pr.get_jtj(), ref.get_jtj()) < 1e-12 and gu.relmax(pr.get_grad(), ref.get_grad()) < 1e-12
    pr.close()
    ref.close()


def test_bad_program_tables_are_refused(amd):
    q = three_part_problem(amd)
    bad = amd.piecewise([(60, 'a*exp(-b*x)'), (50, 'c + d*x**2'), (30, 'a*cos(d*x)')], ['a', 'b', 'c', 'd'])
    with pytest.raises(ValueError, match='covers 140 rows'):
        amd.nonlinear_fit(data=(q['x'], q['y'], q['yerr']), model=bad, prior=q['prior'])
