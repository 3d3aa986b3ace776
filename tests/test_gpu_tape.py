"""-m gpu: the expression-tape model (the stand-in for the user's own fit function,
src/lsqfit/_gsl.pyx:742-760) at widths where forward-mode AD in 16-parameter passes would crawl:
reverse-mode Jacobian (one forward + one reverse sweep per row) against the analytic sum-model
kernels on the same function, and against central differences."""
import numpy as np
import pytest

from tests import gpu_util as gu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def amd():
    import lsqfit_amd
    from lsqfit_amd import _lib
    _lib.load()
    return lsqfit_amd


@pytest.mark.parametrize('N,P,block', [(1000, 64, 0), (4096, 1024, 256), (777, 38, 111), (16500, 512, 0)])   # (the last: two rows per lane)
def test_tape_jacobian_equals_analytic_kernel(amd, N, P, block):
    from lsqfit_amd import models, synth
    d = synth.make_cosmix(N=N, P=P, seed=5, block=block, prior_corr=False)
    tape = models.tape_sum('a*cos(w*x)', P // 2)
    assert tape.n_param == P and len(tape.tape) == 7 * (P // 2) - 1
    wh = amd.Whitening(d['ymean'], d['yerr'], *d['prior'])
    pa = amd.DeviceProblem(d['model'], d['x'], wh)
    pt = amd.DeviceProblem(tape, d['x'], wh)
    p = d['p_true'] * (1 + 1e-3 * np.random.default_rng(1).standard_normal(P))
    ca, ct = pa.normal(p), pt.normal(p)
    assert ct == pytest.approx(ca, rel=1e-12)
    assert gu.relmax(pt.get_jtj(), pa.get_jtj()) < 1e-11
    assert gu.relmax(pt.get_grad(), pa.get_grad()) < 1e-10
    assert gu.relmax(pt.get_J_data(), pa.get_J_data()) < 1e-11
    assert pt.chi2(p) == pytest.approx(pa.chi2(p), rel=1e-12)
    fa = amd.nonlinear_fit(data=(d['x'], d['ymean'], d['yerr']), model=d['model'], prior=d['prior'], problem=pa)
    ft = amd.nonlinear_fit(data=(d['x'], d['ymean'], d['yerr']), model=tape, prior=d['prior'], problem=pt)
    # (39 iterations at 16500 rows: rounding differences of 1e-11 in J reach 1.4e-8 in the covariance)
    assert ft.nit == fa.nit and gu.relmax(ft.pmean, fa.pmean) < 1e-9 and gu.relmax(ft.cov, fa.cov) < (1e-8 if N < 16384 else 1e-7)
    pa.close()
    pt.close()


def test_tape_repeated_parameters_and_every_opcode(amd):
    """A parameter that occurs several times on the tape (its adjoints add up) and every opcode, against
    central differences of the device's own function values."""
    rng = np.random.default_rng(2)
    N = 130
    x = np.stack([rng.uniform(0.5, 2.0, N), rng.uniform(0.1, 1.0, N)], axis=1)
    text = ('b1*exp(-b2*x) + b1*b1*sin(b3*z) - log(b4 + x*x)/(1 + b2*b2) + sqrt(b4)*arctan(b3*x)'
            ' + (b2*x)**b3 - cos(b1 - z)**3 + b4**-2')
    model = amd.expr(text, ['b1', 'b2', 'b3', 'b4'], xnames=('x', 'z'))
    p = np.array([1.3, 0.7, 1.9, 2.2])
    ym, ys = np.zeros(N), np.ones(N)
    wh = amd.Whitening(ym, ys)
    pr = amd.DeviceProblem(model, x, wh)
    pr.normal(p)
    J = pr.get_J_data()
    f0 = pr.fcn(p)
    for j in range(4):
        h = 1e-6 * max(1.0, abs(p[j]))
        e = np.zeros(4)
        e[j] = h
        fd = (pr.fcn(p + e) - pr.fcn(p - e)) / (2 * h)
        assert np.max(np.abs(J[:, j] - fd)) < 1e-7 * max(1.0, np.max(np.abs(fd))), j
    assert np.all(np.isfinite(f0))
    pr.close()


_SEG_SCRIPT = r'''
import sys, json
sys.path.insert(0, %(root)r)
import numpy as np
import lsqfit_amd as amd
rng = np.random.default_rng(11)
N = 700
x = np.stack([rng.uniform(0.5, 2.0, N), rng.uniform(0.1, 1.0, N)], axis=1)
cases = {
    # every parameter exactly once (columns stored, no zero fill)
    'once': ('a1*cos(w1*x) + a2*cos(w2*z) - a3*exp(-w3*x) + a4*sin(w4*x*z)', ['a1', 'w1', 'a2', 'w2', 'a3', 'w3', 'a4', 'w4']),
    # one parameter never read (its column must be zero), none read twice
    'unused': ('a1*cos(w1*x) - a2*z + w2*w2_0', ['a1', 'w1', 'a2', 'w2', 'dead', 'w2_0']),
    # parameters shared between the terms of the sum (adjoints accumulate across segments and chunks)
    'shared': ('a1*exp(-w1*x) + a1*a1*sin(w1*z) - a2/(1 + w1*w1) + a2*cos(a1*x) + (w1*x)**a2 - a1', ['a1', 'w1', 'a2']),
    # the root is not a sum: whole-tape kernel whatever the knob says
    'product': ('(a1 + w1*x)*(a2 - cos(w1*z))', ['a1', 'w1', 'a2']),
}
out = {}
for name, (text, names) in cases.items():
    model = amd.expr(text, names, xnames=('x', 'z'))
    p = rng.uniform(0.6, 1.7, len(names))
    wh = amd.Whitening(np.zeros(N), np.linspace(0.5, 2.0, N))
    pr = amd.DeviceProblem(model, x, wh)
    pr.normal(p)
    out[name] = dict(J=pr.get_J_data().tolist(), f=pr.fcn(p).tolist(), p=p.tolist())
    pr.close()
print('RESULT' + json.dumps(out))
'''


def test_segmented_tape_kernel_matches_the_whole_tape_and_forward_mode_kernels(tmp_path):
    """The kernel that differentiates a root-level sum term by term (parameters stored when each is read once,
    accumulated atomically otherwise; chunks of terms on different waves) against the whole-tape reverse kernel
    (LSQAMD_TAPE=w), the forward-mode kernel (LSQAMD_TAPE=f) and the two-rows-per-lane variant of itself
    (LSQAMD_TAPE_ROWS=2; 700 rows: a ragged last group) on the same tapes.  A process each: the knobs are read once."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / 'seg.py'
    script.write_text(_SEG_SCRIPT % dict(root=root))
    res = {}
    for knob in ('', 'w', 'f', 'rows2'):
        env = dict(os.environ)
        env.pop('LSQAMD_TAPE', None)
        env.pop('LSQAMD_TAPE_ROWS', None)
        if knob == 'rows2':
            env['LSQAMD_TAPE_ROWS'] = '2'        # two data rows per lane (the default from 16384 rows on)
        elif knob:
            env['LSQAMD_TAPE'] = knob
        out = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        line = [ln for ln in out.stdout.splitlines() if ln.startswith('RESULT')][-1]
        res[knob] = json.loads(line[len('RESULT'):])
    for name in res['']:
        J = np.array(res[''][name]['J'])
        assert np.all(np.isfinite(J))
        if name == 'unused':
            assert np.all(J[:, 4] == 0.0)
        for other in ('w', 'f', 'rows2'):
            Jo = np.array(res[other][name]['J'])
            scale = np.max(np.abs(Jo), axis=0) + 1e-300
            assert np.max(np.abs(J - Jo) / scale) < 1e-13, (name, other)
            fo = np.array(res[other][name]['f'])
            assert np.max(np.abs(np.array(res[''][name]['f']) - fo)) <= 1e-14 * np.max(np.abs(fo)), (name, other)


def test_compiled_formula_cache_is_bounded(monkeypatch):
    """A sweep whose literal constants change per data set builds a code object per variant: the cache keeps at most
    LSQAMD_JIT_CACHE_CAP loaded, unloads the ones no handle holds (least recently used first) and never one that is held."""
    import ctypes as C
    import lsqfit_amd as amd
    from lsqfit_amd import _lib
    lib = _lib.load()

    def stats():
        st = (C.c_int64 * 3)()
        assert lib.lsqamd_jit_cache_stats(st) == 0
        return st[0], st[1], st[2]
    rng = np.random.default_rng(3)
    x = np.linspace(0.1, 2.0, 50)
    sd = np.full(50, 0.01)
    keep_y = 1.3 * np.exp(-0.7 * x) + 0.25
    held = amd.nonlinear_fit(data=(x, keep_y, sd), model=amd.expr('a*exp(-b*x) + 0.25', ['a', 'b']), p0=[1.0, 1.0])
    monkeypatch.setenv('LSQAMD_JIT_CACHE_CAP', '8')
    ev0 = stats()[2]
    for k in range(30):
        c = 0.25 + 0.001 * (k + 1)                       # a new literal -> a new formula -> a new code object
        y = 1.3 * np.exp(-0.7 * x) + c + sd * rng.standard_normal(50)
        fit = amd.nonlinear_fit(data=(x, y, sd), model=amd.expr('a*exp(-b*x) + %r' % c, ['a', 'b']), p0=[1.0, 1.0])
        assert fit.problem.lib.lsqamd_debug_flags(fit.problem.h) & 8 and abs(fit.pmean[0] - 1.3) < 0.05
        fit.problem.close()
        del fit
        loaded, n_held, evicted = stats()
        assert loaded <= 8 and n_held >= 1
    assert stats()[2] - ev0 >= 20
    # the kernel of the problem that stayed open was never unloaded: it still runs, same bits
    again = amd.nonlinear_fit(data=(x, keep_y, sd), model=held.model, p0=[1.0, 1.0], problem=held.problem)
    assert np.array_equal(again.pmean, held.pmean) and again.chi2 == held.chi2
