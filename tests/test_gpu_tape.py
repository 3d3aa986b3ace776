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


@pytest.mark.parametrize('N,P,block', [(1000, 64, 0), (4096, 1024, 256), (777, 38, 111)])
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
    assert ft.nit == fa.nit and gu.relmax(ft.pmean, fa.pmean) < 1e-9 and gu.relmax(ft.cov, fa.cov) < 1e-8
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
