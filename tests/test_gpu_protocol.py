"""-m gpu: the plugin behind lsqfit's OWN side of the call, as far as this image allows (no gvar: SURVEY.md 8c).

tests/lsqfit_protocol.py builds ``p0, nf, chiv`` exactly as ``nonlinear_fit.__init__`` does (src/lsqfit/__init__.py:539-575:
BufferDict parameters / data, the four flatfcn forms, ``chiv`` with numpy.concatenate / multiply / dot on object arrays, user
functions written with ``gv.exp``) and then makes the reference's call, with nothing else:

    fit = FITTERS[name](p0, nf, chiv, tol=tol, maxit=maxit)                              (src/lsqfit/__init__.py:662-664)

The plugin records ``chiv`` itself and fits on the device; the caller's reduction of ``fit.f / fit.J / fit.cov / fit.x``
(:665-679,:709-725) must print the reference's own example outputs: examples/simple.out, p-corr.out, x-err.out."""
import types

import numpy as np
import pytest

from oracle import gvar_lite
from tests import lsqfit_protocol as lp
from tests.helpers import load
from tests.test_oracle_kat import check_header, parse_parameter_table

pytestmark = pytest.mark.gpu
KAT = load('kat.json')


@pytest.fixture(scope='module')
def fitters():
    import lsqfit_amd
    from lsqfit_amd import _lib, fitter
    _lib.load()
    registry = {}

    class FakeLsqfit:                 # what register() needs of the lsqfit module: nonlinear_fit.FITTERS (:453)
        class nonlinear_fit:
            FITTERS = registry
    fitter.register(FakeLsqfit)
    return registry


def _fit(fitters, ex, name='mi355x_lm', **fitterargs):
    p0, nf, chiv, pdf = lp.fitter_call(**ex)
    fit = fitters[name](p0, nf, chiv, tol=1e-8, maxit=1000, **fitterargs)     # lsqfit's defaults (:148-160)
    return types.SimpleNamespace(**lp.reduce(fit, pdf, p0.size)), chiv, fit


def test_simple_example_through_the_protocol(fitters):
    r, chiv, fit = _fit(fitters, lp.simple_example())
    out = KAT['simple']['out']
    check_header(r, out)                                            # chi2/dof 0.17 [5], Q 0.97, logGBF 0.65538
    assert [gvar_lite.fmt(m, s) for m, s in zip(r.pmean, r.psdev)] == ['0.253(32)', '0.449(65)']      # simple.out's parameter table
    assert 'a   0.253 (32)' in out and 'b   0.449 (65)' in out
    assert (np.dtype(object), True) in chiv.calls                   # the plugin differentiated chiv by recording it
    assert r.description.startswith('methods = lm/more/') and r.stopping_criterion in (1, 2, 3)
    assert fit.f.shape == (7,) and fit.J.shape == (7, 2) and fit.cov.shape == (2, 2)
    assert np.allclose(np.linalg.inv(fit.J.T @ fit.J), fit.cov, rtol=1e-6)       # "J.T @ J = inv of self.cov" (:668)


def test_p_corr_example_through_the_protocol(fitters):
    r, chiv, fit = _fit(fitters, lp.p_corr_example(KAT['p_corr']))
    check_header(r, KAT['p_corr']['out'])                           # chi2/dof 0.61 [11], Q 0.82, logGBF 19.129
    assert gvar_lite.fmt_array(r.pmean, r.psdev) == '[0.149(17) 2.97(34) 1.23(61) 0.59(15)]'
    assert '%.4f' % (r.cov[0, 1] / np.sqrt(r.cov[0, 0] * r.cov[1, 1])) == '0.9571'
    # the same through the reference's default solver name and through the scipy-style plugin
    r2, _, _ = _fit(fitters, lp.p_corr_example(KAT['p_corr']), solver='qr')
    assert r2.description == 'methods = lm/more/qr' and gvar_lite.fmt_array(r2.pmean, r2.psdev) == '[0.149(17) 2.97(34) 1.23(61) 0.59(15)]'
    r3, _, _ = _fit(fitters, lp.p_corr_example(KAT['p_corr']), name='mi355x_trf')
    assert gvar_lite.fmt_array(r3.pmean, r3.psdev) == '[0.149(17) 2.97(34) 1.23(61) 0.59(15)]'


def test_x_err_example_through_the_protocol(fitters):
    r, chiv, fit = _fit(fitters, lp.x_err_example(KAT['x_err']))
    out = KAT['x_err']['out']
    check_header(r, out)                                            # chi2/dof 0.35 [15], Q 0.99, logGBF -40.156
    assert [gvar_lite.fmt(m, s) for m, s in zip(r.pmean, r.psdev)] == parse_parameter_table(out)
    assert r.nit in (12, 13)                                        # x-err.out prints 13 (no reference test asserts it)


@pytest.mark.parametrize('nexp', [2, 3])
def test_y_vs_x_example_through_the_protocol(fitters, nexp):
    """examples/y-vs-x.py / .out: the one reference fixture with a MODIFIED svd mode (svdcut/n = 1e-12/1: the 8 x 8 block is
    whitened in its eigen basis, a non-triangular weight matrix inside chiv), dictionary parameters, a Python sum over
    zip(a, E) of ai * np.exp(-Ei * x)."""
    r, chiv, fit = _fit(fitters, lp.y_vs_x_example(KAT['y_vs_x'], nexp))
    want_p = {2: '[0.4024(40) 0.4471(46) 0.90104(51) 1.8282(14)]',
              3: '[0.4019(40) 0.406(14) 0.61(36) 0.90039(54) 1.8026(82) 2.83(19)]'}[nexp]
    assert gvar_lite.fmt_array(r.pmean, r.psdev) == want_p
    assert '%.5g' % r.logGBF == {2: '111.69', 3: '116.29'}[nexp] and r.dof == 8
    assert '%.2g' % (r.chi2 / r.dof) == {2: '2.2', 3: '0.63'}[nexp]


NIST = load('nist.json')


@pytest.mark.parametrize('name', sorted(NIST))
def test_nist_problems_through_the_protocol(fitters, name):
    """examples/nist.py: the 27 StRD problems with their functions written as the reference writes them (b1, b2, ... = b; gvar's
    exp / cos / arctan), handed to the plugin as lsqfit hands them over: nist.out's parameter strings (or agreement to sigma / 10,
    the reference's own criterion, examples/nist.py:85-99), the certified values, dof, logGBF."""
    ex, pr = lp.nist_example(name, NIST)
    p0, nf, chiv, pdf = lp.fitter_call(**ex)
    fit = fitters['mi355x_lm'](p0, nf, chiv, tol=pr['tol'], maxit=1000, solver='qr')      # the reference's default solver
    r = types.SimpleNamespace(**lp.reduce(fit, pdf, p0.size))
    got = gvar_lite.fmt_array(r.pmean, r.psdev)
    em, es = gvar_lite.parse_array(pr['expected_p'][1:-1].split())
    if got != pr['expected_p']:
        assert np.all(np.abs(r.pmean - em) <= np.maximum(es, r.psdev) / 10.), (got, pr['expected_p'])
    assert np.all(np.abs(r.pmean - pr['certified']) <= 1e-2 * pr['certified_sd'] + 1e-9 * np.abs(pr['certified']))
    np.testing.assert_allclose(r.psdev, pr['certified_sd'], rtol=2e-3)
    assert r.dof == pr['out']['dof'] and r.description == 'methods = lm/more/qr'
    if name != 'lanczos1':          # (sigma_y = 8.9e-14: chi2 is roundoff, examples/nist.py:16-19)
        assert '%.5g' % r.logGBF == pr['out']['logGBF'], (r.logGBF, pr['out']['logGBF'])
