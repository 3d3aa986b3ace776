"""-m "not gpu": the seam between lsqfit's side of the plugin call and the tracer, as far as it can be crossed without gvar and
without a GPU.  tests/lsqfit_protocol.py reproduces what ``nonlinear_fit`` hands a plugin (BufferDict over an object buffer,
the four flatfcn forms, ``chiv`` with numpy.concatenate / multiply / dot on object arrays, a gv-style function table);
``lsqfit_amd.trace.trace_residual`` -- what ``mi355x_lm(p0, nf, chiv)`` runs when it is called the way lsqfit calls it -- must
record that ``chiv`` into a tape whose values and derivatives are those of ``chiv`` itself (floats in, finite differences)."""
import numpy as np
import pytest

import lsqfit_amd as amd
import importlib
tr = importlib.import_module("lsqfit_amd.trace")
from tests import lsqfit_protocol as lp
from tests.helpers import load
from tests.test_trace import run_tape

KAT = load('kat.json')
NIST = load('nist.json')
EXAMPLES = {'simple': lambda: lp.simple_example(), 'p_corr': lambda: lp.p_corr_example(KAT['p_corr']),
            'x_err': lambda: lp.x_err_example(KAT['x_err']), 'y_vs_x_3': lambda: lp.y_vs_x_example(KAT['y_vs_x'], 3),
            'nist_misra1a': lambda: lp.nist_example('misra1a', NIST)[0], 'nist_nelson': lambda: lp.nist_example('nelson', NIST)[0],
            'nist_roszman1': lambda: lp.nist_example('roszman1', NIST)[0], 'nist_enso': lambda: lp.nist_example('enso', NIST)[0]}


@pytest.mark.parametrize('name', sorted(EXAMPLES))
def test_chiv_as_lsqfit_builds_it_is_recorded_faithfully(name):
    ex = EXAMPLES[name]()
    p0, nf, chiv, pdf = lp.fitter_call(**ex)
    assert p0.dtype == np.float64 and p0.flags['C_CONTIGUOUS'] and nf == pdf.nchiv
    rec = tr.trace_residual(chiv, p0.size)
    assert rec.n_rows == nf
    # the plugin asked for the object branch with an object array of its own numbers -- exactly once
    assert chiv.calls == [(np.dtype(object), True)]
    rng = np.random.default_rng(3)
    for _ in range(3):
        p = p0 * (1 + 0.05 * rng.standard_normal(p0.size)) + 0.01 * rng.standard_normal(p0.size)
        want = np.asarray(chiv(p), float)                 # lsqfit's float branch (src/lsqfit/_utilities.pyx:81-83)
        got = run_tape(rec.model, rec.x, p)
        # (1e-10: the eigen-whitened rows of y-vs-x's 8 x 8 block are sums of terms 1e6 times their result, in another order)
        np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-10 * np.max(np.abs(want)))


def test_protocol_pieces_behave_like_their_originals():
    """BufferDict over an object buffer (src/lsqfit/__init__.py:2005,:2009-2012), buf swapping (:2026,:2040), dictionary outputs
    copied key by key (:2034-2036), gv functions on scalars, arrays and objects with the method."""
    d = lp.BufferDict()
    d['a'] = np.array(1.5)
    d['v'] = np.array([1.0, 2.0, 3.0])
    d['m'] = np.arange(4.0).reshape(2, 2)
    assert d.size == 8 and d.shape is None and list(d.keys()) == ['a', 'v', 'm']
    assert d['a'] == 1.5 and d['m'].shape == (2, 2)
    o = lp.BufferDict(d, buf=np.zeros(8, float))
    o.buf = np.arange(8.0) * 2
    assert o['a'] == 0.0 and list(o['v']) == [2.0, 4.0, 6.0] and o['m'][1, 1] == 14.0
    yo = lp.BufferDict(d, buf=8 * [None])
    assert yo.buf.dtype == object
    yo['v'] = [7, 8, 9]
    assert list(yo.flat)[1:4] == [7, 8, 9]

    class WithMethod:
        def exp(self):
            return 'called'
    assert lp.gv.exp(WithMethod()) == 'called'
    assert lp.gv.exp(0.0) == 1.0
    assert np.allclose(lp.gv.exp(np.array([0.0, 1.0])), [1.0, np.e])
    assert list(lp.gv.exp(np.array([WithMethod(), WithMethod()], object))) == ['called', 'called']


def test_user_errors_inside_fcn_surface_through_the_plugin():
    """Python exceptions raised by the user's function are stashed and re-raised by the reference's plugin after the driver
    returns (src/lsqfit/_gsl.pyx:680-685,:738-740; tests/test_lsqfit.py:1684-1698); here they surface from the recording."""
    ex = lp.p_corr_example(KAT['p_corr'])

    def bad(x, p):
        raise ZeroDivisionError('user bug')
    ex['fcn'] = bad
    p0, nf, chiv, pdf = lp.fitter_call(**ex)
    with pytest.raises(ZeroDivisionError, match='user bug'):
        tr.trace_residual(chiv, p0.size)
