"""-m "not gpu": oracle/port.py (the threaded host port of the cosmix normal equations: what the bench's CPU baseline and the
full-size oracle checks run on) against the plain numpy restatement of tests/gpu_util.py and the eigen-whitened oracle PDF."""
import numpy as np
import pytest

from lsqfit_amd import synth
from oracle.port import CosmixPort
from tests import gpu_util as gu


@pytest.mark.parametrize('kw', [dict(N=1024, P=64, seed=3, block=128, prior_corr=True), dict(N=700, P=32, seed=4, block=0, prior_corr=False),
                                dict(N=2048, P=64, seed=5, block=2048, prior_corr=True), dict(N=900, P=16, seed=6, block=100, prior_corr=False)])
def test_port_equals_the_plain_restatement(kw):
    d = synth.make_cosmix(**kw)
    port = CosmixPort(d)
    ne, cf, ld = gu.numpy_normal_equations(d)
    p = d['p0'] + 0.01 * np.random.default_rng(1).standard_normal(d['p0'].size)
    A, g, c = port.normal_eq(p)
    A0, g0, c0 = ne(p)
    assert gu.relmax(A, A0) < 1e-12 and gu.relmax(g, g0) < 1e-12 and c == pytest.approx(c0, rel=1e-12)
    assert port.chi2_fn(p) == pytest.approx(c0, rel=1e-12) and port.logdet == pytest.approx(ld, rel=1e-12)
    assert np.array_equal(A, A.T)
    # and the oracle proper (gvar-PDF restatement, eigen whitening): same invariants
    c1, A1, g1, _, _ = gu.oracle_normal(d, p)
    tol = 1e-10 if kw['block'] == 0 else 1e-8
    assert gu.relmax(A, A1) < tol and gu.relmax(g, g1) < tol and c == pytest.approx(c1, rel=tol)
    assert set(port.phases) == {'trig', 'whiten', 'syrk', 'cholesky'} and port.phases['syrk'] > 0
    port.close()
