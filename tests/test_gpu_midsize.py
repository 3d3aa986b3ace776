"""-m gpu: mid-size problems (P = 128 .. 768, tile-aligned and not) device vs oracle.  These shapes are where
the tile-aligned machinery switches on one piece at a time -- the chained back substitution (P >= 256, a
multiple of 128), fused trailing-update launches with the pivot-wave diagonal kernel (P >= 640), the fused
whitening kernel in its 32- and 64-term forms, captured LM steps (second fit on the same handle) -- and the
small fuzz cases never reach them."""
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


CASES = [dict(N=1024, P=256, block=0, prior_corr=False),
         dict(N=1536, P=384, block=256, prior_corr=True),
         dict(N=768, P=640, block=128, prior_corr=True),
         dict(N=1500, P=300, block=100, prior_corr=False),       # nothing tile-aligned
         dict(N=1024, P=256, block=512, prior_corr=False),       # two large blocks: paired tile rows, 32-term tiles
         dict(N=40000, P=24, block=0, prior_corr=False),         # many rows, few parameters: 156 K-chunks, narrow J^T f
         dict(N=33000, P=128, block=100, prior_corr=True)]       # the same with one aligned tile and small blocks


@pytest.mark.parametrize('shape', CASES, ids=lambda s: 'N%d_P%d_B%d' % (s['N'], s['P'], s['block']))
def test_midsize_fit_matches_oracle(amd, shape):
    from lsqfit_amd import synth
    d = synth.make_cosmix(seed=1000 + shape['P'], **shape)
    P = shape['P']
    p0 = d['p0'] * (1 + 3e-4 * np.random.default_rng(shape['N']).standard_normal(P))
    wh = amd.Whitening(d['ymean'], d['yerr'], *d['prior'])
    pr = amd.DeviceProblem(d['model'], d['x'], wh)
    kw = dict(data=(d['x'], d['ymean'], d['yerr']), model=d['model'], prior=d['prior'], p0=p0, problem=pr, tol=1e-10)
    fit = amd.nonlinear_fit(**kw)
    again = amd.nonlinear_fit(**kw)                        # same handle: the captured steps replay from the first iteration
    assert fit.error is None and fit.nit >= 3
    assert np.array_equal(fit.pmean, again.pmean) and np.array_equal(fit.cov, again.cov) and fit.nit == again.nit
    ref = gu.oracle_fit(d, solver='cholesky', tol=1e-10, p0=p0)
    assert fit.dof == ref.dof
    assert fit.chi2 == pytest.approx(ref.chi2, rel=1e-8)
    A, g = pr.get_jtj(), pr.get_grad()
    dd = np.sqrt(np.diag(A))
    # both covariances come from normal equations: they agree to cond(J^T J) eps, 1e-6 when that is smaller
    cond = np.linalg.cond(A / np.outer(dd, dd))
    assert gu.relmax(fit.pmean, ref.pmean) < 1e-6
    assert gu.relmax(fit.cov, ref.cov) < max(1e-6, 50 * cond * 2.2e-16), cond
    assert fit.logGBF == pytest.approx(ref.logGBF, rel=1e-8, abs=1e-6)
    # the solve by itself against LAPACK (factorisation + chained / grouped back substitution)
    v = pr.solve_damped(1e-3, dd)
    vref = np.linalg.solve(A + 1e-3 * np.diag(dd ** 2), g)
    assert np.max(np.abs(np.abs(v) - np.abs(vref))) < 1e-8 * np.max(np.abs(vref))
    pr.close()


_KNOB_SCRIPT = r'''
import sys
import numpy as np
sys.path.insert(0, %(root)r)
import lsqfit_amd as amd
from lsqfit_amd import synth
out = {}
for name, shape in (('a', dict(N=1536, P=384, block=256, prior_corr=True)), ('b', dict(N=3328, P=3200, block=128, prior_corr=False))):
    d = synth.make_cosmix(seed=77, **shape)
    p0 = d['p0'] * (1 + 3e-4 * np.random.default_rng(5).standard_normal(shape['P']))
    fit = amd.nonlinear_fit(data=(d['x'], d['ymean'], d['yerr']), model=d['model'], prior=d['prior'], p0=p0,
                            tol=1e-10, maxit=6 if name == 'b' else 1000)
    out[name + '_p'] = fit.pmean
    out[name + '_cov'] = fit.cov[::97, ::89].copy()
    out[name + '_chi2'] = np.array([fit.chi2, fit.logGBF])
np.savez(sys.argv[1], **out)
'''


def _knob_run(tmp_path, extra, name):
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / 'knob.py'
    script.write_text(_KNOB_SCRIPT % dict(root=root))
    path = str(tmp_path / (name + '.npz'))
    r = subprocess.run([sys.executable, str(script), path], env=dict(os.environ, **extra), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return np.load(path)


@pytest.fixture(scope='module')
def default_path_result(tmp_path_factory):
    """The fits of _KNOB_SCRIPT on the default paths: one process for the whole module, not one per knob."""
    return _knob_run(tmp_path_factory.mktemp('knob_default'), {}, 'default')


@pytest.mark.parametrize('knob', ['LSQAMD_BACKSOLVE=g', 'LSQAMD_BACKSOLVE=s', 'LSQAMD_HOST_LM=1', 'LSQAMD_FUSE_MIN_TILES=-1',
                                  'LSQAMD_TRAIL_HALVES=1', 'LSQAMD_SYNTH_NARROW=100000000', 'LSQAMD_POTF2=v3'])
def test_developer_knobs_select_equivalent_paths(tmp_path, knob, default_path_result):
    """Every alternative path a developer knob selects (grouped / per-block back substitution, host-side LM
    bookkeeping, unfused factorisation, half tiles everywhere, 32-term whitening tiles everywhere, the four-wave
    diagonal kernel) gives the default path's fit to rounding: a P = 384 fit to convergence and six LM steps of a
    P = 3200 fit (25 tile rows: fused launches, half tiles, the chained back substitution)."""
    res = [default_path_result, _knob_run(tmp_path, dict([knob.split('=')]), 'knob')]
    for k in res[0].files:
        # different summation orders: the iterates agree to rounding amplified by the conditioning of the step;
        # the covariance (P = 3200 on 3328 data rows leans on the prior) to cond * eps
        tol = 1e-4 if k.endswith('_cov') else 1e-7
        scale = np.max(np.abs(res[0][k]))
        assert np.max(np.abs(res[0][k] - res[1][k])) <= tol * scale, (knob, k, np.max(np.abs(res[0][k] - res[1][k])) / scale)
