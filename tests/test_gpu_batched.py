"""-m gpu: the batched engine (lsqamdb_*, device-resident LM state, hipGraph rounds) against
the single-fit device path and the oracle."""
import numpy as np
import pytest

from oracle import fit as ofit
from tests import gpu_util as gu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def amd():
    import lsqfit_amd
    from lsqfit_amd import _lib
    _lib.load()
    return lsqfit_amd


def make(N, P, B, seed):
    from lsqfit_amd import synth
    d = synth.make_cosmix(N=N, P=P, seed=seed, block=0, prior_corr=False)
    K = P // 2
    pm, ps = d['prior']
    # empirical-Bayes style sweep: prior width on the amplitudes a_k differs per fit
    z = 0.1 * 10 ** (2.0 * np.arange(B) / max(B - 1, 1))
    psb = np.tile(ps, (B, 1))
    psb[:, :K] = z[:, None]
    pmb = np.tile(pm, (B, 1))
    return d, pmb, psb


@pytest.mark.parametrize('use_graph', [False, True])
def test_batched_matches_single_fits_and_oracle(amd, use_graph):
    d, pmb, psb = make(512, 32, 6, 51)
    bf = amd.BatchedFits(d['model'], d['x'], d['ymean'], d['yerr'], pmb, psb)
    p0 = np.tile(d['p0'], (6, 1))
    p0[3] += 0.02                                  # different starts as well as different priors
    out = bf.run(p0=p0, use_graph=use_graph)
    assert np.all(out['status'] == 0) and np.all(out['stopping_criterion'] >= 1)
    if use_graph:
        assert out['graph_rounds'] >= out['rounds'] - 1 > 0
    for b in range(6):
        single = amd.nonlinear_fit(data=(d['x'], d['ymean'], d['yerr']), model=d['model'],
                                   prior=(pmb[b], psb[b]), p0=p0[b])
        assert gu.relmax(out['pmean'][b], single.pmean) < 1e-10
        assert out['chi2'][b] == pytest.approx(single.chi2, rel=1e-10)
        assert out['nit'][b] == single.nit
        assert out['stopping_criterion'][b] == single.stopping_criterion
        assert gu.relmax(bf.cov(b), single.cov) < 1e-9
        assert out['logGBF'][b] == pytest.approx(single.logGBF, rel=1e-10, abs=1e-8)
        dd = dict(d, prior=(pmb[b], psb[b]), p0=p0[b])
        ref = gu.oracle_fit(dd, solver='cholesky')
        assert gu.relmax(out['pmean'][b], ref.pmean) < 1e-6
        assert gu.relmax(bf.cov(b), ref.cov) < 1e-6
        assert out['chi2'][b] / out['dof'] == pytest.approx(ref.chi2 / ref.dof, rel=1e-6)
        assert out['logGBF'][b] == pytest.approx(ref.logGBF, rel=1e-8, abs=1e-6)
    bf.close()


def test_batched_maxit_and_rejections(amd):
    """maxit retires fits that have not converged; a far start forces rejected trial steps;
    every counter agrees with the single-fit driver."""
    d, pmb, psb = make(256, 16, 4, 52)
    bf = amd.BatchedFits(d['model'], d['x'], d['ymean'], d['yerr'], pmb, psb)
    p0 = np.tile(d['p0'], (4, 1))
    p0[1, 8:] += 1.0                               # frequencies far off: LM must back off
    for maxit, tol in [(3, 1e-14), (40, 1e-8)]:
        out = bf.run(p0=p0, maxit=maxit, tol=tol)
        for b in range(4):
            single = amd.nonlinear_fit(data=(d['x'], d['ymean'], d['yerr']), model=d['model'],
                                       prior=(pmb[b], psb[b]), p0=p0[b], maxit=maxit, tol=tol)
            assert out['nit'][b] == single.nit
            assert out['stopping_criterion'][b] == single.stopping_criterion
            assert (out['status'][b] == 11) == (single.error is not None)
            assert gu.relmax(out['pmean'][b], single.pmean) < 1e-9
            # (trial counts are NOT compared: near the minimum, accepting a step whose chi2 change is
            # at rounding level depends on the reduction order of |f|^2 -- measured 12 vs 16, 25 vs 17)
            assert out['nfev'][b] >= out['nit'][b] + 1
        if maxit == 3:
            assert np.all(out['nit'] == 3) and np.all(out['status'] == 11)
        else:
            assert np.any(out['nfev'] > out['nit'] + 1)      # rejected trials happened somewhere
    bf.close()


def test_config5_sweep_128x4096x512(amd):
    """BASELINE.json configs[4]: 128 batched fits of 4096 x 512 with a hipGraph-captured LM
    round.  Three of the fits against the single-fit device path AND the oracle; the rest through the
    smoothness of logGBF(z) and chi2/dof."""
    d, pmb, psb = make(4096, 512, 128, 20264)
    bf = amd.BatchedFits(d['model'], d['x'], d['ymean'], d['yerr'], pmb, psb)
    out = bf.run(use_graph=True)
    assert np.all(out['status'] == 0)
    assert out['graph_rounds'] > 0
    # chi2 ~ N - P (+ prior part): 512 of the 4096 'dof' are absorbed by the parameters
    assert np.all((out['chi2'] / out['dof'] > 0.6) & (out['chi2'] / out['dof'] < 1.6))
    g = out['logGBF']
    assert np.all(np.isfinite(g))
    assert np.abs(np.diff(g, 2)).max() < 0.05 * (g.max() - g.min()) + 1.0   # smooth in z
    for b in (0, 63, 127):
        single = amd.nonlinear_fit(data=(d['x'], d['ymean'], d['yerr']), model=d['model'],
                                   prior=(pmb[b], psb[b]))
        assert gu.relmax(out['pmean'][b], single.pmean) < 1e-9
        assert out['logGBF'][b] == pytest.approx(single.logGBF, rel=1e-9)
        assert out['nit'][b] == single.nit
        # ... and against the ORACLE (the CPU restatement of the reference's algorithm) at full size, to the
        # north_star tolerance: parameters, chi2/dof, covariance, logGBF
        ref = gu.oracle_fit(dict(d, prior=(pmb[b], psb[b])), solver='cholesky')
        assert gu.relmax(out['pmean'][b], ref.pmean) < 1e-6
        assert out['chi2'][b] / out['dof'] == pytest.approx(ref.chi2 / ref.dof, rel=1e-6)
        assert gu.relmax(bf.cov(b), ref.cov) < 1e-6
        assert out['logGBF'][b] == pytest.approx(ref.logGBF, rel=1e-6, abs=1e-6)
    print('config5: %d rounds, %.1f ms on device, %d fits' % (out['rounds'], out['device_ms'], 128))
    bf.close()


def test_batched_fits_of_a_compiled_tape_model(amd):
    """A sweep over a user formula: the same lockstep engine with the tape COMPILED (jit.hip) -- one launch per
    evaluation with the fits as blockIdx.y, whatever P is.  sum_256 a*cos(w*x) as a tape (P = 512, 1791 instructions:
    beyond the 1024 the interpreter fallback of batched fits takes) against the hand-written cosmix kernels."""
    from lsqfit_amd import models
    d, pmb, psb = make(2048, 512, 8, 53)
    tape = models.tape_sum('a*cos(w*x)', 256)
    assert len(tape.tape) > 1024
    a = amd.BatchedFits(d['model'], d['x'], d['ymean'], d['yerr'], pmb, psb)
    b = amd.BatchedFits(tape, d['x'], d['ymean'], d['yerr'], pmb, psb)
    oa, ob = a.run(use_graph=True), b.run(use_graph=True)
    assert np.all(ob['status'] == 0) and ob['graph_rounds'] > 0
    assert gu.relmax(ob['pmean'], oa['pmean']) < 1e-9
    assert np.allclose(ob['chi2'], oa['chi2'], rtol=1e-10) and np.allclose(ob['logGBF'], oa['logGBF'], rtol=1e-10, atol=1e-8)
    assert np.array_equal(ob['nit'], oa['nit'])
    assert gu.relmax(b.cov(3), a.cov(3)) < 1e-8
    print('batched tape model: %.1f ms on device, cosmix kernels %.1f ms' % (ob['device_ms'], oa['device_ms']))
    a.close()
    b.close()
