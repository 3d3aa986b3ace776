"""-m gpu: the row-sharded DEVICE path end to end with world_size 2 and 3.  RCCL refuses two ranks
on one device, so on a 1-GPU box the ranks share cuda:0 and the all-reduce hook runs over gloo on
the same float64 workspace views -- everything else is exactly what bench.py runs per rank under
torch.distributed.run: shard plan, per-rank DeviceProblem (fused J^T f, device-resident trial
step), reduce hook for the packed normal equations and for every trial chi2, replicated solve.
Checks: all ranks bit-identical to each other, equal to the unsharded device fit to 1e-9, plus
fit.p sensitivities (f1) and chi2 at many points (f2) on the shards."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _special(case):
    """Data whose covariance components interleave (rows 0, 7, 14.. tied together: the whitening reorders the rows and a shard is
    a range of the REORDERED rows), and data correlated with the prior (concat(y, prior) whitened as one vector, prior entries
    travelling as rows -- on whichever rank their component lands)."""
    from lsqfit_amd import synth
    d = synth.make_cosmix(N=420, P=12, seed=93, block=0, prior_corr=False)
    N, P = 420, 12
    sd = np.asarray(d['yerr'], float)
    rng = np.random.default_rng(7)
    cov = np.diag(sd ** 2)
    if case == 'interleaved':
        for start in range(7):                      # seven components of 20 rows each, stride 7, inside the first 140 rows
            idx = np.arange(start, 140, 7)
            cov[np.ix_(idx, idx)] = np.outer(sd[idx], sd[idx]) * 0.5 ** np.abs(np.subtract.outer(np.arange(idx.size), np.arange(idx.size)))
        d['yerr'] = cov
        return d, None
    cross = np.zeros((N, P))                        # every prior entry tied to three data rows of its own
    psd = np.asarray(d['prior'][1], float)
    for j in range(P):
        rows = 30 * j + np.array([3, 4, 11])
        cross[rows, j] = 0.3 * sd[rows] * psd[j] * rng.uniform(0.5, 1.0, 3)
    return d, cross


def _problem(case):
    from lsqfit_amd import synth
    if case in ('interleaved', 'cross'):
        return _special(case)[0]
    if case in ('blocks', 'trf', 'varpro', 'qr'):
        return synth.make_cosmix(N=1536, P=128, seed=91, block=256, prior_corr=True)
    return synth.make_cosmix(N=1000, P=30, seed=92, block=0, prior_corr=False)


def _fit_kw(case, P):
    if case == 'trf':        # bounded fit (amplitudes boxed): the reflective method on the shards
        K = P // 2
        lo = np.concatenate([np.full(K, 0.8), np.full(K, -np.inf)])
        hi = np.concatenate([np.full(K, 1.2), np.full(K, np.inf)])
        return dict(fitter='mi355x_trf', bounds=(lo, hi), tol=(1e-10, 1e-10, 1e-10))
    if case == 'varpro':     # variable projection of the amplitudes on the shards
        return dict(linear=np.arange(P // 2), tol=1e-10)
    if case == 'qr':         # QR-grade covariance: every Gram matrix of the orthogonalisation is all-reduced
        return dict(solver='qr')
    return dict(alg=('lm' if case == 'blocks' else 'dogleg'))


def _worker(rank, world, outdir, cases):
    """One set of `world` processes runs ALL the cases of that world size one after the other (a process costs an import of
    torch and a HIP context: on a slow box that, not the fits, is what these tests take)."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    # rendezvous through a file in the test's own directory: no port to lose a race for
    dist.init_process_group('gloo', init_method='file://' + os.path.join(outdir, 'rendezvous'), rank=rank, world_size=world)
    import lsqfit_amd as amd
    from lsqfit_amd.dist import sharded_problem
    for case in cases:
        d = _problem(case)
        if case == 'cross':
            from lsqfit_amd.whiten import joint_whitening
            wh = joint_whitening(d['ymean'], d['yerr'], d['prior'][0], d['prior'][1], _special(case)[1])
        else:
            wh = amd.Whitening(d['ymean'], d['yerr'], *d['prior'])
        pr = sharded_problem(d['model'], d['x'], wh, rank, world)
        fit = amd.nonlinear_fit(data=(d['x'], d['ymean'], d['yerr']), model=d['model'], prior=d['prior'],
                                problem=pr, **_fit_kw(case, d['p0'].size))
        extra = {}
        if case in ('interleaved', 'cross'):
            extra['f_rows'] = pr.fcn(fit.pmean)          # this shard's rows in the whitening's order
        G = np.random.default_rng(1).standard_normal((2, d['p0'].size))
        GD = pr.dpdy(G)                                  # this rank's data columns, then the prior's
        pts = fit.pmean + 1e-3 * np.random.default_rng(2).standard_normal((5, d['p0'].size))
        c2 = pr.chi2_points(pts)                         # reduced over ranks through the hook
        np.savez(os.path.join(outdir, '%s_r%d.npz' % (case, rank)), pmean=fit.pmean, cov=fit.cov, chi2=fit.chi2, nit=fit.nit,
                 logGBF=fit.logGBF, rows=np.array(pr.rows), GD=GD, c2=c2, **extra)
        pr.close()
        dist.barrier()
    dist.destroy_process_group()


CASES = {2: ['blocks', 'diag', 'trf', 'varpro', 'qr', 'interleaved', 'cross'], 3: ['blocks', 'interleaved', 'cross']}


@pytest.fixture(scope='module')
def runs(tmp_path_factory):
    """world -> directory with every case's per-rank results (the processes of a world size are started once, on first use)."""
    done = {}

    def get(world):
        if world not in done:
            import torch.multiprocessing as mp
            out = str(tmp_path_factory.mktemp('world%d' % world))
            ctx = mp.get_context('spawn')
            procs = [ctx.Process(target=_worker, args=(r, world, out, CASES[world])) for r in range(world)]
            for p in procs:
                p.start()
            for p in procs:
                p.join(900)
            for p in procs:
                if p.is_alive():
                    p.kill()
            done[world] = (out, [p.exitcode for p in procs])
        return done[world]
    return get


def _results(runs, case, world):
    out, codes = runs(world)
    files = [os.path.join(out, '%s_r%d.npz' % (case, r)) for r in range(world)]
    assert all(os.path.exists(f) for f in files), 'case %s did not finish on every rank (exit codes %s)' % (case, codes)
    return [np.load(f) for f in files]


@pytest.mark.parametrize('case,world', [('interleaved', 2), ('interleaved', 3), ('cross', 2), ('cross', 3)])
def test_sharded_fit_of_reordered_rows(case, world, runs):
    """Row sharding for data with interleaved covariance components and for data correlated with the prior (declared limits of
    rounds 2-3): shards are ranges of the whitening's reordered rows, never cutting a component."""
    import lsqfit_amd as amd
    res = _results(runs, case, world)
    for r in res[1:]:
        for k in ('pmean', 'cov', 'chi2', 'nit', 'logGBF', 'c2'):
            assert np.array_equal(res[0][k], r[k]), k
    d, cross = _special(case)
    ref = amd.nonlinear_fit(data=(d['x'], d['ymean'], d['yerr']), model=d['model'], prior=d['prior'], cross=cross)
    rel = lambda a, b: np.max(np.abs(a - b)) / np.max(np.abs(b))
    assert rel(res[0]['pmean'], ref.pmean) < 1e-9 and rel(res[0]['cov'], ref.cov) < 1e-8
    # (twelve parameters: the unsharded fit is ONE launch, DESIGN 16, whose last rounding-sized step may fall the other side of
    #  the xtol test than the general path's the shards take -- same end point, an iteration apart)
    assert abs(res[0]['chi2'] / ref.chi2 - 1) < 1e-10 and abs(int(res[0]['nit']) - ref.nit) <= 1
    assert res[0]['logGBF'] == pytest.approx(ref.logGBF, rel=1e-10)
    # the shards tile the reordered rows, whole components each; the function values they report are those of the unsharded fit
    wh = ref.whitening
    order = wh.row_src if case == 'cross' else wh.perm
    rows = [tuple(r['rows']) for r in res]
    assert rows[0][0] == 0 and rows[-1][1] == wh.n_data and all(a[1] == b[0] for a, b in zip(rows, rows[1:]))
    for b in wh.blocks:
        assert sum(1 for a, e in rows if a <= b['row0'] and b['row0'] + b['size'] <= e) == 1
    full = ref.problem.fcn(ref.pmean)
    if case == 'interleaved':
        full = full[order]                       # (unsharded fcn() hands the rows back in the caller's order)
    for r, (a, e) in enumerate(rows):
        got = res[r]['f_rows']
        assert got.shape == (e - a,) and np.allclose(got, full[a:e], rtol=1e-7, atol=1e-9)      # (at each fit's own end point: 1e-9 apart)
    pts = ref.pmean + 1e-3 * np.random.default_rng(2).standard_normal((5, d['p0'].size))
    assert rel(res[0]['c2'], ref.problem.chi2_points(pts)) < 1e-8


@pytest.mark.parametrize('case,world', [('blocks', 2), ('blocks', 3), ('diag', 2), ('trf', 2), ('varpro', 2), ('qr', 2)])
def test_sharded_device_fit(case, world, runs):
    import lsqfit_amd as amd
    res = _results(runs, case, world)
    for r in res[1:]:                                  # replicated decisions: bit-identical ranks
        for k in ('pmean', 'cov', 'chi2', 'nit', 'logGBF', 'c2'):
            assert np.array_equal(res[0][k], r[k]), k
    d = _problem(case)
    ref = amd.nonlinear_fit(data=(d['x'], d['ymean'], d['yerr']), model=d['model'], prior=d['prior'],
                            **_fit_kw(case, d['p0'].size))
    rel = lambda a, b: np.max(np.abs(a - b)) / np.max(np.abs(b))
    assert rel(res[0]['pmean'], ref.pmean) < 1e-9
    assert rel(res[0]['cov'], ref.cov) < 1e-8
    assert abs(res[0]['chi2'] / ref.chi2 - 1) < 1e-10
    if case == 'trf':     # dozens of reflections: sums in another order move the count a little
        assert abs(int(res[0]['nit']) - ref.nit) <= max(2, ref.nit // 4)
        assert np.any(np.minimum(ref.pmean[:64] - 0.8, 1.2 - ref.pmean[:64]) < 1e-6)
    else:
        assert int(res[0]['nit']) == ref.nit
    # f1 on shards: data columns of G D tile the unsharded ones; prior columns agree on every rank
    N, P = d['ymean'].size, d['p0'].size
    G = np.random.default_rng(1).standard_normal((2, P))
    GDref = ref.dp_dinputs(G)
    for r in res:
        a, b = r['rows']
        assert rel(r['GD'][:, :b - a], GDref[:, a:b]) < 1e-7
        assert rel(r['GD'][:, b - a:], GDref[:, N:]) < 1e-7
    pts = ref.pmean + 1e-3 * np.random.default_rng(2).standard_normal((5, P))
    assert rel(res[0]['c2'], ref.problem.chi2_points(pts)) < 1e-8
