"""-m gpu: the row-sharded DEVICE path end to end with world_size 2 and 3.  RCCL refuses two ranks
on one device, so on a 1-GPU box the ranks share cuda:0 and the all-reduce hook runs over gloo on
the same float64 workspace views -- everything else is exactly what bench.py runs per rank under
torch.distributed.run: shard plan, per-rank DeviceProblem (fused J^T f, device-resident trial
step), reduce hook for the packed normal equations and for every trial chi2, replicated solve.
Checks: all ranks bit-identical to each other, equal to the unsharded device fit to 1e-9, plus
fit.p sensitivities (f1) and chi2 at many points (f2) on the shards."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem(case):
    from lsqfit_amd import synth
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


def _worker(rank, world, port, outdir, case):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    # rendezvous through a file in the test's own directory: no port to lose a race for
    dist.init_process_group('gloo', init_method='file://' + os.path.join(outdir, 'rendezvous'), rank=rank, world_size=world)
    import lsqfit_amd as amd
    from lsqfit_amd.dist import sharded_problem
    d = _problem(case)
    wh = amd.Whitening(d['ymean'], d['yerr'], *d['prior'])
    pr = sharded_problem(d['model'], d['x'], wh, rank, world)
    fit = amd.nonlinear_fit(data=(d['x'], d['ymean'], d['yerr']), model=d['model'], prior=d['prior'],
                            problem=pr, **_fit_kw(case, d['p0'].size))
    G = np.random.default_rng(1).standard_normal((2, d['p0'].size))
    GD = pr.dpdy(G)                                  # this rank's data columns, then the prior's
    pts = fit.pmean + 1e-3 * np.random.default_rng(2).standard_normal((5, d['p0'].size))
    c2 = pr.chi2_points(pts)                         # reduced over ranks through the hook
    np.savez(os.path.join(outdir, 'r%d.npz' % rank), pmean=fit.pmean, cov=fit.cov, chi2=fit.chi2, nit=fit.nit,
             logGBF=fit.logGBF, rows=np.array(pr.rows), GD=GD, c2=c2)
    pr.close()
    dist.destroy_process_group()


@pytest.mark.parametrize('case,world', [('blocks', 2), ('blocks', 3), ('diag', 2), ('trf', 2), ('varpro', 2), ('qr', 2)])
def test_sharded_device_fit(case, world, tmp_path):
    import torch.multiprocessing as mp
    import lsqfit_amd as amd
    port = _free_port()
    ctx = mp.get_context('spawn')
    procs = [ctx.Process(target=_worker, args=(r, world, port, str(tmp_path), case)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    res = [np.load(os.path.join(str(tmp_path), 'r%d.npz' % r)) for r in range(world)]
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
