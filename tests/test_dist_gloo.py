"""N > 1 path on CPU: world_size-2/3 gloo processes run the row-sharded fit with the
product's shard plan and all-reduce hook; the per-shard arithmetic is the oracle's
(the HIP kernels need a GPU).  Checks: sharded == unsharded, every rank identical."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    # rendezvous through a file in the test's own directory: no port to lose a race for
    dist.init_process_group('gloo', init_method='file://' + os.path.join(outdir, 'rendezvous'), rank=rank, world_size=world)
    from lsqfit_amd import synth
    from lsqfit_amd.dist import WorkspaceView, make_reduce_hook, shard_rows
    from lsqfit_amd.whiten import Whitening
    from oracle import lm as olm
    from tests import gpu_util as gu

    d = synth.make_cosmix(N=192, P=12, seed=77, block=32, prior_corr=True)
    wh = Whitening(d['ymean'], d['yerr'], *d['prior'])
    a, b = shard_rows(wh.n_data, [(k['row0'], k['size']) for k in wh.blocks], world)[rank]
    P = d['p0'].size
    # this rank's whitening rows (block-diagonal whitening restricted to its blocks)
    W = np.zeros((b - a, b - a))
    for k in wh.blocks:
        if a <= k['row0'] < b:
            r0 = k['row0'] - a
            W[r0:r0 + k['size'], r0:r0 + k['size']] = k['Wt'].T
    x, ym = d['x'][a:b], d['ymean'][a:b]
    pm, prec = wh.prior_mean, wh.prior_prec
    adds_prior = rank == 0                      # lsqamd_set_adds_prior convention

    ws = torch.zeros(8 * (P * P + P + 1) + 64, dtype=torch.uint8)   # stands in for the device workspace
    view = WorkspaceView(ws)
    hook = make_reduce_hook(view, group=None, sync=None)
    base = ws.data_ptr()

    def normal_eq(p):
        J = W @ gu.cosmix_jac(x, p)
        r = W @ (gu.cosmix_fcn(x, p) - ym)
        A, g, c2 = J.T @ J, J.T @ r, float(r @ r)
        if adds_prior:
            dp = p - pm
            A, g, c2 = A + prec, g + prec @ dp, c2 + float(dp @ prec @ dp)
        buf = view(base, P * P + P + 1)
        buf[:P * P] = torch.from_numpy(A.reshape(-1))
        buf[P * P:P * P + P] = torch.from_numpy(g)
        buf[P * P + P] = c2
        hook(base, P * P + P + 1)                # in-place sum over ranks
        out = buf.numpy().copy()
        return out[:P * P].reshape(P, P), out[P * P:P * P + P], float(out[-1])

    def chi2_fn(p):
        r = W @ (gu.cosmix_fcn(x, p) - ym)
        c2 = float(r @ r)
        if adds_prior:
            dp = p - pm
            c2 += float(dp @ prec @ dp)
        buf = view(base, 1)
        buf[0] = c2
        hook(base, 1)
        return float(buf[0])

    res = olm.lm_normal(d['p0'], normal_eq, chi2_fn, tol=(1e-10, 1e-10, 1e-10))
    np.savez(os.path.join(outdir, 'rank%d.npz' % rank), x=res.x, cov=res.cov, chi2=res.fnorm2, nit=res.nit,
             rows=np.array([a, b]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_sharded_fit_equals_unsharded(tmp_path, world):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    outs = [np.load(tmp_path / ('rank%d.npz' % r)) for r in range(world)]
    # every rank took identical decisions and holds identical results (replicated solve)
    for o in outs[1:]:
        assert np.array_equal(o['x'], outs[0]['x']) and np.array_equal(o['cov'], outs[0]['cov'])
        assert int(o['nit']) == int(outs[0]['nit'])
    rows = sorted(tuple(o['rows']) for o in outs)
    assert rows[0][0] == 0 and rows[-1][1] == 192 and all(r[0] % 32 == 0 for r in rows)
    # and they equal the unsharded oracle fit
    sys.path.insert(0, ROOT)
    from lsqfit_amd import synth
    from tests import gpu_util as gu
    d = synth.make_cosmix(N=192, P=12, seed=77, block=32, prior_corr=True)
    ref = gu.oracle_fit(d, solver='cholesky', tol=(1e-10, 1e-10, 1e-10))
    assert gu.relmax(outs[0]['x'], ref.pmean) < 1e-8
    assert gu.relmax(outs[0]['cov'], ref.cov) < 1e-7
    assert float(outs[0]['chi2']) == pytest.approx(ref.chi2, rel=1e-8)


def test_reduce_hook_rejects_foreign_pointer():
    import torch
    from lsqfit_amd.dist import WorkspaceView
    ws = torch.zeros(256, dtype=torch.uint8)
    v = WorkspaceView(ws)
    assert v(ws.data_ptr() + 16, 4).numel() == 4
    with pytest.raises(ValueError):
        v(ws.data_ptr() + 250, 4)
    with pytest.raises(ValueError):
        v(ws.data_ptr() - 8, 1)


def test_headline_blocks_over_eight_ranks():
    """C4: 65536 rows in 256 covariance blocks of 256 -> 8 ranks, 32 whole blocks (8192 rows) each; ragged cases never cut a block."""
    from lsqfit_amd.dist import shard_rows
    blocks = [(r0, 256) for r0 in range(0, 65536, 256)]
    ranges = shard_rows(65536, blocks, 8)
    assert ranges == [(k * 8192, (k + 1) * 8192) for k in range(8)]
    # 250 blocks of 256 + 1536 loose rows: every boundary on a block edge or among the loose rows, rows balanced to a block
    n = 250 * 256 + 1536
    ranges = shard_rows(n, [(r0, 256) for r0 in range(0, 250 * 256, 256)], 8)
    assert ranges[0][0] == 0 and ranges[-1][1] == n and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    for a, b in ranges:
        assert a >= 250 * 256 or a % 256 == 0
        assert abs((b - a) - n / 8) <= 256


class _FakeHandle:
    """stands in for DeviceProblem's two communicator calls (the real ones need a GPU): ids are counted, joins recorded"""
    made = 0

    def __init__(self, log):
        self.log = log

    def comm_unique_id(self):
        _FakeHandle.made += 1
        return bytes([_FakeHandle.made]) * 128

    def comm_init(self, uid, rank, world):
        self.log.append(bytes(uid))


def _id_worker(rank, world, outdir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from lsqfit_amd import dist as ldist
    log = []
    dist.init_process_group('gloo', init_method='file://' + os.path.join(outdir, 'rdv1'), rank=rank, world_size=world)
    ldist.attach_rccl(_FakeHandle(log), rank, world)            # first problem: a new id
    ldist.attach_rccl(_FakeHandle(log), rank, world)            # second problem of the job: the same id, no new one made
    if rank == 1:
        ldist.forget_communicators()                            # ONE rank loses its memory (the advisor's deadlock: the others
    ldist.attach_rccl(_FakeHandle(log), rank, world)            #   skipped the broadcast this rank waited in): all agree on a new id
    sub = dist.new_group([0, 1])                                # a subgroup has its own entry
    ldist.attach_rccl(_FakeHandle(log), rank, world, group=sub)
    ldist.attach_rccl(_FakeHandle(log), rank, world, group=sub)
    dist.barrier()
    dist.destroy_process_group()
    dist.init_process_group('gloo', init_method='file://' + os.path.join(outdir, 'rdv2'), rank=rank, world_size=world)
    ldist.attach_rccl(_FakeHandle(log), rank, world)            # a NEW default group: the remembered id belongs to a dead one
    dist.barrier()
    dist.destroy_process_group()
    np.save(os.path.join(outdir, 'ids%d.npy' % rank), np.array([l[0] for l in log]))


def test_communicator_id_cache_is_agreed_collectively(tmp_path):
    """lsqfit_amd.dist.attach_rccl: reuse of a remembered communicator id is decided by ALL ranks together, entries are keyed by
    the group's ranks and die with the ProcessGroup object (round-5 advisor: id(group) reuse, stale entries after
    destroy_process_group, a rank with a cache hit skipping the broadcast the others wait in)."""
    import torch.multiprocessing as mp
    mp.spawn(_id_worker, args=(2, str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(tmp_path / 'ids0.npy'), np.load(tmp_path / 'ids1.npy')
    assert np.array_equal(a, b)                 # both ranks joined the same communicator every time (ids are made by rank 0)
    assert a[0] == a[1]                         # second problem: shared
    assert a[2] != a[1]                         # one rank forgot: a fresh id for everybody, no deadlock
    assert a[3] == a[4] and a[3] not in (a[0], a[2])   # the subgroup's own
    assert a[5] not in a[:5]                    # after re-initialising torch.distributed: never the dead group's id
