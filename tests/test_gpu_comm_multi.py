"""-m gpu: the collective INSIDE the library (comm.hip comm_all_reduce: reduce-scatter + all-gather
over 256-byte slices with in-place offsets, tail all-reduce, the count-1 trial scalar) driven by 2
and 3 ranks.  RCCL refuses two ranks on one device and the pool's boxes have one GPU, so the seven
RCCL symbols comm.hip binds are provided by a TEST-ONLY stand-in (tests/fake_rccl.cpp, built here,
selected through LSQAMD_RCCL_PATH, host-staged through POSIX shared memory, every wait bounded):
everything on the library's side of those seven calls is the product's code, the same code an
8-GPU run executes.  Checks, as tests/test_gpu_dist2.py does for the hook: all ranks bit-identical,
equal to the unsharded device fit; both LSQAMD_COMM_ALGO forms; element counts that are not
multiples of 32 * nranks; a missing rank is an error, not a hang."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope='module')
def fake_rccl(tmp_path_factory):
    out = str(tmp_path_factory.mktemp('fake_rccl') / 'libfake_rccl.so')
    cmd = [os.environ.get('HIPCC', 'hipcc'), '-O2', '-std=c++17', '-fPIC', '-shared',     # host code only: no kernels
           os.path.join(HERE, 'fake_rccl.cpp'), '-o', out, '-lrt']
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    return out


def _problem(case):
    from lsqfit_amd import synth
    if case == 'blocks':      # P = 384: npk + P + 1 = 6 tiles * 128^2 + 385 = 98689 doubles (odd; slices of 49344 / 32896 + tails)
        return synth.make_cosmix(N=1536, P=384, seed=191, block=256, prior_corr=True)
    if case in ('small', 'small8'):       # P = 30: the packed buffer is below the 64 KiB-per-rank switch -> single all-reduce
        return synth.make_cosmix(N=1000, P=30, seed=192, block=0, prior_corr=False)
    if case == 'eight':       # 8 blocks for 8 ranks; 98689 doubles = 8 slices of 12320 (whole 256-byte lines) + a tail of 129
        return synth.make_cosmix(N=2048, P=384, seed=193, block=256, prior_corr=True)
    if case == 'c4_packed':   # the named shape's parameter count: ONE packed buffer of 528 tiles * 128^2 + 4097 doubles = 69.2 MB per exchange
        return synth.make_cosmix(N=2048, P=4096, seed=194, block=256, prior_corr=True)
    raise ValueError(case)


def _one(rank, world, outdir, case, fake, algo, missing):
    import lsqfit_amd as amd
    from lsqfit_amd.dist import shard_rows
    # algo = 'rsag' | 'allreduce' | 'rsag.gG': the last form asks for the exchange in G groups of tile rows on the handle's
    # exchange stream (LSQAMD_EXCHANGE_GROUPS, read at lsqamd_create)
    os.environ['LSQAMD_COMM_ALGO'] = algo.split('.')[0]
    os.environ['LSQAMD_EXCHANGE_GROUPS'] = algo.split('.g')[1].rstrip('s') if '.g' in algo else '1'
    os.environ['LSQAMD_EXCHANGE_MODE'] = 'split' if algo.endswith('s') and '.g' in algo else 'signal'
    tag = '%s_%s' % (case, algo)
    d = _problem(case)
    wh = amd.Whitening(d['ymean'], d['yerr'], *d['prior'])
    ranges = shard_rows(wh.n_data, [(b['row0'], b['size']) for b in wh.blocks], world)
    pr = amd.DeviceProblem(d['model'], d['x'], wh, rows=ranges[rank], adds_prior=(rank == 0))
    # the id travels through a file (any host channel will do: lsqamd_comm_unique_id's contract)
    idf = os.path.join(outdir, 'comm_id_' + tag)
    if rank == 0:
        uid = pr.comm_unique_id()
        with open(idf + '.tmp', 'wb') as fh:
            fh.write(uid)
        os.replace(idf + '.tmp', idf)
    else:
        import time
        t0 = time.time()
        while not os.path.exists(idf):
            time.sleep(0.01)
            assert time.time() - t0 < 300
        uid = open(idf, 'rb').read()
    if missing:
        # rank 1 of 2 never joins: rank 0's init must come back with an error inside the bounded wait
        try:
            pr.comm_init(uid, rank, world)
            np.savez(os.path.join(outdir, '%s_r%d.npz' % (tag, rank)), error='')
        except RuntimeError as e:
            np.savez(os.path.join(outdir, '%s_r%d.npz' % (tag, rank)), error=str(e))
        pr.close()
        return
    pr.comm_init(uid, rank, world)
    assert pr.comm_info() == (rank, world)
    pr.timing(True)
    kw = dict(maxit=2) if case == 'c4_packed' else {}
    fit = amd.nonlinear_fit(data=(d['x'], d['ymean'], d['yerr']), model=d['model'], prior=d['prior'], problem=pr, **kw)
    tm = pr.timings()
    s = fit.fitter_results.summary
    m = 65536 if case == 'small8' else 5             # 65536 sums over 8 ranks: 8 slices of exactly 8192 doubles, no tail
    pts = fit.pmean + 1e-3 * np.random.default_rng(2).standard_normal((m, d['p0'].size))
    c2 = pr.chi2_points(pts)                         # m sums through the communicator (count 5: all-reduce form)
    np.savez(os.path.join(outdir, '%s_r%d.npz' % (tag, rank)), pmean=fit.pmean, cov=fit.cov, chi2=fit.chi2, nit=fit.nit,
             logGBF=fit.logGBF, c2=c2, reduces=tm['reduce'][1], expect=s.njev + s.nfev - 1, error='')
    pr.comm_destroy()
    pr.close()


def _worker(rank, world, outdir, jobs, fake, missing):
    """One set of `world` processes runs all (case, algo) jobs of that world size, a fresh communicator each (a process costs an
    import of torch and a HIP context: on a slow box that, not the fits, is what these tests take)."""
    os.environ['LSQAMD_RCCL_PATH'] = fake
    if missing:
        os.environ['LSQAMD_FAKE_RCCL_TIMEOUT_S'] = '3'
    sys.path.insert(0, ROOT)
    import torch
    torch.cuda.set_device(0)
    for case, algo in jobs:
        _one(rank, world, outdir, case, fake, algo, missing)


# 'rsag.gG': G groups signalled out of ONE product launch (the default form); 'rsag.gGs': one product launch per group
JOBS = {2: [('blocks', 'rsag'), ('blocks', 'allreduce'), ('blocks', 'rsag.g2'), ('blocks', 'rsag.g2s')],
        3: [('blocks', 'rsag'), ('small', 'rsag'), ('blocks', 'rsag.g3'), ('small', 'rsag.g2'), ('blocks', 'rsag.g3s')],
        8: [('eight', 'rsag'), ('small8', 'rsag'), ('c4_packed', 'rsag'), ('eight', 'rsag.g2'), ('c4_packed', 'rsag.g4')]}


def _spawn(world, outdir, jobs, fake, missing=False, nstart=None):
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    procs = [ctx.Process(target=_worker, args=(r, world, outdir, jobs, fake, missing))
             for r in range(world if nstart is None else nstart)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(900)
    for p in procs:
        if p.is_alive():
            p.kill()
    return [p.exitcode for p in procs]


@pytest.fixture(scope='module')
def runs(tmp_path_factory, fake_rccl):
    done = {}

    def get(world):
        if world not in done:
            out = str(tmp_path_factory.mktemp('ranks%d' % world))
            done[world] = (out, _spawn(world, out, JOBS[world], fake_rccl))
        return done[world]
    return get


def _results(runs, world, case, algo='rsag'):
    out, codes = runs(world)
    files = [os.path.join(out, '%s_%s_r%d.npz' % (case, algo, r)) for r in range(world)]
    assert all(os.path.exists(f) for f in files), 'job (%s, %s) did not finish on every rank (exit codes %s)' % (case, algo, codes)
    return [np.load(f) for f in files]


@pytest.mark.parametrize('case,world,algo', [('blocks', 2, 'rsag'), ('blocks', 3, 'rsag'), ('blocks', 2, 'allreduce'),
                                             ('small', 3, 'rsag')])
def test_library_collective_with_several_ranks(case, world, algo, runs):
    import lsqfit_amd as amd
    res = _results(runs, world, case, algo)
    for r in res:
        assert int(r['reduces']) == int(r['expect'])       # one packed exchange per Jacobian, one scalar per trial
    for r in res[1:]:                                      # every rank received the same bytes: identical decisions
        for k in ('pmean', 'cov', 'chi2', 'nit', 'logGBF', 'c2'):
            assert np.array_equal(res[0][k], r[k]), k
    d = _problem(case)
    ref = amd.nonlinear_fit(data=(d['x'], d['ymean'], d['yerr']), model=d['model'], prior=d['prior'])
    rel = lambda a, b: np.max(np.abs(a - b)) / np.max(np.abs(b))
    assert rel(res[0]['pmean'], ref.pmean) < 1e-9
    assert rel(res[0]['cov'], ref.cov) < 1e-8
    assert abs(res[0]['chi2'] / ref.chi2 - 1) < 1e-10
    assert int(res[0]['nit']) == ref.nit
    pts = ref.pmean + 1e-3 * np.random.default_rng(2).standard_normal((5, d['p0'].size))
    assert rel(res[0]['c2'], ref.problem.chi2_points(pts)) < 1e-8


def test_rsag_and_allreduce_forms_agree_bit_for_bit(runs):
    """Both forms sum in rank order in the stand-in, so the fits must be the same bits: what differs
    is only comm.hip's slicing (offsets, tail), which this pins."""
    a = _results(runs, 2, 'blocks', 'rsag')
    b = _results(runs, 2, 'blocks', 'allreduce')
    for k in ('pmean', 'cov', 'chi2', 'nit'):
        assert np.array_equal(a[0][k], b[0][k]), k


@pytest.mark.parametrize('case,world,groups', [('blocks', 2, '2'), ('blocks', 2, '2s'), ('blocks', 3, '3'), ('blocks', 3, '3s'), ('small', 3, '2'),
                                               ('eight', 8, '2'), ('c4_packed', 8, '4')])
def test_grouped_exchange_is_bit_identical(case, world, groups, runs):
    """LSQAMD_EXCHANGE_GROUPS = G: the J^T J tiles in G groups of tile rows, group g's packed tiles summed over the ranks
    on the handle's exchange stream while the rest of the product is computed, the step's stream waiting for the last
    event before the damped matrix is built (api.hip eval_normal_dev).  Two forms: ONE product launch whose workgroups count
    a group's finished entries, a one-wave kernel on the exchange stream waiting for the count ('G', the default); one
    product launch per group ('Gs', LSQAMD_EXCHANGE_MODE=split).  Every element is still the same sum of the same split-K
    slabs and the stand-in sums ranks in rank order whatever the slice: fits must be the same BITS as with one exchange, on
    every rank; one exposed wait per Jacobian in the `reduce` timer.  ('small': P = 30 is one tile -- the request is ignored.)"""
    a = _results(runs, world, case, 'rsag')
    b = _results(runs, world, case, 'rsag.g%s' % groups)
    for r in b:
        assert int(r['reduces']) == int(r['expect'])
        for k in ('pmean', 'cov', 'chi2', 'nit', 'logGBF', 'c2'):
            assert np.array_equal(a[0][k], r[k]), k


def test_missing_rank_is_an_error_not_a_hang(tmp_path, fake_rccl):
    codes = _spawn(2, str(tmp_path), [('small', 'rsag')], fake_rccl, missing=True, nstart=1)
    assert codes == [0]
    res = [np.load(os.path.join(str(tmp_path), 'small_rsag_r0.npz'))]
    assert 'EREDUCE' in str(res[0]['error']) or 'missing' in str(res[0]['error'])


@pytest.mark.parametrize('case', ['eight', 'small8'])
def test_eight_ranks(case, runs):
    """The node the north star names has 8 GPUs: 8 ranks through comm.hip's slicing -- 98689 doubles (8 slices of 12320 + a tail of
    129; `eight`), 65536 many-point sums (8 slices of exactly 8192, no tail; `small8`), the count-1 trial scalar -- on the stand-in.
    Every rank bit-identical, equal to the unsharded fit.  (Says nothing about xGMI: no scaling number follows from this.)"""
    import lsqfit_amd as amd
    res = _results(runs, 8, case)
    assert len(res) == 8
    for r in res:
        assert int(r['reduces']) == int(r['expect'])
    for r in res[1:]:
        for k in ('pmean', 'cov', 'chi2', 'nit', 'logGBF', 'c2'):
            assert np.array_equal(res[0][k], r[k]), k
    d = _problem(case)
    ref = amd.nonlinear_fit(data=(d['x'], d['ymean'], d['yerr']), model=d['model'], prior=d['prior'])
    rel = lambda a, b: np.max(np.abs(a - b)) / np.max(np.abs(b))
    assert rel(res[0]['pmean'], ref.pmean) < 1e-9 and rel(res[0]['cov'], ref.cov) < 1e-8
    assert abs(res[0]['chi2'] / ref.chi2 - 1) < 1e-10 and int(res[0]['nit']) == ref.nit
    m = res[0]['c2'].size
    pts = ref.pmean + 1e-3 * np.random.default_rng(2).standard_normal((m, d['p0'].size))
    assert rel(res[0]['c2'], ref.problem.chi2_points(pts)) < 1e-8


def test_named_parameter_count_packed_exchange_over_eight_ranks(runs):
    """P = 4096 as in the headline configuration: the 69.2 MB packed [J^T J | J^T f | chi2] buffer goes through the reduce-scatter +
    all-gather of 8 ranks once per Jacobian (two LM iterations; 256 rows = one covariance block per rank)."""
    import lsqfit_amd as amd
    res = _results(runs, 8, 'c4_packed')
    for r in res[1:]:
        for k in ('pmean', 'cov', 'chi2', 'nit'):
            assert np.array_equal(res[0][k], r[k]), k
    d = _problem('c4_packed')
    ref = amd.nonlinear_fit(data=(d['x'], d['ymean'], d['yerr']), model=d['model'], prior=d['prior'], maxit=2)
    rel = lambda a, b: np.max(np.abs(a - b)) / np.max(np.abs(b))
    assert int(res[0]['nit']) == ref.nit == 2
    assert rel(res[0]['pmean'], ref.pmean) < 1e-9 and abs(res[0]['chi2'] / ref.chi2 - 1) < 1e-9
    # (2048 rows for 4096 parameters, two iterations from the start: A = J^T J + prior has cond ~ 1e6 and its entries carry the 1e-11
    #  of differently ordered sums at chi2 ~ 4e9 -- the inverse agrees to cond x that)
    assert rel(res[0]['cov'], ref.cov) < 1e-3
