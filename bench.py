#!/usr/bin/env python3
"""fp64 LM steps/sec at (N_data, N_param) = (65536, 4096) on N MI355X GPUs.

One "step" = one accepted Levenberg-Marquardt iteration of the device path
(>= 1 damped Cholesky solve + >= 1 trial residual evaluation + exactly one
Jacobian assembly + one J^T J / J^T f formation + one all-reduce when N > 1;
SURVEY.md 8d), on the synthetic correlated-Gaussian workload C4: cosmix model,
256-row covariance blocks (fake_fitargs recipe), dense correlated 4096x4096
prior.  Inputs are resident in HBM before the timed region; the data rows are
sharded over the ranks (total work fixed -> "strong" scaling).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8 ...
    python bench.py --workload c5            # BASELINE config 5 on its own (128 lockstep fits of 4096 x 512)

Rank 0 prints ONE JSON line (contract in the task statement) with two extra
objects: "roofline" for the dominant kernel (the fp64-MFMA SYRK J^T J, timed
with HIP events on the stream it runs on) and "cpu_baseline" (the numpy oracle
timed on this box's host cores on a bounded sample; a baseline, not a target).
On one GPU the headline run (workload c4) also measures the other BASELINE.json
configurations -- c2, c3, c5 and the shape every rank of an 8-GPU run sees
(shard8192) -- each with its step time, its dominant kernel's roofline and a
chi2 match against the CPU port: config.other_workloads (SURVEY.md 8d "reported
per config"; --no-others skips them).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP64_MFMA_TFLOPS = 78.6   # v_mfma_f64_16x16x4_f64: 128 FLOP/clk/CU x 256 CU x 2.4 GHz
HBM_PEAK = 8.0e12              # bytes/s (MI355X_MICROARCH.md)

WORKLOADS = {
    # name: (N_data, N_param, block, dense prior, seed)   -- BASELINE.json configs
    'c4': (65536, 4096, 256, True, 20263),
    'c3': (8192, 1024, 8192, True, 20262),
    'c2': (4096, 256, 0, False, 20261),
    # what ONE of 8 ranks holds of c4 (32 covariance blocks = 8192 rows, all 4096 parameters, the replicated prior): the
    # per-rank step of the 8-GPU run, measured on one GPU (same inputs as `--ndata 8192`)
    'shard8192': (8192, 4096, 256, True, 20263),
    # BASELINE config 5: 128 fits of (4096, 512) in lockstep, differing in the prior width of the amplitudes
    'c5': (4096, 512, 0, False, 20264),
}
C5_FITS = 128


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', default='c4', choices=sorted(WORKLOADS))
    ap.add_argument('--ndata', type=int, default=0, help='override N_data (debug)')
    ap.add_argument('--nparam', type=int, default=0, help='override N_param (debug)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-others', action='store_true', help='headline only: skip config.other_workloads (c2, c3, c5, shard8192)')
    ap.add_argument('--two-pass', action='store_true', help='time the steps without phase timers, the phases in a second pass (always so for P <= 1024 on one rank)')
    ap.add_argument('--cpu-seconds', type=float, default=30.0)
    ap.add_argument('--whole-fit-maxit', type=int, default=200, help='iteration cap of the whole fits reported in config.whole_fit (0: skip them)')
    ap.add_argument('--whole-fit-algs', default='dogleg,ddogleg,subspace2D,lmaccel',
                    help='other trust-region methods the whole fit is repeated with (config.whole_fit.by_algorithm; empty: none)')
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline: the oracle's LM driver (oracle/lm.py, the GSL restatement) on a host-core port of the cosmix workload's
# normal equations (oracle/port.py CosmixPort).  "strong" mode of SURVEY.md 8d: every host core, blocked kernels --
#   trig       cos / sin of the N x K phase matrix in row chunks on a thread pool (numpy releases the GIL inside its loops),
#              written straight into the preallocated Jacobian (no hstack, no 2 GB temporaries);
#   whiten     J_b = W_b J_b per covariance block (blocks up to 1024 rows inside the same pool task, one BLAS thread each;
#              larger ones as one multi-threaded GEMM);
#   syrk       J^T J with BLAS dsyrk (the triangle only: half the flops of J.T @ J), mirrored once;
#   cholesky   the damped factorisations + solves of the trial steps (LAPACK potrf / potrs, all cores).
# A baseline, not the target (the GPU / CPU ratio says nothing about kernel quality; roofline.frac does).
def cpu_baseline(d, budget_s, maxit=8, faithful=True):
    """Oracle (numpy / BLAS 'port') LM steps/s on the same inputs.  Bounded: the driver stops after the first LM iteration
    that ends beyond the time budget (at least one iteration)."""
    from oracle import lm as olm
    from oracle.port import CosmixPort
    port = CosmixPort(d)

    class TimedLin(olm._NormalLin):       # the oracle's normal-equation algebra, its factorisations timed
        def step(self, mu, diag):
            t0 = time.perf_counter()
            try:
                return olm._NormalLin.step(self, mu, diag)
            finally:
                port.phases['cholesky'] += time.perf_counter() - t0

    t0 = time.perf_counter()
    res = olm.lm_normal(d['p0'], port.normal_eq, port.chi2_fn, tol=(1e-8, 1e-10, 1e-10), maxit=maxit,
                        stop=lambda: time.perf_counter() - t0 > budget_s, lin=TimedLin())
    elapsed = time.perf_counter() - t0       # (lm_normal's closing covariance -- one more factorisation -- is inside: part of a fit)
    port.close()
    steps = max(1, res.nit)
    try:
        import threadpoolctl
        blas_threads = max([p.get('num_threads', 1) for p in threadpoolctl.threadpool_info()] or [1])
    except Exception:
        blas_threads = port.cores
    phases = {k: round(v, 4) for k, v in port.phases.items()}
    phases['other'] = round(max(0.0, elapsed - sum(port.phases.values())), 4)
    out = dict(value=steps / elapsed, unit='LM steps/s', cores=int(max(blas_threads, port.workers)), kind='port',
               sample='%d full-size LM step(s) of the same workload (N=%d, P=%d): oracle LM driver on a host port of the normal equations '
                      '(threaded trig on %d workers, per-block whitening, BLAS dsyrk, LAPACK Cholesky on %d threads), whitening setup '
                      'excluded; baseline, not the target' % (steps, port.N, port.P, port.workers, blas_threads),
               phases_s=phases, seconds=round(elapsed, 3))
    if faithful:
        out['faithful_qr_1thread'] = faithful_qr_estimate(port.N + port.P, port.P)
    return out, res


def chi2_match(lib, h, d, res):
    """the parity half of the metric at the bench's own size: the device repeats, from the same start, the LM iterations the CPU
    port (the GSL restatement) just took; chi2 must agree to the north_star tolerance (outside the timed region)"""
    import ctypes as C
    from lsqfit_amd import _lib
    rc = lib.lsqamd_init(h, _lib.dptr(np.ascontiguousarray(d['p0'])))
    for _ in range(res.nit):
        if rc == 0:
            rc = lib.lsqamd_step(h, None)
    s2 = _lib.Summary()
    lib.lsqamd_finish(h, C.byref(s2))
    rel = abs(s2.chi2 - res.fnorm2) / res.fnorm2
    return {'after_lm_steps': int(res.nit), 'device_chi2': s2.chi2, 'cpu_port_chi2': float(res.fnorm2), 'rel_diff': rel,
            'ok': bool(rc == 0 and rel < 1e-6)}


def scipy_lm_sanity(d, budget_s):
    """scipy.optimize.least_squares(method='lm') (MINPACK, analytic Jacobian) on the uncorrelated problem, as an
    independent sanity number next to the port's: Jacobian evaluations per second and the chi2 it ends on."""
    import scipy.optimize
    x, ymean = d['x'], d['ymean']
    pm, perr = d['prior']
    sd = np.asarray(d['yerr'], float)
    K = pm.size // 2
    t0 = time.perf_counter()

    def f(p):
        return np.concatenate([(np.cos(np.outer(x, p[K:])) @ p[:K] - ymean) / sd, (p - pm) / perr])

    def jac(p):
        wx = np.outer(x, p[K:])
        J = np.hstack([np.cos(wx), -p[:K] * x[:, None] * np.sin(wx)]) / sd[:, None]
        return np.vstack([J, np.diag(1.0 / np.asarray(perr, float))])

    r = scipy.optimize.least_squares(f, d['p0'], jac=jac, method='lm', xtol=1e-8, ftol=1e-10, gtol=1e-10, max_nfev=200)
    dt = time.perf_counter() - t0
    return dict(value=r.njev / dt, unit='Jacobian evaluations/s', njev=int(r.njev), nfev=int(r.nfev), chi2=float(2.0 * r.cost),
                status=int(r.status), seconds=dt)


def faithful_qr_estimate(n, P):
    """SURVEY.md 8d "faithful" mode: what the reference's default solver costs per LM step -- one thread, an
    UNBLOCKED column-pivoted Householder QR of the n x P Jacobian (lm/more/qr, src/lsqfit/__init__.py:1336;
    gsl_linalg_QRPT_decomp: level-2 operations on a row-major matrix).  Timed with the oracle's C restatement of
    that algorithm (oracle/csrc/qrpt_unblocked.c, gcc -O2, no BLAS) on a bounded sample and scaled by the flop count
    2 n P^2 - 2/3 P^3 -- far too slow to run at full size inside a bench (hours); Jacobian assembly (Python-object
    AD in the reference) is not included.  (Rounds 1-2 scaled LAPACK's blocked dgeqp3, 7x kinder to the reference.)"""
    try:
        import ctypes as C
        from oracle import build_c
        lib = build_c.load()
        m, q = 3072, 1024
        A = np.ascontiguousarray(np.random.default_rng(0).standard_normal((m, q)))
        tau, work, perm = np.empty(q), np.empty(2 * q), np.empty(q, dtype=np.int64)
        dp = C.POINTER(C.c_double)
        t0 = time.perf_counter()
        lib.oracle_qrpt_unblocked(A.ctypes.data_as(dp), m, q, tau.ctypes.data_as(dp), perm.ctypes.data_as(C.POINTER(C.c_long)),
                                  work.ctypes.data_as(dp))
        dt = time.perf_counter() - t0
        rate = (2.0 * m * q * q - 2.0 / 3.0 * q ** 3) / dt
        flops = 2.0 * n * P * P - 2.0 / 3.0 * P ** 3
        return dict(value=rate / flops, unit='LM steps/s', cores=1, extrapolated=True, kind='port',
                    sample='unblocked pivoted Householder QR (C restatement of gsl_linalg_QRPT_decomp) of a %d x %d sample on '
                           'one thread: %.2f GFLOP/s in %.1f s, scaled to the %d x %d Jacobian (%.2e flop per step); QR only'
                           % (m, q, rate / 1e9, dt, n, P, flops))
    except Exception as e:
        return dict(value=None, unit='LM steps/s', cores=1, extrapolated=True, sample='failed: %r' % (e,))


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per
    GPU, torch.distributed.run on 127.0.0.1) BEFORE anything in this process touches the GPU, hand
    their output through and exit with their code."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()          # counting devices does not initialise the GPU
    if have < args.gpus and os.environ.get('LSQAMD_DIST_BACKEND', 'nccl') == 'nccl':
        raise SystemExit('bench.py: --gpus %d but only %d GPU(s) are visible' % (args.gpus, have))
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    # a wall-clock limit on the children: a collective that never completes (a rank that died, a
    # fabric problem) must end as a non-zero exit, never as a bench that waits forever.  The ranks run
    # in a process group of their own so that the whole tree can be ended; this process never
    # re-execs (it has not touched the GPU, but the children have).
    import signal
    limit = float(os.environ.get('LSQAMD_BENCH_TIMEOUT_S', '1500'))
    child = subprocess.Popen(cmd, start_new_session=True)
    try:
        rc = child.wait(timeout=limit)
    except subprocess.TimeoutExpired:
        sys.stderr.write('bench.py: the %d ranks did not finish within %.0f s (LSQAMD_BENCH_TIMEOUT_S): ending them\n'
                         % (args.gpus, limit))
        for sig in (signal.SIGTERM, signal.SIGKILL):
            try:
                os.killpg(child.pid, sig)
            except ProcessLookupError:
                break
            try:
                child.wait(timeout=10)
                break
            except subprocess.TimeoutExpired:
                continue
        raise SystemExit(124)
    raise SystemExit(rc)


def workload_label(name, N, P, block, dense_prior, world):
    return '%s: cosmix N_data=%d N_param=%d, %s data covariance, %s prior, rows sharded over %d GPU%s' % (
        name, N, P, ('%d-row block-diagonal' % block) if block > 1 else 'diagonal',
        'dense correlated' if dense_prior else 'diagonal', world, '' if world == 1 else 's')


def lm_workload(name, args, env, steps, warmup, headline):
    """K accepted LM steps of one single-fit workload on this rank's rows.  -> (line, ctx): `line` is the bench line's body
    (headline) or the compact entry of config.other_workloads; ctx keeps the problem for the CPU-baseline / chi2 leg."""
    import ctypes as C
    import lsqfit_amd
    from lsqfit_amd import _lib, synth
    from lsqfit_amd.dist import sharded_problem
    torch, dist, rank, world = env['torch'], env['dist'], env['rank'], env['world']
    N, P, block, dense_prior, seed = WORKLOADS[name]
    if headline and args.ndata:
        N = args.ndata
    if headline and args.nparam:
        P = args.nparam
    block = min(block, N)
    t0 = time.perf_counter()
    d = synth.make_cosmix(N=N, P=P, seed=seed, block=block, prior_corr=dense_prior)
    t_generate = time.perf_counter() - t0     # the synthetic inputs (not part of the product's set-up)
    t0 = time.perf_counter()
    wh = lsqfit_amd.Whitening(d['ymean'], d['yerr'], *d['prior'])
    # N > 1: the sums run inside the library (RCCL reduce-scatter + all-gather on the handle's
    # stream).  LSQAMD_COLLECTIVE=hook selects the torch.distributed hook instead; without the
    # variable a failed communicator set-up ends the run (never a silent change of transport).
    want = os.environ.get('LSQAMD_COLLECTIVE') or None
    if world > 1 and want is None and os.environ.get('LSQAMD_DIST_BACKEND', 'nccl') != 'nccl':
        want = 'hook'      # (LSQAMD_COLLECTIVE=rccl + LSQAMD_RCCL_PATH: the library's collective over a named RCCL build)
    note = None
    try:
        pr = sharded_problem(d['model'], d['x'], wh, rank, world, collective=want)
        ok = 1
    except RuntimeError as e:
        if want is not None or world == 1:
            raise
        ok, note = 0, repr(e)
    if world > 1 and want is None:
        flag = torch.tensor([ok], dtype=torch.int32, device='cuda')
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            # the measured configuration is the collective INSIDE the library; a run that cannot set it
            # up fails loudly (LSQAMD_COLLECTIVE=hook asks for the torch.distributed hook explicitly)
            if ok:
                pr.close()
            raise SystemExit('bench.py: rank %d: the library communicator (RCCL) could not be set up on every rank: %s'
                             % (rank, note or 'another rank failed'))
    if world == 1 and os.environ.get('LSQAMD_BENCH_SELF_COMM'):
        # developer switch: a ONE-rank RCCL communicator, so that the exchange code (comm.hip, the grouped exchange of
        # LSQAMD_EXCHANGE_GROUPS) runs -- and its launch overhead can be measured -- on a one-GPU box
        pr.comm_init(pr.comm_unique_id(), 0, 1)
        pr.collective = 'rccl'
    collective = {None: 'none (one rank)', 'rccl': 'RCCL reduce-scatter + all-gather inside the library, on the step\'s stream',
                  'hook': 'torch.distributed all_reduce through the C-ABI hook'}[pr.collective]
    groups = int(os.environ.get('LSQAMD_EXCHANGE_GROUPS', '1') or 1)
    if pr.collective == 'rccl' and groups > 1:
        collective = ('RCCL reduce-scatter + all-gather inside the library, in %d groups of J^T J tile rows on the handle\'s exchange '
                      'stream, overlapped with the next group\'s product (LSQAMD_EXCHANGE_GROUPS)' % groups)
    pr.set_options((1e-8, 1e-10, 1e-10), 1000)
    t_setup = time.perf_counter() - t0
    lib, h = pr.lib, pr.h

    rng = np.random.Generator(np.random.PCG64(seed + 1))
    ps = np.concatenate([np.full(P // 2, 0.5), np.full(P // 2, 0.1)])

    def fresh_start():
        # every restart draws the same start on all ranks (same seed stream)
        return np.ascontiguousarray(d['p0'] + 0.3 * ps * rng.standard_normal(P))

    state = dict(reinits=0, converged=True, trials0=0)

    def one_step():
        if state['converged']:
            p0 = fresh_start()
            rc = lib.lsqamd_init(h, _lib.dptr(p0))
            pr._raise_reduce()
            if rc != 0:
                raise RuntimeError('init failed: %s' % lib.lsqamd_last_error(h))
            state['reinits'] += 1
            state['converged'] = False
        info = C.c_int32(0)
        rc = lib.lsqamd_step(h, C.byref(info))
        pr._raise_reduce()
        if rc < 0:
            raise RuntimeError('step failed: %s' % lib.lsqamd_last_error(h))
        if rc != 0 or info.value != 0:
            state['converged'] = True

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        one_step()
    # Phase timers (HIP events around every phase, on the handle's stream) ride along in the timed region at the
    # named shape (0.1 % of a 20 ms step).  A SMALL problem's step is a captured graph (P <= 1024, one rank) that the
    # timers would force back to eager launches with ~18 event records per step -- 0.30 instead of 0.19 ms at
    # (4096, 256): there the timed region runs uninstrumented and the phases are timed in a second pass of the same
    # number of steps right after it (said so in roofline.timing_pass).
    two_pass = (P <= 1024 and world == 1) or args.two_pass
    if not two_pass:
        pr.timing(True)
        pr.timing_reset()
    # the headline: exactly K steps, once (the contract).  A companion with a sub-millisecond step is timed three times over
    # K steps and the MEDIAN region counts (all three are in the entry): on the pool's shared hosts a 40 ms region catches a
    # one-off stall of 10-70 ms (another tenant's CPU burst preempting the polling thread, a deferred free in the runtime)
    # every few runs -- tools/dbg_c2_after_c4.py shows single lsqamd_step calls of 14 ms among 0.16 ms ones
    regions = []
    for _ in range(1 if (headline or not two_pass) else 3):
        state['reinits'] = 0
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            one_step()
        barrier()
        regions.append((time.perf_counter() - t0, state['reinits']))
    elapsed, reinits_timed = sorted(regions)[len(regions) // 2]
    if two_pass:
        pr.timing(True)
        pr.timing_reset()
        for _ in range(steps):
            one_step()
        torch.cuda.synchronize()
    state['reinits'] = reinits_timed
    tm = pr.timings()
    pr.timing(False)
    s = _lib.Summary()
    lib.lsqamd_finish(h, C.byref(s))
    whole_fit = whole_fits(args, env, pr, d, wh, P) if headline and args.whole_fit_maxit > 0 else None
    pr.set_options((1e-8, 1e-10, 1e-10), 1000)
    reduce_ms = [tm['reduce'][0] / max(1, tm['reduce'][1])]
    step_ms = [1e3 * elapsed / steps]
    # the exposed share of the exchange, MEASURED: time the step's stream spent waiting for the collective over the time the
    # collective took on the stream it ran on (HIP events; 1 when it runs on the step's own stream, < 1 when the grouped
    # exchange overlaps it with the next group's product)
    coll_ms, wait_ms = tm['exch_coll'][0], tm['exch_wait'][0]
    exposed = [(wait_ms / coll_ms) if coll_ms > 0 else None]
    coll_per_step = [coll_ms / max(1, tm['exch_wait'][1]) if coll_ms > 0 else None]
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        # per-rank view of the exchange: average duration of the `reduce` phase (HIP events around the
        # collective on the step's stream: includes waiting for the slowest rank) and the rank's own step time
        mine = torch.tensor([reduce_ms[0], step_ms[0], exposed[0] if exposed[0] is not None else -1.0,
                             coll_per_step[0] if coll_per_step[0] is not None else -1.0], dtype=torch.float64, device='cuda')
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        reduce_ms = [float(t[0].item()) for t in every]
        step_ms = [float(t[1].item()) for t in every]
        exposed = [float(t[2].item()) if float(t[2].item()) >= 0 else None for t in every]
        coll_per_step = [float(t[3].item()) if float(t[3].item()) >= 0 else None for t in every]
        elapsed = float(tt.item())
    ctx = dict(d=d, wh=wh, pr=pr, lib=lib, h=h, N=N, P=P, block=block)
    if rank != 0:
        return None, ctx

    n_local = pr.N
    B = block if block > 1 else 0
    spl = (lib.lsqamd_debug_flags(h) >> 8) & 0xffffff

    def per(ms_cnt):
        return (ms_cnt[0] / ms_cnt[1] * 1e-3) if ms_cnt[1] else None
    # both fractions per kernel and the blended per-step fraction.  Algorithmic counts per launch
    #  over the HIP-event average of the phase that contains the kernel; MFMA phases against the
    # fp64-MFMA peak, streaming phases against HBM (8 TB/s, MI355X_MICROARCH.md)
    pk = {}
    t = per(tm['syrk'])
    syrk_flops = float(n_local) * P * (P + 1)          # algorithmic, upper triangle, 2 flop/MAC
    if t:
        pk['J^T J product'] = {'kernel': 'gemm_tn_f64_interior_kernel<false, true> (J^T J; the name in profiles/*kernel_stats*.csv)',
                               'bound': 'mfma', 'flops': syrk_flops, 'ms': t * 1e3, 'frac': syrk_flops / t / 1e12 / PEAK_FP64_MFMA_TFLOPS}
    t = per(tm['whiten'])
    if t and B:
        fl = float(B) * n_local * (P + 1)
        pk['whitening (J_b = W_b [df/dp] per block)'] = {
            'kernel': 'whiten_synth_kernel<1, 64, false>' if B < 1024 else 'gemm_tn_f64_interior_kernel<true, false> (one triangular block)',
            'bound': 'mfma', 'flops': fl, 'ms': t * 1e3, 'frac': fl / t / 1e12 / PEAK_FP64_MFMA_TFLOPS}
    t = per(tm['cholesky'])
    if t:
        fl = P ** 3 / 3.0
        pk['potrf_upper (damped normal equations)'] = {
            'kernel': 'trail_potf2_kernel + potf2_v4_kernel + gemm_tn_f64_panel_oneshot_kernel (latency-bound pivot chain, DESIGN.md 4.2)',
            'bound': 'mfma', 'flops': fl, 'ms': t * 1e3, 'frac': fl / t / 1e12 / PEAK_FP64_MFMA_TFLOPS}
    t = per(tm['residual'])
    if t:
        by = 8.0 * n_local * (3 + (1 if B else 0))      # x, ymean, weight in; r out (+ the raw vector of block rows)
        pk['trial residual (model + whitening + |f|^2)'] = {'bound': 'hbm', 'bytes': by, 'ms': t * 1e3, 'frac': by / t / HBM_PEAK,
                                                            'note': 'transcendental-bound at this shape (N K sincos), not a streaming kernel'}
    t = per(tm['grad'])
    if t:
        by = 8.0 * (spl * P * (P + 128.0) / 2 + P * P / 2)   # upper tiles of the split-K slabs in, packed tiles out
        pk['finalize_pack + prior (slab sum, J^T f)'] = {'bound': 'hbm', 'bytes': by, 'ms': t * 1e3, 'frac': by / t / HBM_PEAK}
    t = per(tm['jacobian'])
    if t and not B:
        by = 8.0 * n_local * (P + 1)
        pk['Jacobian rows'] = {'bound': 'hbm', 'bytes': by, 'ms': t * 1e3, 'frac': by / t / HBM_PEAK}
    step_flops = syrk_flops + (float(B) * n_local * (P + 1) if B else 0.0) + P ** 3 / 3.0 + 2.0 * n_local * P + 2.0 * P * P
    blended = {'flops_per_step': step_flops, 'ms_per_step': 1e3 * elapsed / steps,
               'achieved_TFLOPs': step_flops / (elapsed / steps) / 1e12,
               'frac': step_flops / (elapsed / steps) / 1e12 / PEAK_FP64_MFMA_TFLOPS,
               'note': 'algorithmic flops of one accepted step on this rank (SYRK + whitening + Cholesky + '
                       'J^T f + solve; rejected trials and model evaluation not counted) over the wall-clock step'}
    timing_pass = ('a second, instrumented pass of %d steps right after the timed region (the timed region '
                   'itself replays captured graphs without event records)' % steps) if two_pass else 'HIP events inside the timed region'
    phases = {k: (v[0] / v[1] if v[1] else None) for k, v in tm.items()}
    calls = {k: v[1] for k, v in tm.items()}
    exch = {'exposed_share_per_rank': exposed, 'collective_ms_per_exchange_per_rank': coll_per_step,
            'groups': groups if pr.collective == 'rccl' else None,
            'how': 'HIP events: (time the step\'s stream waits for the collective) / (time the collective takes on the stream it runs on)'} \
        if (world > 1 or pr.collective == 'rccl') else None
    if not headline:
        # the phase that takes most of the step names the dominant kernel of this workload
        dom = max(pk.items(), key=lambda kv: kv[1]['ms'] * calls.get({'J^T J product': 'syrk', 'whitening (J_b = W_b [df/dp] per block)': 'whiten',
                                                                        'potrf_upper (damped normal equations)': 'cholesky',
                                                                        'trial residual (model + whitening + |f|^2)': 'residual',
                                                                        'finalize_pack + prior (slab sum, J^T f)': 'grad', 'Jacobian rows': 'jacobian'}[kv[0]], 0))
        k, v = dom
        work = v.get('flops', v.get('bytes'))
        unit = 'TFLOP/s' if v['bound'] == 'mfma' else 'GB/s'
        peak = PEAK_FP64_MFMA_TFLOPS if v['bound'] == 'mfma' else HBM_PEAK / 1e9
        ach = work / (v['ms'] * 1e-3) / (1e12 if v['bound'] == 'mfma' else 1e9)
        line = {'workload': workload_label(name, N, P, block, dense_prior, world), 'ms_per_step': 1e3 * elapsed / steps,
                'ms_per_step_of_every_timed_region': [1e3 * e / steps for e, _ in regions],
                'value': steps / elapsed, 'unit': 'LM steps/s', 'steps': steps, 'warmup': warmup,
                'restarts_in_timed_region': state['reinits'], 'setup_s': round(t_setup, 3), 'generate_s': round(t_generate, 3),
                'phases_ms_per_call': phases, 'phases_calls': calls,
                'roofline': {'phase': k, 'kernel': v.get('kernel', k), 'bound': v['bound'],
                             ('flops_per_launch' if v['bound'] == 'mfma' else 'bytes_per_launch'): work, 'avg_launch_ms': v['ms'],
                             'achieved': ach, 'peak': peak, 'unit': unit, 'frac': ach / peak, 'timing_pass': timing_pass,
                             'per_kernel': pk, 'blended_step': blended}}
        return line, ctx

    # HBM traffic of the dominant kernel is a PMC quantity (separate rocprofv3 --pmc pass,
    # FETCH_SIZE x2 gfx950 correction): taken from the committed profile of this workload
    traffic, traffic_src, clk = None, None, None
    try:
        pmc = [f for f in ('r06_syrk_pmc.json', 'r05_syrk_pmc.json', 'r04_syrk_pmc.json', 'r03_syrk_pmc.json', 'r02_syrk_pmc.json', 'r01_syrk_pmc.json')
               if os.path.exists(os.path.join(ROOT, 'profiles', f))][0]
        prof = json.load(open(os.path.join(ROOT, 'profiles', pmc)))
        clk = prof['summary'].get('effective_clock_GHz')
        if world == 1 and (N, P) == (65536, 4096):
            traffic = prof['summary']['hbm_read_bytes_corrected'] + prof['summary'].get('hbm_write_bytes', 0)
            traffic_src = 'profiles/%s (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE, per launch)' % pmc
    except Exception:
        pass
    syrk_ms, syrk_n = tm['syrk']
    ach = (syrk_flops / (syrk_ms / syrk_n * 1e-3) / 1e12) if syrk_n else 0.0
    out = {
        # BASELINE.json's metric string; the "chi2 match" half is config.chi2_match below
        'metric': 'fp64 LM steps/sec at (N_data,N_param)=(%d,%d); chi2 match vs GSL' % (N, P),
        'value': steps / elapsed, 'unit': 'LM steps/s', 'n_gpus': world, 'steps': steps,
        'warmup': warmup, 'ms_per_step': 1e3 * elapsed / steps, 'higher_is_better': True,
        'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': workload_label(name, N, P, block, dense_prior, world),
                   'solver': 'lm/more/cholesky', 'collective': collective, 'restarts_in_timed_region': state['reinits'],
                   'setup_s': round(t_setup, 3), 'generate_s': round(t_generate, 3),
                   # N > 1: what creating the library's communicator took on rank 0 (inside setup_s), and how many handles share it
                   'comm_init_ms': (round(pr.comm_stats()[0], 3) if pr.collective == 'rccl' else None),
                   'comm_handles': (pr.comm_stats()[1] if pr.collective == 'rccl' else None),
                   # per-rank exchange, measured (see `exchange` below): the worst rank's exposed share
                   'exchange_exposed_share': (max([e for e in exposed if e is not None], default=None) if exch else None),
                   'exchange': exch,
                   # (the timed steps are steps of fits restarted from random points 0.3 sigma off the prior mean: far
                   #  from converged -- whole_fit below is the fit they belong to)
                   'chi2_dof_last': s.chi2 / max(1, wh.nchiv - P), 'whole_fit': whole_fit},
        'phases_ms_per_call': phases,
        'phases_calls': calls,
        'per_rank': {'reduce_ms_per_call': reduce_ms, 'ms_per_step': step_ms},
        'roofline': {'bound': 'mfma', 'kernel': 'gemm_tn_f64_interior_kernel<false, true> (J^T J; the name in profiles/*kernel_stats*.csv)', 'achieved': ach,
                     'peak': PEAK_FP64_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                     'frac': ach / PEAK_FP64_MFMA_TFLOPS, 'traffic': traffic,
                     'traffic_unit': 'bytes/launch', 'traffic_source': traffic_src,
                     'algorithmic_bytes': 8.0 * n_local * P,
                     'flops_per_launch': syrk_flops,
                     'avg_launch_ms': (syrk_ms / syrk_n) if syrk_n else None,
                     'timing_pass': timing_pass,
                     # informational: the chip sustains ~2.18 GHz (not the nominal 2.4) under this
                     # kernel (GRBM_GUI_ACTIVE in the committed PMC profile); frac stays vs nominal
                     'sustained_clock_GHz': clk,
                     'frac_of_peak_at_sustained_clock':
                         (ach / (128 * 256 * clk * 1e9 / 1e12)) if clk else None,
                     'per_kernel': pk, 'blended_step': blended},
    }
    return out, ctx


def whole_fits(args, env, pr, d, wh, P):
    """ONE whole fit of the same problem, outside the timed region, from the start SURVEY.md 8d names (p0 = the prior mean):
    what the timed steps are steps OF -- iterations to convergence, trial solves taken and rejected, the criterion that
    ended it, chi2/dof at the end -- and, from the phase timers of that run, what a rejected trial costs (one damped
    factorisation + solve + one residual evaluation, no Jacobian).  Then the same fit with the other trust-region methods
    the reference offers (alg = dogleg, ddogleg, subspace2D, lmaccel; src/lsqfit/_gsl.pyx:622-635, doc/source/overview.rst:2107-2118
    "2-3x faster"): iterations, wall time, chi2/dof (must equal lm's).  Every rank runs them (the sums are collective)."""
    import ctypes as C
    from lsqfit_amd import _lib
    torch = env['torch']
    lib, h = pr.lib, pr.h
    crit = {0: 'none: iteration limit (%d)' % args.whole_fit_maxit, 1: 'xtol', 2: 'gtol', 3: 'ftol', 4: 'no progress'}

    def run(p_start, alg='lm', timers=False):
        pr.set_options((1e-8, 1e-10, 1e-10), max(1, args.whole_fit_maxit), alg=alg)
        if timers:
            pr.timing(True)
            pr.timing_reset()
        sf = _lib.Summary()
        t0 = time.perf_counter()
        rc = lib.lsqamd_run(h, _lib.dptr(np.ascontiguousarray(p_start)), C.byref(sf))
        pr._raise_reduce()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        tf = pr.timings() if timers else None
        if timers:
            pr.timing(False)
        return rc, sf, wall, tf

    rc_fit, sf, fit_s, tf = run(d['p0'], 'lm', timers=True)

    def per_call(k):
        return (tf[k][0] / tf[k][1]) if tf[k][1] else 0.0
    whole_fit = {
        'start': 'prior mean (SURVEY.md 8d)', 'alg': 'lm', 'rc': int(rc_fit), 'nit_to_convergence': int(sf.nit), 'trials_total': int(sf.ntrial),
        'rejected_trials': int(sf.ntrial - sf.nit), 'stopping_criterion': int(sf.stopping_criterion),
        'criterion': crit.get(int(sf.stopping_criterion), str(sf.stopping_criterion)),
        'chi2_dof': sf.chi2 / max(1, wh.nchiv - P), 'wall_s': fit_s, 'device_ms': sf.t_run_ms,
        'ms_per_trial': per_call('cholesky') + per_call('solve') + per_call('residual'),
        'ms_per_rejected_trial_note': 'a rejected trial costs ms_per_trial (factorisation + back substitution + residual; HIP-event '
                                      'averages over this fit); an accepted one adds the Jacobian / whitening / J^T J phases',
        'ms_per_accepted_step': per_call('jacobian') + per_call('whiten') + per_call('syrk') + per_call('grad') + per_call('reduce')
                                + per_call('cholesky') + per_call('solve') + per_call('residual'),
        'steps_per_s_to_convergence': sf.nit / fit_s if fit_s > 0 else None,
    }
    # the same fit started a hundredth of the prior's widths from the generating values: inside the region where the Gauss-Newton model holds
    # (tools/trace_cosmix.py: with data errors of 0.1 % the fit from the prior mean spends ~9 iterations in ten on damped
    # steps that take chi2 down by 15-30 % each -- SURVEY.md 8d's "5-10 steps" is what the near start needs)
    # (in units of the prior's widths: a RELATIVE perturbation grows with the frequency index -- 1e-4 (k + 1) is 0.2 at k = 2048,
    #  a phase error of 1.3 rad at x_max and outside the basin: that start does not converge in 200 iterations)
    p_near = np.ascontiguousarray(d['p_true'] + 0.01 * np.concatenate([np.full(P // 2, 0.5), np.full(P // 2, 0.1)])
                                  * np.random.default_rng(6).standard_normal(P))
    rc_near, sn, near_s, _ = run(p_near, 'lm')
    whole_fit['near_start'] = {'start': 'p_true + 0.01 sigma_prior delta', 'rc': int(rc_near), 'nit_to_convergence': int(sn.nit),
                               'trials_total': int(sn.ntrial), 'stopping_criterion': int(sn.stopping_criterion),
                               'chi2_dof': sn.chi2 / max(1, wh.nchiv - P), 'wall_s': near_s,
                               'steps_per_s_to_convergence': sn.nit / near_s if near_s > 0 else None}
    algs = [a for a in args.whole_fit_algs.split(',') if a]
    if algs:
        table = {'lm': {'from_prior_mean': {'nit': int(sf.nit), 'trials': int(sf.ntrial), 'wall_s': fit_s, 'chi2_dof': whole_fit['chi2_dof'],
                                            'criterion': whole_fit['criterion'], 'rc': int(rc_fit)},
                        'near_start': {'nit': int(sn.nit), 'trials': int(sn.ntrial), 'wall_s': near_s,
                                       'chi2_dof': sn.chi2 / max(1, wh.nchiv - P), 'criterion': crit.get(int(sn.stopping_criterion)), 'rc': int(rc_near)}}}
        for alg in algs:
            row = {}
            for label, start in (('from_prior_mean', d['p0']), ('near_start', p_near)):
                try:
                    rc, sa, wall, _ = run(start, alg)
                    c2 = sa.chi2 / max(1, wh.nchiv - P)
                    ref = table['lm'][label]['chi2_dof']
                    row[label] = {'nit': int(sa.nit), 'trials': int(sa.ntrial), 'wall_s': wall, 'chi2_dof': c2,
                                  'criterion': crit.get(int(sa.stopping_criterion)), 'rc': int(rc),
                                  'chi2_dof_rel_diff_vs_lm': abs(c2 - ref) / ref, 'wall_vs_lm': wall / table['lm'][label]['wall_s']}
                except Exception as e:       # a method that fails must not take the bench line down
                    row[label] = {'error': repr(e)}
            table[alg] = row
        whole_fit['by_algorithm'] = table
    pr.set_options((1e-8, 1e-10, 1e-10), 1000)
    return whole_fit


def c5_workload(args, env, runs=3):
    """BASELINE config 5: 128 fits of (4096, 512) that differ in the prior width of the amplitudes, in lockstep on one GPU
    (lsqamdb_*: one captured hipGraph per round of the batch).  A "step" here is one lockstep round = one trial LM step of every
    still-active fit.  Timed: `runs` whole sweeps (wall clock around lsqamdb_run, inputs resident); then one instrumented sweep
    (eager rounds, HIP events around the batched J^T J launch) for the roofline of the dominant kernel."""
    import lsqfit_amd as amd
    from lsqfit_amd import synth
    torch = env['torch']
    N, P, _, _, seed = WORKLOADS['c5']
    B = C5_FITS
    t0 = time.perf_counter()
    d = synth.make_cosmix(N=N, P=P, seed=seed, block=0, prior_corr=False)
    t_generate = time.perf_counter() - t0
    pm = np.tile(d['prior'][0], (B, 1))
    ps = np.tile(d['prior'][1], (B, 1))
    ps[:, :P // 2] = (0.1 * 10 ** (2 * np.arange(B) / (B - 1)))[:, None]
    t0 = time.perf_counter()
    bf = amd.BatchedFits(d['model'], d['x'], d['ymean'], d['yerr'], pm, ps)
    t_setup = time.perf_counter() - t0
    bf.run(covariance=False)                      # warm-up sweep (graph capture path, caches)
    torch.cuda.synchronize()
    walls, rounds, dev_ms = [], [], []
    out = None
    for _ in range(runs):
        t0 = time.perf_counter()
        out = bf.run(covariance=False)
        torch.cuda.synchronize()
        walls.append(time.perf_counter() - t0)
        rounds.append(out['rounds'])
        dev_ms.append(out['device_ms'])
    total_rounds, total_s = int(sum(rounds)), float(sum(walls))
    fit_steps = int(out['nit'].sum())
    bf.timing(True)
    ti = bf.run(covariance=False)
    tms = bf.timings()
    bf.timing(False)
    syrk_ms, syrk_n = tms['syrk']
    # algorithmic flops of ALL the J^T J launches of a sweep: every Jacobian evaluation of every fit is N P (P + 1)
    flops_total = float(ti['njev'].sum()) * N * P * (P + 1)
    ach = flops_total / (syrk_ms * 1e-3) / 1e12 if syrk_ms > 0 else 0.0
    chol_ms, chol_n = tms['cholesky']
    line = {'workload': 'c5: %d lockstep fits of cosmix N_data=%d N_param=%d, diagonal data covariance, per-fit prior widths, '
                        'hipGraph-captured rounds, 1 GPU' % (B, N, P),
            'ms_per_step': 1e3 * total_s / total_rounds, 'value': total_rounds / total_s, 'unit': 'lockstep LM rounds/s (x %d fits)' % B,
            'steps': total_rounds, 'sweeps_timed': runs, 'ms_per_sweep': 1e3 * total_s / runs, 'device_ms_per_sweep': float(np.mean(dev_ms)),
            'rounds_per_sweep': int(rounds[-1]), 'graph_rounds_per_sweep': int(out['graph_rounds']),
            'fit_steps_per_sweep': fit_steps, 'fit_steps_per_s': fit_steps * runs / total_s,
            'nit_min_max': [int(out['nit'].min()), int(out['nit'].max())], 'all_converged': bool(np.all(out['stopping_criterion'] >= 1)),
            'setup_s': round(t_setup, 3), 'generate_s': round(t_generate, 3),
            'roofline': {'phase': 'batched J^T J', 'kernel': 'gemm_tn_f64_interior_kernel<false, true, true> (batched over the fits that moved)',
                         'bound': 'mfma', 'flops_per_launch': flops_total / max(1, syrk_n), 'launches': int(syrk_n),
                         'avg_launch_ms': syrk_ms / max(1, syrk_n), 'achieved': ach, 'peak': PEAK_FP64_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': ach / PEAK_FP64_MFMA_TFLOPS,
                         'timing_pass': 'one extra sweep with eager rounds and HIP events around every batched J^T J launch (the timed sweeps replay graphs)',
                         'per_kernel': {'potrf_upper_batched': {'bound': 'mfma', 'ms': chol_ms / max(1, chol_n), 'launches': int(chol_n),
                                                                 'flops': float(ti['nfev'].sum() - B) * P ** 3 / 3.0 / max(1, chol_n),
                                                                 'frac': (float(ti['nfev'].sum() - B) * P ** 3 / 3.0) / max(chol_ms * 1e-3, 1e-30) / 1e12 / PEAK_FP64_MFMA_TFLOPS}}}}
    line['_chi2_inputs'] = (d, pm, ps, out['chi2'].copy(), out['nit'].copy())
    bf.close()
    return line


def c5_chi2(line):
    """chi2 match of config 5: the first and the last fit of the sweep against the oracle's LM driver on the host port, run to
    convergence (a CPU leg: after every GPU measurement of the run)"""
    d, pm, ps, chi2, nit = line.pop('_chi2_inputs')
    B = pm.shape[0]
    try:
        from oracle import lm as olm
        from oracle.port import CosmixPort
        cm = []
        for b in (0, B - 1):
            db = dict(d, prior=(pm[b], ps[b]), p0=np.where(pm[b] != 0.0, pm[b], pm[b] + 0.1 * ps[b]))
            port = CosmixPort(db)
            res = olm.lm_normal(db['p0'], port.normal_eq, port.chi2_fn, tol=(1e-8, 1e-10, 1e-10), maxit=200)
            port.close()
            rel = abs(chi2[b] - res.fnorm2) / res.fnorm2
            cm.append({'fit': b, 'prior_width': float(ps[b, 0]), 'device_chi2': float(chi2[b]), 'cpu_port_chi2': float(res.fnorm2),
                       'device_nit': int(nit[b]), 'cpu_port_nit': int(res.nit), 'rel_diff': rel, 'ok': bool(rel < 1e-6)})
        line['chi2_match'] = cm
    except Exception as e:
        line['chi2_match'] = {'error': repr(e)}


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit('bench.py: --gpus must be >= 1')
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        launch_ranks(args)
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    import torch
    import torch.distributed as dist
    if world > 1:
        # launched by a launcher we do not control: every rank carries its own wall-clock limit, so a
        # collective that never completes ends the run with a non-zero exit instead of hanging it
        import threading
        limit = float(os.environ.get('LSQAMD_BENCH_TIMEOUT_S', '1500'))

        def give_up():
            sys.stderr.write('bench.py: rank %d still running after %.0f s (LSQAMD_BENCH_TIMEOUT_S): exiting\n' % (rank, limit))
            sys.stderr.flush()
            os._exit(124)
        wd = threading.Timer(limit, give_up)
        wd.daemon = True
        wd.start()
    ndev = max(1, torch.cuda.device_count())
    dev_index = local_rank % ndev            # one process per GPU on a real node; wraps on smaller boxes
    torch.cuda.set_device(dev_index)
    if world > 1:
        # "nccl" IS RCCL on ROCm.  LSQAMD_DIST_BACKEND=gloo lets the N > 1 code path be smoke-tested on
        # a 1-GPU box (RCCL refuses two ranks on one device); it is never the measured configuration.
        backend = os.environ.get('LSQAMD_DIST_BACKEND', 'nccl')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', dev_index))
        else:
            dist.init_process_group(backend)
    env = dict(torch=torch, dist=dist, rank=rank, world=world)

    if args.workload == 'c5':
        # config 5 on its own: independent fits ("replicas only") -- every rank would run its own share with no
        # exchange; the line is measured on one GPU
        if world > 1:
            raise SystemExit('bench.py: --workload c5 is a one-GPU measurement (independent fits: N GPUs run N shares, no collective)')
        c5 = c5_workload(args, env, runs=max(1, args.steps // 20))
        if args.no_cpu_baseline:
            c5.pop('_chi2_inputs')
        else:
            c5_chi2(c5)
        rf = c5.pop('roofline')
        out = {'metric': 'fp64 LM steps/sec at (N_data,N_param)=(4096,512) x %d lockstep fits; chi2 match vs GSL' % C5_FITS,
               'value': c5['fit_steps_per_s'], 'unit': 'LM steps/s (accepted iterations summed over the fits)', 'n_gpus': 1,
               'steps': c5['steps'], 'warmup': 1, 'ms_per_step': c5['ms_per_step'], 'higher_is_better': True, 'scaling': 'weak',
               'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
               'config': dict(c5, step='one lockstep round of the batch (a replayed hipGraph): ms_per_step x steps = the timed sweeps'),
               'roofline': rf, 'cpu_baseline': None}
        print(json.dumps(out))
        return

    out, ctx = lm_workload(args.workload, args, env, args.steps, args.warmup, headline=True)
    # the other BASELINE configurations beside the headline (one GPU, the c4 run only): SURVEY.md 8d "reported per config".
    # EVERY GPU measurement of the run comes first, the CPU legs (baseline, chi2 matches) after all of them: the host port
    # leaves BLAS / pool threads spinning for a while, and a 0.2 ms step measured right behind it picked that up
    companions = rank == 0 and world == 1 and args.workload == 'c4' and not args.no_others and not args.ndata and not args.nparam
    others, kept = {}, {}
    t_all = time.perf_counter()
    if companions:
        for name in ('c2', 'c3', 'shard8192'):
            t0 = time.perf_counter()
            try:
                line, c = lm_workload(name, args, env, max(args.steps, 200) if name == 'c2' else min(args.steps, 20),
                                      max(args.warmup, 20) if name == 'c2' else args.warmup, headline=False)
                line['measured_in_s'] = round(time.perf_counter() - t0, 2)
                others[name], kept[name] = line, c
            except Exception as e:
                others[name] = {'error': repr(e)}
        t0 = time.perf_counter()
        try:
            others['c5'] = c5_workload(args, env)
            others['c5']['measured_in_s'] = round(time.perf_counter() - t0, 2)
        except Exception as e:
            others['c5'] = {'error': repr(e)}
    pr, d = ctx['pr'], ctx['d']
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            try:
                cb, res = cpu_baseline(d, args.cpu_seconds)
                out['cpu_baseline'] = cb
                out['config']['chi2_match'] = chi2_match(ctx['lib'], ctx['h'], d, res)
                if not ctx['block'] > 1 and np.ndim(d['prior'][1]) == 1 and ctx['N'] * ctx['P'] <= 4096 * 256:     # (the uncorrelated workload, c2)
                    try:
                        out['cpu_baseline']['scipy_least_squares_lm'] = scipy_lm_sanity(d, args.cpu_seconds)
                    except Exception as e:
                        out['cpu_baseline']['scipy_least_squares_lm'] = {'value': None, 'sample': 'failed: %r' % (e,)}
            except Exception as e:   # the baseline must never take the measurement down
                out['cpu_baseline'] = {'value': None, 'unit': 'LM steps/s', 'cores': 0, 'kind': 'port',
                                       'sample': 'failed: %r' % (e,)}
        else:
            out['cpu_baseline'] = None
    pr.close()
    if companions:
        for name, c in kept.items():
            t0 = time.perf_counter()
            if not args.no_cpu_baseline:
                try:      # a bounded leg of the CPU port (<= 2 LM iterations) that the device then repeats: chi2 must agree
                    cb, res = cpu_baseline(c['d'], 2.0, maxit=2, faithful=False)
                    others[name]['chi2_match'] = chi2_match(c['lib'], c['h'], c['d'], res)
                    others[name]['cpu_port_steps_per_s'] = cb['value']
                except Exception as e:
                    others[name]['chi2_match'] = {'error': repr(e)}
            c['pr'].close()
            others[name]['measured_in_s'] = round(others[name]['measured_in_s'] + time.perf_counter() - t0, 2)
        if '_chi2_inputs' in others.get('c5', {}):
            t0 = time.perf_counter()
            if args.no_cpu_baseline:
                others['c5'].pop('_chi2_inputs')
            else:
                c5_chi2(others['c5'])
            others['c5']['measured_in_s'] = round(others['c5']['measured_in_s'] + time.perf_counter() - t0, 2)
        out['config']['other_workloads'] = others
        out['config']['other_workloads_s'] = round(sum(v.get('measured_in_s', 0.0) for v in others.values()), 2)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
