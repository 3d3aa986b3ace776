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

Rank 0 prints ONE JSON line (contract in the task statement) with two extra
objects: "roofline" for the dominant kernel (the fp64-MFMA SYRK J^T J, timed
with HIP events on the stream it runs on) and "cpu_baseline" (the numpy oracle
timed on this box's host cores on a bounded sample; a baseline, not a target).
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

WORKLOADS = {
    # name: (N_data, N_param, block, dense prior, seed)   -- BASELINE.json configs
    'c4': (65536, 4096, 256, True, 20263),
    'c3': (8192, 1024, 8192, True, 20262),
    'c2': (4096, 256, 0, False, 20261),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', default='c4', choices=sorted(WORKLOADS))
    ap.add_argument('--ndata', type=int, default=0, help='override N_data (debug)')
    ap.add_argument('--nparam', type=int, default=0, help='override N_param (debug)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--two-pass', action='store_true', help='time the steps without phase timers, the phases in a second pass (always so for P <= 1024 on one rank)')
    ap.add_argument('--cpu-seconds', type=float, default=40.0)
    ap.add_argument('--whole-fit-maxit', type=int, default=200, help='iteration cap of the one whole fit reported in config.whole_fit (0: skip it)')
    return ap.parse_args()


def cpu_baseline(d, wh_blocks_from, budget_s):
    """Oracle (numpy, 'port') LM steps/s on the same inputs: same normal-equation
    algorithm (whitening GEMMs, J^T J, damped Cholesky), all host cores via the
    BLAS numpy links.  Bounded: stops after the first LM step that crosses the
    time budget (at least one step)."""
    import scipy.linalg as sla
    from oracle import lm as olm
    x, ymean = d['x'], d['ymean']
    pm, perr = d['prior']
    P = pm.size
    K = P // 2
    sd = np.asarray(d['yerr']['sdev'] if isinstance(d['yerr'], dict) else d['yerr'], float)
    blocks = d['yerr']['blocks'] if isinstance(d['yerr'], dict) else []
    # whitening setup (untimed, as on the GPU side): W_b = inv(chol(C_b))
    Ws = []
    for r0, cov in blocks:
        L = sla.cholesky(cov, lower=True)
        Ws.append((r0, sla.solve_triangular(L, np.eye(cov.shape[0]), lower=True)))
    inblk = np.zeros(ymean.size, bool)
    for r0, W in Ws:
        inblk[r0:r0 + W.shape[0]] = True
    wdiag = np.where(inblk, 1.0, 1.0 / sd)
    prec = np.linalg.inv(perr) if np.ndim(perr) == 2 else np.diag(1.0 / np.asarray(perr) ** 2)

    def resid_raw(p):
        return np.cos(np.outer(x, p[K:])) @ p[:K] - ymean

    def whiten(v):
        out = v * (wdiag[:, None] if v.ndim == 2 else wdiag)
        for r0, W in Ws:
            out[r0:r0 + W.shape[0]] = W @ v[r0:r0 + W.shape[0]]
        return out

    def chi2_fn(p):
        r = whiten(resid_raw(p))
        dp = p - pm
        return float(r @ r + dp @ prec @ dp)

    def normal_eq(p):
        wx = np.outer(x, p[K:])
        c, s = np.cos(wx), np.sin(wx)
        J = whiten(np.hstack([c, -p[:K] * x[:, None] * s]))
        r = whiten(c @ p[:K] - ymean)
        dp = p - pm
        return J.T @ J + prec, J.T @ r + prec @ dp, float(r @ r + dp @ prec @ dp)

    # ONE time-boxed run of the driver: it stops after the first LM iteration that ends beyond
    # the budget (at the named shape an iteration of the port costs ~17 s: three iterations with
    # the default budget), and the device repeats exactly these iterations for chi2_match
    t0 = time.perf_counter()
    res = olm.lm_normal(d['p0'], normal_eq, chi2_fn, tol=(1e-8, 1e-10, 1e-10), maxit=8,
                        stop=lambda: time.perf_counter() - t0 > budget_s)
    elapsed = time.perf_counter() - t0
    steps = res.nit
    try:
        import threadpoolctl
        cores = max([p.get('num_threads', 1) for p in threadpoolctl.threadpool_info()] or [1])
    except Exception:
        cores = os.cpu_count() or 1
    return dict(value=steps / elapsed, unit='LM steps/s', cores=int(cores), kind='port',
                sample='%d full-size LM step(s) of the same workload (N=%d, P=%d), numpy/OpenBLAS oracle, '
                       'whitening setup excluded' % (steps, ymean.size, P),
                faithful_qr_1thread=faithful_qr_estimate(ymean.size + P, P)), res


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
    import lsqfit_amd
    from lsqfit_amd import _lib, synth
    from lsqfit_amd.dist import sharded_problem
    import ctypes as C

    N, P, block, dense_prior, seed = WORKLOADS[args.workload]
    if args.ndata:
        N = args.ndata
    if args.nparam:
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
    collective = {None: 'none (one rank)', 'rccl': 'RCCL reduce-scatter + all-gather inside the library, on the step\'s stream',
                  'hook': 'torch.distributed all_reduce through the C-ABI hook'}[pr.collective]
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

    for _ in range(args.warmup):
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
    state['reinits'] = 0
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    barrier()
    elapsed = time.perf_counter() - t0
    reinits_timed = state['reinits']
    if two_pass:
        pr.timing(True)
        pr.timing_reset()
        for _ in range(args.steps):
            one_step()
        torch.cuda.synchronize()
    state['reinits'] = reinits_timed
    tm = pr.timings()
    pr.timing(False)
    s = _lib.Summary()
    lib.lsqamd_finish(h, C.byref(s))
    # ONE whole fit of the same problem, outside the timed region, from the start SURVEY.md 8d names (p0 = the prior mean):
    # what the steps above are steps OF -- iterations to convergence, trial solves taken and rejected, the criterion that
    # ended it, chi2/dof at the end -- and, from the phase timers of that run, what a rejected trial costs (one damped
    # factorisation + solve + one residual evaluation, no Jacobian).  Every rank runs it (the sums are collective).
    whole_fit = None
    if args.whole_fit_maxit > 0:
        pr.set_options((1e-8, 1e-10, 1e-10), max(1, args.whole_fit_maxit))
        pr.timing(True)
        pr.timing_reset()
        sf = _lib.Summary()
        t0 = time.perf_counter()
        rc_fit = lib.lsqamd_run(h, _lib.dptr(np.ascontiguousarray(d['p0'])), C.byref(sf))
        pr._raise_reduce()
        torch.cuda.synchronize()
        fit_s = time.perf_counter() - t0
        tf = pr.timings()
        pr.timing(False)

        def per_call(k):
            return (tf[k][0] / tf[k][1]) if tf[k][1] else 0.0
        whole_fit = {
            'start': 'prior mean (SURVEY.md 8d)', 'rc': int(rc_fit), 'nit_to_convergence': int(sf.nit), 'trials_total': int(sf.ntrial),
            'rejected_trials': int(sf.ntrial - sf.nit), 'stopping_criterion': int(sf.stopping_criterion),
            'criterion': {0: 'none: iteration limit (%d)' % args.whole_fit_maxit, 1: 'xtol', 2: 'gtol', 3: 'ftol'}.get(int(sf.stopping_criterion), str(sf.stopping_criterion)),
            'chi2_dof': sf.chi2 / max(1, wh.nchiv - P), 'wall_s': fit_s, 'device_ms': sf.t_run_ms,
            'ms_per_trial': per_call('cholesky') + per_call('solve') + per_call('residual'),
            'ms_per_rejected_trial_note': 'a rejected trial costs ms_per_trial (factorisation + back substitution + residual; HIP-event '
                                          'averages over this fit); an accepted one adds the Jacobian / whitening / J^T J phases',
            'ms_per_accepted_step': per_call('jacobian') + per_call('whiten') + per_call('syrk') + per_call('grad') + per_call('reduce')
                                    + per_call('cholesky') + per_call('solve') + per_call('residual'),
        }
        # the same fit started a hundredth of the prior's widths from the generating values: inside the region where the Gauss-Newton model holds
        # (tools/trace_cosmix.py: with data errors of 0.1 % the fit from the prior mean spends ~9 iterations in ten on damped
        # steps that take chi2 down by 15-30 % each -- SURVEY.md 8d's "5-10 steps" is what the near start needs)
        # (in units of the prior's widths: a RELATIVE perturbation grows with the frequency index -- 1e-4 (k + 1) is 0.2 at k = 2048,
        #  a phase error of 1.3 rad at x_max and outside the basin: that start does not converge in 200 iterations)
        p_near = np.ascontiguousarray(d['p_true'] + 0.01 * np.concatenate([np.full(P // 2, 0.5), np.full(P // 2, 0.1)])
                                      * np.random.default_rng(6).standard_normal(P))
        sn = _lib.Summary()
        t0 = time.perf_counter()
        rc_near = lib.lsqamd_run(h, _lib.dptr(p_near), C.byref(sn))
        pr._raise_reduce()
        torch.cuda.synchronize()
        near_s = time.perf_counter() - t0
        whole_fit['steps_per_s_to_convergence'] = sf.nit / fit_s if fit_s > 0 else None
        whole_fit['near_start'] = {'start': 'p_true + 0.01 sigma_prior delta', 'rc': int(rc_near), 'nit_to_convergence': int(sn.nit),
                                   'trials_total': int(sn.ntrial), 'stopping_criterion': int(sn.stopping_criterion),
                                   'chi2_dof': sn.chi2 / max(1, wh.nchiv - P), 'wall_s': near_s,
                                   'steps_per_s_to_convergence': sn.nit / near_s if near_s > 0 else None}
    pr.set_options((1e-8, 1e-10, 1e-10), 1000)
    reduce_ms = [tm['reduce'][0] / max(1, tm['reduce'][1])]
    step_ms = [1e3 * elapsed / args.steps]
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        # per-rank view of the exchange: average duration of the `reduce` phase (HIP events around the
        # collective on the step's stream: includes waiting for the slowest rank) and the rank's own step time
        mine = torch.tensor([reduce_ms[0], step_ms[0]], dtype=torch.float64, device='cuda')
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        reduce_ms = [float(t[0].item()) for t in every]
        step_ms = [float(t[1].item()) for t in every]
        elapsed = float(tt.item())

    if rank == 0:
        n_local = pr.N
        # HBM traffic of the dominant kernel is a PMC quantity (separate rocprofv3 --pmc pass,
        # FETCH_SIZE x2 gfx950 correction): taken from the committed profile of this workload
        traffic, traffic_src, clk = None, None, None
        try:
            pmc = [f for f in ('r05_syrk_pmc.json', 'r04_syrk_pmc.json', 'r03_syrk_pmc.json', 'r02_syrk_pmc.json', 'r01_syrk_pmc.json') if os.path.exists(os.path.join(ROOT, 'profiles', f))][0]
            prof = json.load(open(os.path.join(ROOT, 'profiles', pmc)))
            clk = prof['summary'].get('effective_clock_GHz')
            if world == 1 and (N, P) == (65536, 4096):
                traffic = prof['summary']['hbm_read_bytes_corrected'] + prof['summary'].get('hbm_write_bytes', 0)
                traffic_src = 'profiles/%s (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE, per launch)' % pmc
        except Exception:
            pass
        syrk_ms, syrk_n = tm['syrk']
        flops = float(n_local) * P * (P + 1)          # algorithmic, upper triangle, 2 flop/MAC
        ach = (flops / (syrk_ms / syrk_n * 1e-3) / 1e12) if syrk_n else 0.0
        out = {
            # BASELINE.json's metric string; the "chi2 match" half is config.chi2_match below
            'metric': 'fp64 LM steps/sec at (N_data,N_param)=(%d,%d); chi2 match vs GSL' % (N, P),
            'value': args.steps / elapsed, 'unit': 'LM steps/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True,
            'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': '%s: cosmix N_data=%d N_param=%d, %s data covariance, %s prior, '
                                   'rows sharded over %d GPU(s)' % (
                                       args.workload, N, P,
                                       ('%d-row block-diagonal' % block) if block > 1 else 'diagonal',
                                       'dense correlated' if dense_prior else 'diagonal', world),
                       'solver': 'lm/more/cholesky', 'collective': collective, 'restarts_in_timed_region': state['reinits'],
                       'setup_s': round(t_setup, 3), 'generate_s': round(t_generate, 3),
                       # N > 1: what creating the library's communicator took on rank 0 (inside setup_s), and how many handles share it
                       'comm_init_ms': (round(pr.comm_stats()[0], 3) if pr.collective == 'rccl' else None),
                       'comm_handles': (pr.comm_stats()[1] if pr.collective == 'rccl' else None),
                       # per-rank exchange: `reduce_ms_per_call` below is the phase between HIP events around the collective on the
                       # step's stream -- ALL of it is exposed (the exchange is not overlapped with anything: DESIGN.md 6.2)
                       'exchange_exposed_share': (1.0 if world > 1 else None),
                       # (the timed steps are steps of fits restarted from random points 0.3 sigma off the prior mean: far
                       #  from converged -- whole_fit below is the fit they belong to)
                       'chi2_dof_last': s.chi2 / max(1, wh.nchiv - P), 'whole_fit': whole_fit},
            'phases_ms_per_call': {k: (v[0] / v[1] if v[1] else None) for k, v in tm.items()},
            'phases_calls': {k: v[1] for k, v in tm.items()},
            'per_rank': {'reduce_ms_per_call': reduce_ms, 'ms_per_step': step_ms},
            'roofline': {'bound': 'mfma', 'kernel': 'gemm_tn_f64_interior_kernel<false, true> (J^T J; the name in profiles/*kernel_stats*.csv)', 'achieved': ach,
                         'peak': PEAK_FP64_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': ach / PEAK_FP64_MFMA_TFLOPS, 'traffic': traffic,
                         'traffic_unit': 'bytes/launch', 'traffic_source': traffic_src,
                         'algorithmic_bytes': 8.0 * n_local * P,
                         'flops_per_launch': flops,
                         'avg_launch_ms': (syrk_ms / syrk_n) if syrk_n else None,
                         'timing_pass': ('a second, instrumented pass of %d steps right after the timed region (the timed region '
                                         'itself replays captured graphs without event records)' % args.steps) if two_pass
                                        else 'HIP events inside the timed region',
                         # informational: the chip sustains ~2.18 GHz (not the nominal 2.4) under this
                         # kernel (GRBM_GUI_ACTIVE in the committed PMC profile); frac stays vs nominal
                         'sustained_clock_GHz': clk,
                         'frac_of_peak_at_sustained_clock':
                             (ach / (128 * 256 * clk * 1e9 / 1e12)) if clk else None},
        }
        # both fractions per kernel and the blended per-step fraction.  Algorithmic counts per launch
        #  over the HIP-event average of the phase that contains the kernel; MFMA phases against the
        # fp64-MFMA peak, streaming phases against HBM (8 TB/s, MI355X_MICROARCH.md)
        HBM_PEAK = 8.0e12
        B = block if block > 1 else 0
        spl = (lib.lsqamd_debug_flags(h) >> 8) & 0xffffff

        def per(ms_cnt):
            return (ms_cnt[0] / ms_cnt[1] * 1e-3) if ms_cnt[1] else None
        pk = {}
        t = per(tm['whiten'])
        if t and B:
            fl = float(B) * n_local * (P + 1)
            pk['whitening (J_b = W_b [df/dp] per block)'] = {'bound': 'mfma', 'flops': fl, 'ms': t * 1e3, 'frac': fl / t / 1e12 / PEAK_FP64_MFMA_TFLOPS}
        t = per(tm['cholesky'])
        if t:
            fl = P ** 3 / 3.0
            pk['potrf_upper (damped normal equations)'] = {'bound': 'mfma', 'flops': fl, 'ms': t * 1e3, 'frac': fl / t / 1e12 / PEAK_FP64_MFMA_TFLOPS}
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
        out['roofline']['per_kernel'] = pk
        step_flops = flops + (float(B) * n_local * (P + 1) if B else 0.0) + P ** 3 / 3.0 + 2.0 * n_local * P + 2.0 * P * P
        out['roofline']['blended_step'] = {'flops_per_step': step_flops, 'ms_per_step': 1e3 * elapsed / args.steps,
                                           'achieved_TFLOPs': step_flops / (elapsed / args.steps) / 1e12,
                                           'frac': step_flops / (elapsed / args.steps) / 1e12 / PEAK_FP64_MFMA_TFLOPS,
                                           'note': 'algorithmic flops of one accepted step on this rank (SYRK + whitening + Cholesky + '
                                                   'J^T f + solve; rejected trials and model evaluation not counted) over the wall-clock step'}
        if not args.no_cpu_baseline and world == 1:
            try:
                cb, res = cpu_baseline(d, wh, args.cpu_seconds)
                out['cpu_baseline'] = cb
                # the parity half of the metric, at the bench's own size: the device repeats, from the
                # same start, the LM iterations the CPU port (the GSL restatement) just took; chi2 must
                # agree to the north_star tolerance (outside the timed region)
                rc = lib.lsqamd_init(h, _lib.dptr(np.ascontiguousarray(d['p0'])))
                for _ in range(res.nit):
                    if rc == 0:
                        rc = lib.lsqamd_step(h, None)
                s2 = _lib.Summary()
                lib.lsqamd_finish(h, C.byref(s2))
                rel = abs(s2.chi2 - res.fnorm2) / res.fnorm2
                out['config']['chi2_match'] = {'after_lm_steps': int(res.nit), 'device_chi2': s2.chi2,
                                               'cpu_port_chi2': float(res.fnorm2), 'rel_diff': rel,
                                               'ok': bool(rc == 0 and rel < 1e-6)}
                if not B and np.ndim(d['prior'][1]) == 1 and N * P <= 4096 * 256:     # (the uncorrelated workload, c2)
                    try:
                        out['cpu_baseline']['scipy_least_squares_lm'] = scipy_lm_sanity(d, args.cpu_seconds)
                    except Exception as e:
                        out['cpu_baseline']['scipy_least_squares_lm'] = {'value': None, 'sample': 'failed: %r' % (e,)}
            except Exception as e:   # the baseline must never take the measurement down
                out['cpu_baseline'] = {'value': None, 'unit': 'LM steps/s', 'cores': 0, 'kind': 'port',
                                       'sample': 'failed: %r' % (e,)}
        else:
            out['cpu_baseline'] = None
        print(json.dumps(out))
    pr.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
