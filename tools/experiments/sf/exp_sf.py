"""Developer experiment (GPU), round 5 review item 1: the factorisation streamed behind the J^T J product (csrc/sf_chol.hip).
(1) where do the workgroups of CU-masked streams run (mask bit layout 'c' vs 'i'); (2) the streamed factorisation against numpy;
(3) its time at the 8-GPU shard shape (8192 rows, 4096 parameters, 4 K-chunks) for a few reservations, next to the serial
sequence it replaces (J^T J launch + slab sum + damped matrix + potrf_upper; bench.py --ndata 8192 phases).
usage (after tools/experiments/sf/build.sh): python tools/experiments/sf/exp_sf.py [probe] [check] [time]"""
import ctypes as C
import sys
import time

import numpy as np
import torch

import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import sf_lib

lib = sf_lib.load()
torch.zeros(1, device='cuda')
hip = C.CDLL([ln.split()[-1] for ln in open('/proc/self/maps') if 'libamdhip64' in ln][0])
what = sys.argv[1:] or ['probe', 'check', 'time']
vp = C.c_void_p


def masked_stream(bits):
    words = (C.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    if rc != 0:
        raise RuntimeError('hipExtStreamCreateWithCUMask -> %d' % rc)
    return s


def cu_set(stream, n_wg):
    """(xcc, se, sh, cu) of every workgroup of a launch of n_wg one-per-CU workgroups (they stay ~10 us: several rounds)"""
    out = torch.zeros(2 * n_wg, dtype=torch.int32, device='cuda')
    torch.cuda.synchronize()
    assert lib.lsqamd_debug_where(stream, n_wg, vp(out.data_ptr()), 150 * 1024) == 0
    hip.hipStreamSynchronize(stream)
    v = out.cpu().numpy().astype(np.uint32).reshape(-1, 2)
    xcc, hw = v[:, 0] & 7, v[:, 1]
    return set(zip(xcc.tolist(), ((hw >> 13) & 7).tolist(), ((hw >> 12) & 1).tolist(), ((hw >> 8) & 15).tolist()))


def probe():
    """which CUs do the two masked streams of a reservation really get?  (KFD's mqd_symmetrically_map_cu_mask: bit i -> XCC i % 8,
    then round robin over the shader engines -- layout 'i'; an XCC whose share of the mask is empty seems to run on ALL its CUs)"""
    good = []
    r = 4
    for mode in 'ci':
        chain = [32 * x + j for x in range(8) for j in range(r)] if mode == 'c' else [8 * j + x for x in range(8) for j in range(r)]
        work = sorted(set(range(256)) - set(chain))
        cs, ws = cu_set(masked_stream(chain), 2048), cu_set(masked_stream(work), 4096)
        per = lambda s: [sum(1 for e in s if e[0] == x) for x in range(8)]
        print("mask layout '%s', %d CUs reserved per XCD: chain stream runs on %d CUs %s, worker stream on %d CUs %s, %d in common"
              % (mode, r, len(cs), per(cs), len(ws), per(ws), len(cs & ws)))
        if not (cs & ws) and per(cs) == [r] * 8:
            good.append(mode)
    return good[0] if good else 'i'


def make(N, P, seed=0, prior_dense=True):
    rng = np.random.default_rng(seed)
    J = rng.standard_normal((N, P)) / np.sqrt(N)
    Lam = None
    if prior_dense:
        B = rng.standard_normal((P, P // 4)) / np.sqrt(P)
        Lam = B @ B.T + 0.5 * np.eye(P)
    g = rng.standard_normal(P)
    d = rng.uniform(0.5, 0.9, P)
    return J, Lam, g, d


def run(J, Lam, g, d, mu, splits, group_rows, reserve, mode, reps=1, scaler=0, idle=16, timeline=False):      # 0 = LSQAMD_SCALE_MORE
    N, P = J.shape
    ldj = P + 16
    Jd = torch.zeros(N, ldj, dtype=torch.float64, device='cuda')
    Jd[:, :P] = torch.from_numpy(J)
    Ld = torch.from_numpy(Lam).cuda() if Lam is not None else None
    gd = torch.from_numpy(g).cuda()
    apk = torch.full((P // 128 * (P // 128 + 1) // 2 * 128 * 128,), float('nan'), dtype=torch.float64, device='cuda')
    M = torch.full((P, P + 128), float('nan'), dtype=torch.float64, device='cuda')
    wb = lib.lsqamd_op_sf_work_bytes(N, P, splits)
    work = torch.empty(wb, dtype=torch.uint8, device='cuda')
    info = C.c_int32(0)
    stream = torch.cuda.Stream()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    dbg = torch.zeros(24 + 4 * (P // 128), dtype=torch.int64, device='cuda')
    for rep in range(reps):
        dd = torch.from_numpy(d).cuda()
        torch.cuda.synchronize()
        with torch.cuda.stream(stream):
            ev0.record()
            rc = lib.lsqamd_op_sf_factor(vp(stream.cuda_stream), vp(Jd.data_ptr()), ldj, N, P, splits, group_rows, reserve, ord(mode),
                                         vp(Ld.data_ptr()) if Ld is not None else None, 1, vp(gd.data_ptr()), mu, scaler,
                                         vp(dd.data_ptr()), vp(apk.data_ptr()), vp(M.data_ptr()), vp(work.data_ptr()), wb,
                                         C.byref(info) if rep == reps - 1 else None, vp(dbg.data_ptr()) if timeline else None, idle)
            ev1.record()
        assert rc == 0, rc
        torch.cuda.synchronize()
        ts.append(ev0.elapsed_time(ev1))
    if timeline:
        reserve_ = reserve
        t = dbg.cpu().numpy().astype(np.float64)
        t0 = t[0]
        us = lambda v: (v - t0) / 100.0
        print('  timeline (us after the chain started): J^T J tiles all taken %.0f, last worker left %.0f' % (us(t[1]), us(t[2])))
        T = P // 128
        acct = t[8 + 4 * T:8 + 4 * T + 9].reshape(3, 3)
        for name, (n, work, hand) in zip(('J^T J tile x K-chunk', 'panel tile', 'trailing update'), acct):
            if n:
                print('  %-22s %5d items, product + epilogue %.1f us each, hand-off (drain, barrier, release, flag%s) %.1f us each'
                      % (name, n, work / n / 100.0, ', slab sum of the last' if name[0] == 'J' else '', hand / n / 100.0))
        extra = t[8 + 4 * T + 9:8 + 4 * T + 11]
        print('  summed over the workers: %.1f ms looking for work (idle included), %.1f ms inside taken items waiting for their inputs; '
              'work itself %.1f ms; %d workers x %.2f ms = %.1f ms' % (extra[0] / 1e5, extra[1] / 1e5, (acct[:, 1].sum() + acct[:, 2].sum()) / 1e5,
                                                                        2 * (256 - 8 * reserve_), us(t[2]) / 1e3, 2 * (256 - 8 * reserve_) * us(t[2]) / 1e3))
        rows = []
        for k in range(T):
            a, b, c, e = t[8 + 4 * k:12 + 4 * k]
            rows.append('%d:%.0f/%.0f/%.0f/%.0f' % (k, us(a), us(b), us(c), us(e)))
        if 'steps' in what:
            print('  step: row ready / diagonal block factored / next tile ready / its panel done  ' + '  '.join(rows))
    return apk.cpu().numpy(), M.cpu().numpy(), dd.cpu().numpy(), info.value, ts


def check(N, P, splits, group_rows, reserve, mode):
    J, Lam, g, d, = make(N, P, seed=P + N)
    mu = 0.37
    apk, M, dnew, info, _ = run(J, Lam, g, d, mu, splits, group_rows, reserve, mode)
    T = P // 128
    A = J.T @ J + Lam
    worst = 0.0
    t = 0
    for tm in range(T):
        for tn in range(tm, T):
            tile = apk[t * 16384:(t + 1) * 16384].reshape(128, 128)
            worst = max(worst, np.abs(tile - A[tm * 128:(tm + 1) * 128, tn * 128:(tn + 1) * 128]).max())
            t += 1
    dref = np.maximum(d, np.sqrt(np.diag(A)))
    Ad = A + mu * np.diag(dref ** 2)
    U = np.linalg.cholesky(Ad).T
    y = np.linalg.solve(U.T, g)
    Ug = np.triu(M[:, :P])
    print('check N=%d P=%d splits=%d group=%d reserve=%d/%s: info %d | A %.1e  D %.1e  U %.1e  U^-T g %.1e (relative to the largest entry)'
          % (N, P, splits, group_rows, reserve, mode, info, worst / np.abs(A).max(), np.abs(dnew - dref).max() / dref.max(),
             np.abs(Ug - U).max() / np.abs(U).max(), np.abs(M[:, P] - y).max() / np.abs(y).max()))
    return info == 0 and np.abs(Ug - U).max() / np.abs(U).max() < 1e-11 and worst / np.abs(A).max() < 1e-13


mode = 'c'
if 'probe' in what:
    mode = probe()
    print('# using mask layout', mode)
for a in what:
    if a.startswith('mode='):
        mode = a[5:]
if 'check' in what:
    ok = check(1024, 512, 2, 2, 4, mode)
    ok = check(2048, 1024, 4, 4, 4, mode) and ok
    ok = check(8192, 4096, 4, 4, 4, mode) and ok
    print('check:', 'PASS' if ok else 'FAIL')
if 'time' in what:
    J, Lam, g, d = make(8192, 4096, seed=1)
    import os
    os.environ.pop('LSQAMD_SF_DEBUG', None)
    for splits, group_rows, reserve, ct in ((4, 1, 1, 1), (4, 1, 1, 2), (4, 2, 2, 2)):
        os.environ['LSQAMD_SF_CHAIN_TILES'] = str(ct)
        _, _, _, info, ts = run(J, Lam, g, d, 0.37, splits, group_rows, reserve, mode, reps=5, idle=16, timeline=True)
        print('time (8192, 4096): %d K-chunks, %d-row groups, %d CUs reserved per XCD, chain makes %d tiles: %s ms (info %d)'
              % (splits, group_rows, reserve, ct, ' '.join('%.3f' % t for t in ts), info))
    if 'c4' in what:
        J, Lam, g, d = make(65536, 4096, seed=2)
        for splits, group_rows, reserve in ((16, 2, 1),):
            os.environ['LSQAMD_SF_CHAIN_TILES'] = '2'
            _, _, _, info, ts = run(J, Lam, g, d, 0.37, splits, group_rows, reserve, mode, reps=4, idle=16, timeline=True)
            print('time (65536, 4096): %d K-chunks, %d-row groups, %d CUs reserved per XCD: %s ms (info %d)'
                  % (splits, group_rows, reserve, ' '.join('%.3f' % t for t in ts), info))
if 'potrf' in what:
    # the same machinery with (almost) no J^T J in front of it: the pace of the persistent factorisation alone, next to
    # potrf_upper's 1.48 ms at P = 4096 (tools/time_potrf.py)
    import os
    for Pq in (4096, 1024):
        J, Lam, g, d = make(128, Pq, seed=3)
        for splits, group_rows, reserve, ct in ((1, 1, 1, 1), (1, 2, 1, 1), (1, 1, 2, 1), (1, 1, 1, 2)):
            os.environ['LSQAMD_SF_CHAIN_TILES'] = str(ct)
            _, _, _, info, ts = run(J, Lam, g, d, 0.37, splits, group_rows, reserve, mode, reps=5, idle=16, timeline=True)
            print('time (128, %d): %d K-chunks, %d-row groups, %d CUs reserved per XCD, chain makes %d tiles: %s ms (info %d)'
                  % (Pq, splits, group_rows, reserve, ct, ' '.join('%.3f' % t for t in ts), info))
