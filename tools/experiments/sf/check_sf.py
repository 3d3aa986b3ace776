"""NOT part of the default suite (round 6): run as `tools/experiments/sf/build.sh && python -m pytest tools/experiments/sf/check_sf.py -m gpu`.
The factorisation streamed behind the J^T J product (csrc/sf_chol.hip; developer entry point lsqamd_op_sf_factor) --
a persistent worker launch on all CUs but a few, the latency chain on CU-masked reserved CUs, hand-offs through device flags --
against numpy: packed tiles of A = J^T J + prior, the updated scaling D, U with A + mu D^2 = U^T U, and U^-T g.  Two runs
must agree bit for bit whatever order the workgroups ran in.  Spec: what gsl's solver init + solve compute behind
src/lsqfit/_gsl.pyx:646-653,:677 ("streamed factorisation": measured, slower than the serial chain, not on the LM path)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
vp = C.c_void_p


@pytest.fixture(scope='module')
def lib():
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import sf_lib
    return sf_lib.load()


def run(lib, J, Lam, g, d, mu, splits, group_rows, reserve, scaler, _retry=True):
    import torch
    N, P = J.shape
    ldj = P + 16
    Jd = torch.zeros(N, ldj, dtype=torch.float64, device='cuda')
    Jd[:, :P] = torch.from_numpy(J)
    Ld = None if Lam is None else torch.from_numpy(np.ascontiguousarray(Lam)).cuda()
    gd, dd = torch.from_numpy(g).cuda(), torch.from_numpy(d.copy()).cuda()
    T = P // 128
    apk = torch.full((T * (T + 1) // 2 * 128 * 128,), float('nan'), dtype=torch.float64, device='cuda')
    M = torch.full((P, P + 128), float('nan'), dtype=torch.float64, device='cuda')
    wb = lib.lsqamd_op_sf_work_bytes(N, P, splits)
    work = torch.empty(wb, dtype=torch.uint8, device='cuda')
    info = C.c_int32(0)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        rc = lib.lsqamd_op_sf_factor(vp(st.cuda_stream), vp(Jd.data_ptr()), ldj, N, P, splits, group_rows, reserve, ord('i'),
                                     None if Ld is None else vp(Ld.data_ptr()), 0 if (Lam is None or Lam.ndim == 1) else 1,
                                     vp(gd.data_ptr()), mu, scaler, vp(dd.data_ptr()), vp(apk.data_ptr()), vp(M.data_ptr()),
                                     vp(work.data_ptr()), wb, C.byref(info), None, 16)
    assert rc == 0, rc
    torch.cuda.synchronize()
    if info.value == -88 and _retry:
        # SF_TIMEOUT_INFO: a bounded wait ran out -- the two CU-masked streams were not running at the same time.  The entry
        # point is a measured-and-shelved experiment (not on any fit's path): a box that does not co-schedule the streams
        # must not turn the suite red; one more try, then the case is skipped (any numerical disagreement still fails)
        out = run(lib, J, Lam, g, d, mu, splits, group_rows, reserve, scaler, _retry=False)
        if out[3] == -88:
            pytest.skip('the CU-masked streams of the streamed factorisation were not co-scheduled on this box (info -88 twice)')
        return out
    return apk.cpu().numpy(), M.cpu().numpy(), dd.cpu().numpy(), info.value


@pytest.mark.parametrize('N,P,splits,group_rows,reserve,prior,scaler', [
    (1024, 512, 2, 2, 4, 'dense', 0), (2048, 1024, 4, 1, 1, 'dense', 0), (1536, 768, 3, 4, 2, 'diag', 2), (512, 256, 1, 1, 1, None, 1)])
def test_streamed_factorisation_matches_numpy(lib, N, P, splits, group_rows, reserve, prior, scaler):
    rng = np.random.default_rng(N + P)
    J = rng.standard_normal((N, P)) / np.sqrt(N)
    if prior == 'dense':
        B = rng.standard_normal((P, P // 4)) / np.sqrt(P)
        Lam = B @ B.T + 0.5 * np.eye(P)
        A = J.T @ J + Lam
    elif prior == 'diag':
        Lam = rng.uniform(0.2, 2.0, P)
        A = J.T @ J + np.diag(Lam)
    else:
        Lam = None
        A = J.T @ J + 0.0
    g, d = rng.standard_normal(P), rng.uniform(0.5, 0.9, P)
    mu = 0.37
    apk, M, dnew, info = run(lib, J, Lam, g, d, mu, splits, group_rows, reserve, scaler)
    assert info == 0
    T = P // 128
    t = 0
    for tm in range(T):
        for tn in range(tm, T):
            tile = apk[t * 16384:(t + 1) * 16384].reshape(128, 128)
            assert np.abs(tile - A[tm * 128:(tm + 1) * 128, tn * 128:(tn + 1) * 128]).max() < 1e-13 * np.abs(A).max()
            t += 1
    cn = np.sqrt(np.diag(A))
    dref = {0: np.maximum(d, cn), 1: d, 2: cn}[scaler]       # LSQAMD_SCALE_MORE / _LEVENBERG / _MARQUARDT
    assert np.abs(dnew - dref).max() < 1e-14 * dref.max()
    U = np.linalg.cholesky(A + mu * np.diag(dref ** 2)).T
    assert np.abs(np.triu(M[:, :P]) - U).max() < 1e-12 * np.abs(U).max()
    y = np.linalg.solve(U.T, g)
    assert np.abs(M[:, P] - y).max() < 1e-11 * np.abs(y).max()
    # whoever ran which tile: the same bits
    apk2, M2, d2, info2 = run(lib, J, Lam, g, d, mu, splits, group_rows, reserve, scaler)
    assert info2 == 0 and np.array_equal(apk, apk2) and np.array_equal(np.triu(M[:, :P]), np.triu(M2[:, :P]))
    assert np.array_equal(M[:, P], M2[:, P]) and np.array_equal(dnew, d2)


def test_streamed_factorisation_reports_a_failed_pivot(lib):
    rng = np.random.default_rng(5)
    N, P = 256, 512                        # rank-deficient J^T J, no prior, no damping: not positive definite
    J = rng.standard_normal((N, P)) / np.sqrt(N)
    apk, M, dnew, info = run(lib, J, None, rng.standard_normal(P), np.ones(P), 0.0, 1, 2, 2, 1)
    assert info > 0
