"""-m gpu: the raw dense kernels through the C ABI against numpy (fp64, rtol 1e-12)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def env():
    import torch
    from lsqfit_amd import _lib
    lib = _lib.load()
    return torch, lib


def dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize('M,N,K', [(128, 128, 64), (200, 130, 77), (5, 3, 9), (256, 384, 1000), (129, 257, 4),
                                   (1, 1, 1), (300, 300, 3000)])
def test_gemm_tn(env, M, N, K):
    torch, lib = env
    rng = np.random.default_rng(M * 7 + N * 3 + K)
    ldx, ldy, ldc = M + (M & 1) + 2, N + (N & 1) + 4, N + 3
    X = rng.standard_normal((K, ldx))
    Y = rng.standard_normal((K, ldy))
    C0 = rng.standard_normal((M, ldc))
    dX, dY, dC = dev(torch, X), dev(torch, Y), dev(torch, C0)
    rc = lib.lsqamd_op_gemm_tn(None, M, N, K, 0.75, dX.data_ptr(), ldx, dY.data_ptr(), ldy, -0.5,
                               dC.data_ptr(), ldc, 0, 0)
    assert rc == 0
    torch.cuda.synchronize()
    want = C0.copy()
    want[:, :N] = 0.75 * X[:, :M].T @ Y[:, :N] - 0.5 * C0[:, :N]
    got = dC.cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12 * np.abs(want).max())


def test_gemm_tn_unaligned_and_upper(env):
    torch, lib = env
    rng = np.random.default_rng(3)
    M = N = 333
    K = 517
    ld = 335                      # odd leading dimension -> scalar load path
    X = rng.standard_normal((K, ld))
    dX = dev(torch, X)
    dC = torch.full((M, N), np.nan, dtype=torch.float64, device='cuda')
    rc = lib.lsqamd_op_gemm_tn(None, M, N, K, 1.0, dX.data_ptr(), ld, dX.data_ptr(), ld, 0.0,
                               dC.data_ptr(), N, 1, 0)
    assert rc == 0
    torch.cuda.synchronize()
    got = dC.cpu().numpy()
    want = X[:, :M].T @ X[:, :N]
    iu = np.triu_indices(M)
    np.testing.assert_allclose(got[iu], want[iu], rtol=1e-12, atol=1e-11)
    # tiles strictly below the diagonal are never touched
    assert np.all(np.isnan(got[128:, :128][128:, :]))


def test_gemm_tn_triangular_x(env):
    torch, lib = env
    rng = np.random.default_rng(4)
    B, N = 300, 70
    Wt = np.triu(rng.standard_normal((B, B)))      # X[k][m] = 0 for k > m
    Y = rng.standard_normal((B, N + 2))
    dX, dY = dev(torch, Wt), dev(torch, Y)
    dC = torch.zeros((B, N + 2), dtype=torch.float64, device='cuda')
    rc = lib.lsqamd_op_gemm_tn(None, B, N, B, 1.0, dX.data_ptr(), B, dY.data_ptr(), N + 2, 0.0,
                               dC.data_ptr(), N + 2, 0, 1)
    assert rc == 0
    torch.cuda.synchronize()
    np.testing.assert_allclose(dC.cpu().numpy()[:, :N], Wt.T @ Y[:, :N], rtol=1e-12, atol=1e-11)


@pytest.mark.parametrize('n', [1, 7, 128, 129, 300, 520])
def test_potrf_upper_with_rhs_column(env, n):
    torch, lib = env
    rng = np.random.default_rng(n)
    G = rng.standard_normal((n + 20, n))
    A = G.T @ G + 0.1 * np.eye(n)
    b = rng.standard_normal(n)
    lda = n + 1 + ((n + 1) & 1)
    Ab = np.zeros((n, lda))
    Ab[:, :n] = np.triu(A)              # only the upper triangle is read
    Ab[:, n] = b
    dA = dev(torch, Ab)
    wbytes = lib.lsqamd_op_potrf_work_bytes(n)
    work = torch.empty(wbytes // 8 + 8, dtype=torch.float64, device='cuda')
    info = torch.zeros(4, dtype=torch.int32, device='cuda')
    rc = lib.lsqamd_op_potrf_upper(None, dA.data_ptr(), n, lda, n + 1, work.data_ptr(), wbytes, info.data_ptr())
    assert rc == 0
    torch.cuda.synchronize()
    assert int(info[0]) == 0
    out = dA.cpu().numpy()
    U = np.triu(out[:, :n])
    np.testing.assert_allclose(U.T @ U, A, rtol=1e-11, atol=1e-11 * np.abs(A).max())
    Uref = np.linalg.cholesky(A).T
    np.testing.assert_allclose(U, Uref, rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(out[:, n], np.linalg.solve(Uref.T, b), rtol=1e-9, atol=1e-10)


def test_diagonal_block_kernel_every_size(env):
    """The diagonal-block kernel (16-row slabs, leaf in registers) at EVERY block size 1..128: the factor
    and the block inverse the row panels and the back substitution multiply with.  (The inverse of
    sizes 16 k + 4 was once lost to an exec-masked store; nothing but this sweep saw it.)"""
    torch, lib = env
    for n in range(1, 129):
        rng = np.random.default_rng(n)
        G = rng.standard_normal((n + 20, n))
        A = G.T @ G + 0.1 * np.eye(n)
        wbytes = lib.lsqamd_op_potrf_work_bytes(n)
        work = torch.full((wbytes // 8 + 8,), 7.0, dtype=torch.float64, device='cuda')
        info = torch.zeros(4, dtype=torch.int32, device='cuda')
        dA = dev(torch, np.triu(A))
        assert lib.lsqamd_op_potrf_upper(None, dA.data_ptr(), n, n, n, work.data_ptr(), wbytes, info.data_ptr()) == 0
        torch.cuda.synchronize()
        Uref = np.linalg.cholesky(A).T
        U = np.triu(dA.cpu().numpy())
        uinv = work.cpu().numpy()[:128 * 128].reshape(128, 128)[:n, :n]
        assert int(info[0]) == 0, n
        assert np.abs(U - Uref).max() < 1e-11, n
        assert np.abs(uinv @ Uref - np.eye(n)).max() < 1e-11, n
        assert np.all(np.tril(uinv, -1) == 0.0), n


def test_potrf_flags_indefinite(env):
    torch, lib = env
    n = 200
    rng = np.random.default_rng(1)
    G = rng.standard_normal((n, n))
    A = G + G.T                       # indefinite
    dA = dev(torch, np.triu(A))
    wbytes = lib.lsqamd_op_potrf_work_bytes(n)
    work = torch.empty(wbytes // 8 + 8, dtype=torch.float64, device='cuda')
    info = torch.zeros(4, dtype=torch.int32, device='cuda')
    rc = lib.lsqamd_op_potrf_upper(None, dA.data_ptr(), n, n, n, work.data_ptr(), wbytes, info.data_ptr())
    assert rc == 0
    torch.cuda.synchronize()
    assert int(info[0]) > 0


@pytest.mark.parametrize('M,N', [(128, 4224), (128, 77), (100, 1000), (64, 512), (33, 129)])
def test_gemm_tn_in_place_row_panel(env, M, N):
    """C aliases Y (the Cholesky row panel U[k, k+1:] = inv(U_kk)^T A[k, k+1:]): every shape the
    launcher accepts must be computed by workgroups that own whole operand columns -- wide N makes
    workgroups outnumber the CUs, the situation in which a two-tile-row split corrupts the panel."""
    torch, lib = env
    rng = np.random.default_rng(M + N)
    ld = N + (N & 1) + 2
    X = np.triu(rng.standard_normal((M, M))) + 3 * np.eye(M)       # upper triangular, k-major
    Y = rng.standard_normal((M, ld))
    dX, dY = dev(torch, X), dev(torch, Y)
    for rep in range(3):
        dY.copy_(torch.from_numpy(Y))
        rc = lib.lsqamd_op_gemm_tn(None, M, N, M, 1.0, dX.data_ptr(), M, dY.data_ptr(), ld, 0.0,
                                   dY.data_ptr(), ld, 0, 1)
        assert rc == 0
        torch.cuda.synchronize()
        want = X.T @ Y[:, :N]
        np.testing.assert_allclose(dY.cpu().numpy()[:, :N], want, rtol=1e-12, atol=1e-12 * np.abs(want).max())


def test_gemm_tn_rejects_unsafe_in_place(env):
    """Two tile rows of an in-place product would race: the launcher refuses instead."""
    torch, lib = env
    X = dev(torch, np.eye(256))
    Y = dev(torch, np.ones((256, 512)))
    rc = lib.lsqamd_op_gemm_tn(None, 256, 512, 256, 1.0, X.data_ptr(), 256, Y.data_ptr(), 512, 0.0,
                               Y.data_ptr(), 512, 0, 0)
    assert rc != 0


_JTJ_HASH_SCRIPT = r'''
import hashlib, sys
import numpy as np
sys.path.insert(0, %(root)r)
import lsqfit_amd as amd
from lsqfit_amd import synth
out = []
for N, P in ((4096, 384), (2048, 128), (8192, 1024)):
    d = synth.make_cosmix(N=N, P=P, seed=7, block=0, prior_corr=False)
    pr = amd.DeviceProblem(d['model'], d['x'], amd.Whitening(d['ymean'], d['yerr'], *d['prior']))
    pr.normal(d['p0'])
    A = pr.get_jtj()
    assert np.array_equal(A, A.T)
    out.append(hashlib.sha256(np.ascontiguousarray(A).tobytes()).hexdigest())
    pr.close()
print(' '.join(out))
'''


def test_syrk_diagonal_tile_schedule_is_bit_identical(tmp_path):
    """Diagonal tiles of the split-K J^T J launch take a triangular schedule (10 MFMAs per k-step and
    wave instead of 16, values mirrored in the epilogue).  The packed normal matrix must be the SAME
    BITS as with the full schedule (LSQAMD_SYRK_DIAG=0): same products, same order over k."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / 'jtj_hash.py'
    script.write_text(_JTJ_HASH_SCRIPT % dict(root=root))
    got = []
    for knob in ('1', '0'):
        env = dict(os.environ, LSQAMD_SYRK_DIAG=knob)
        r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        got.append(r.stdout.strip().splitlines()[-1])
    assert got[0] == got[1] and len(got[0].split()) == 3


@pytest.mark.parametrize('P', [256, 640, 1152])
def test_chained_back_substitution_matches_lapack_and_repeats_bit_for_bit(P):
    """U v = y as ONE launch: workgroup i owns block row i and hands its 1 KiB piece of v to the rows
    above through tagged 8-byte granules (chol.hip backsolve_chain_kernel).  The solution must agree
    with LAPACK and -- the hand-offs carry no timing dependence -- repeat bit for bit."""
    import lsqfit_amd as amd
    from lsqfit_amd import synth
    d = synth.make_cosmix(N=2 * P, P=P, seed=3, block=0, prior_corr=False)
    pr = amd.DeviceProblem(d['model'], d['x'], amd.Whitening(d['ymean'], d['yerr'], *d['prior']))
    pr.normal(d['p0'])
    A, g = pr.get_jtj(), pr.get_grad()
    diag = np.sqrt(np.diag(A))
    mu = 1e-2
    ref = np.linalg.solve(A + mu * np.diag(diag ** 2), g)
    first = pr.solve_damped(mu, diag)
    assert np.max(np.abs(np.abs(first) - np.abs(ref))) <= 1e-9 * np.max(np.abs(ref))
    for _ in range(40):
        assert np.array_equal(pr.solve_damped(mu, diag), first)
    pr.close()


@pytest.mark.parametrize('n', [1536, 4096])
def test_potrf_upper_tile_aligned_fused_path(env, n):
    """Tile-aligned sizes take the fused launches (trailing update + next diagonal block by the pivot-wave
    kernel): n = 1536 stays within one round of tile workgroups, n = 4096 goes through every regime --
    64-column half tiles while an update needs more than one round, whole tiles, the unfused tail.  Factor,
    forward-substituted right-hand side, and the 128 x 128 inverse blocks against LAPACK."""
    torch, lib = env
    rng = np.random.default_rng(n)
    G = rng.standard_normal((n + 64, n))
    A = G.T @ G + 0.1 * np.eye(n)
    b = rng.standard_normal(n)
    lda = n + 128                       # the product's padding: every Cholesky GEMM stays tile-aligned
    Ab = np.zeros((n, lda))
    Ab[:, :n] = np.triu(A)
    Ab[:, n] = b
    dA = dev(torch, Ab)
    wbytes = lib.lsqamd_op_potrf_work_bytes(n)
    work = torch.full((wbytes // 8 + 8,), float('nan'), dtype=torch.float64, device='cuda')
    info = torch.zeros(4, dtype=torch.int32, device='cuda')
    rc = lib.lsqamd_op_potrf_upper(None, dA.data_ptr(), n, lda, n + 128, work.data_ptr(), wbytes, info.data_ptr())
    assert rc == 0
    torch.cuda.synchronize()
    assert int(info[0]) == 0
    out = dA.cpu().numpy()
    U = np.triu(out[:, :n])
    Uref = np.linalg.cholesky(A).T
    scale = np.abs(Uref).max()
    assert np.abs(U - Uref).max() < 1e-11 * scale
    assert np.abs(out[:, n] - np.linalg.solve(Uref.T, b)).max() < 1e-10 * np.abs(b).max()
    w = work.cpu().numpy()[:(n // 128) * 128 * 128].reshape(n // 128, 128, 128)
    assert np.all(np.isfinite(w))                                    # zeros above the diagonal included
    for k in (0, 1, n // 256, n // 128 - 1):
        Ukk = Uref[128 * k:128 * (k + 1), 128 * k:128 * (k + 1)]
        assert np.abs(w[k] @ Ukk - np.eye(128)).max() < 1e-9
        assert np.all(np.tril(w[k], -1) == 0.0)


_POTRF_VARIANT_SCRIPT = r'''
import ctypes as C, sys
import numpy as np
sys.path.insert(0, %(root)r)
import torch
from lsqfit_amd import _lib
lib = _lib.load()
for n in (96, 300, 1536):
    rng = np.random.default_rng(n)
    G = rng.standard_normal((n + 20, n))
    A = G.T @ G + 0.1 * np.eye(n)
    b = rng.standard_normal(n)
    lda = n + 128 if n %% 128 == 0 else n + 1 + ((n + 1) & 1)
    ncols = n + 128 if n %% 128 == 0 else n + 1
    Ab = np.zeros((n, lda)); Ab[:, :n] = np.triu(A); Ab[:, n] = b
    dA = torch.from_numpy(Ab).cuda()
    wb = lib.lsqamd_op_potrf_work_bytes(n)
    work = torch.zeros(wb // 8 + 8, dtype=torch.float64, device='cuda')
    info = torch.zeros(4, dtype=torch.int32, device='cuda')
    assert lib.lsqamd_op_potrf_upper(None, dA.data_ptr(), n, lda, ncols, work.data_ptr(), wb, info.data_ptr()) == 0
    torch.cuda.synchronize()
    out = dA.cpu().numpy()
    Uref = np.linalg.cholesky(A).T
    assert int(info[0]) == 0
    assert np.abs(np.triu(out[:, :n]) - Uref).max() < 1e-10 * np.abs(Uref).max(), n
    assert np.abs(out[:, n] - np.linalg.solve(Uref.T, b)).max() < 1e-9 * np.abs(b).max(), n
    nb0 = min(n, 128)
    w = work.cpu().numpy()[:128 * 128].reshape(128, 128)[:nb0, :nb0]
    assert np.abs(w @ Uref[:nb0, :nb0] - np.eye(nb0)).max() < 1e-9, n
print('ok')
'''


@pytest.mark.parametrize('variant', ['v3', 'v2', 'lds'])
def test_selectable_diagonal_kernels_still_factor(tmp_path, variant):
    """LSQAMD_POTF2 = v3 (four-wave 16-row slabs), v2 (4-row sweep), lds (LDS-resident): the earlier
    formulations of the diagonal-block kernel stay selectable and correct (ragged, padded and tile-aligned n)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / 'potrf_variant.py'
    script.write_text(_POTRF_VARIANT_SCRIPT % dict(root=root))
    r = subprocess.run([sys.executable, str(script)], env=dict(os.environ, LSQAMD_POTF2=variant), capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith('ok'), (r.stdout[-500:], r.stderr[-2000:])
