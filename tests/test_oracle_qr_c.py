"""-m "not gpu": the oracle's C restatement of an unblocked column-pivoted Householder QR (oracle/csrc/qrpt_unblocked.c:
the algorithm class of the reference's default `qr` solver, src/lsqfit/_gsl.pyx:646-647) against LAPACK's pivoted QR
through scipy -- same pivots, same |R|, R^T R = (A P)^T (A P)."""
import numpy as np
import pytest
import scipy.linalg as sla

from oracle import build_c


@pytest.mark.parametrize('m,n,seed', [(40, 7, 1), (300, 64, 2), (64, 64, 3), (1000, 130, 4)])
def test_unblocked_qrpt_matches_lapack(m, n, seed):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((m, n)) * np.logspace(0, 3, n)[None, :]
    R, perm = build_c.qrpt(A)
    Rl, pl = sla.qr(A, mode='r', pivoting=True)
    assert np.array_equal(perm, pl)
    assert np.allclose(np.abs(R), np.abs(Rl[:n]), rtol=1e-10, atol=1e-10 * np.abs(Rl).max())
    AP = A[:, perm]
    assert np.allclose(R.T @ R, AP.T @ AP, rtol=1e-11, atol=1e-9 * np.abs(AP.T @ AP).max())
    assert np.all(np.abs(np.diag(R))[:-1] >= np.abs(np.diag(R))[1:] * (1 - 1e-12))     # pivoting: non-increasing diagonal


def test_covariance_from_the_unblocked_factor_is_the_oracles():
    """gsl_multifit_nlinear_covar's recipe on this R equals oracle.lm's covar() (which uses LAPACK's pivoted QR)."""
    from oracle import lm as olm
    rng = np.random.default_rng(5)
    J = rng.standard_normal((80, 9))
    R, perm = build_c.qrpt(J)
    Rinv = sla.solve_triangular(R, np.eye(9))
    cov = np.zeros((9, 9))
    cov[np.ix_(perm, perm)] = Rinv @ Rinv.T
    lin = olm._DenseLin('qr')
    lin.set(J, np.zeros(80))
    assert np.allclose(cov, lin.covar(), rtol=1e-10, atol=1e-13)
