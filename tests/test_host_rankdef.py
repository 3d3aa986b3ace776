"""-m "not gpu": the host arithmetic of the rank-deficient covariance (lsqfit_amd/csrc/rankdef.hip) against the
oracle's restatement of gsl_multifit_nlinear_covar (pivoted QR, epsrel = 0: src/lsqfit/_gsl.pyx:704-706) and
against the thresholded SVD of src/lsqfit/_scipy.py:170-175 evaluated by numpy on the Jacobian itself."""
import ctypes as C

import numpy as np
import pytest

from oracle import lm as olm


@pytest.fixture(scope='module')
def lib():
    from lsqfit_amd import _lib
    return _lib.load()


def truncated(lib, G, n_rows, scipy_form):
    from lsqfit_amd import _lib
    G = np.ascontiguousarray(G, np.float64)
    out = np.empty_like(G)
    k = C.c_int32(-1)
    assert lib.lsqamd_op_truncated_inverse(_lib.dptr(G), G.shape[0], n_rows, scipy_form, _lib.dptr(out), C.byref(k)) == 0
    return out, k.value


def jacobians():
    rng = np.random.default_rng(8)
    J1 = np.diag([0.0, 2.3, -0.7])                                    # the reference's plugin test: a dead column
    J2 = rng.standard_normal((40, 7))
    J2[:, 4] = 0.0                                                    # one parameter the data does not see
    J3 = rng.standard_normal((60, 9))
    J3[:, 2] = 0.0
    J3[:, 7] = 0.0
    J4 = rng.standard_normal((30, 6))                                 # full rank: nothing dropped
    return [J1, J2, J3, J4]


@pytest.mark.parametrize('k', range(4))
def test_gsl_covar_recipe(lib, k):
    J = jacobians()[k]
    lin = olm._DenseLin('qr')
    lin.set(J, np.zeros(J.shape[0]))
    want = lin.covar()
    got, dropped = truncated(lib, J.T @ J, J.shape[0], 0)
    assert dropped == int(np.sum(np.all(J == 0.0, axis=0)))
    assert np.allclose(got, want, rtol=1e-9, atol=1e-12 * np.abs(want).max())
    assert np.array_equal(got, got.T)


@pytest.mark.parametrize('k', range(4))
def test_scipy_pseudo_inverse_recipe(lib, k):
    J = jacobians()[k]
    _, s, VT = np.linalg.svd(J, full_matrices=False)                  # _scipy.py:170-175
    thr = np.finfo(float).eps * max(J.shape) * s[0]
    keep = s > thr
    want = (VT[keep].T / s[keep] ** 2) @ VT[keep]
    got, dropped = truncated(lib, J.T @ J, J.shape[0], 1)
    assert dropped == int(np.sum(~keep))
    assert np.allclose(got, want, rtol=1e-9, atol=1e-12 * np.abs(want).max())


def test_dependent_columns_are_a_null_direction_for_the_pseudo_inverse(lib):
    """two identical columns: rank P - 1 without any zero column; the SVD form drops the difference direction"""
    rng = np.random.default_rng(9)
    J = rng.standard_normal((50, 5))
    J[:, 3] = J[:, 1]
    _, s, VT = np.linalg.svd(J, full_matrices=False)
    keep = s > np.finfo(float).eps * 50 * s[0]
    want = (VT[keep].T / s[keep] ** 2) @ VT[keep]
    got, dropped = truncated(lib, J.T @ J, 50, 1)
    assert dropped == 1 and np.allclose(got, want, rtol=1e-7, atol=1e-10 * np.abs(want).max())
