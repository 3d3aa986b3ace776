"""Helpers shared by the -m gpu tests (tests only)."""
import numpy as np

from lsqfit_amd import synth
from oracle import fit as ofit


def cosmix_fcn(x, p):
    K = p.size // 2
    return np.cos(np.outer(x, p[K:])) @ p[:K]


def cosmix_jac(x, p):
    K = p.size // 2
    wx = np.outer(x, p[K:])
    return np.hstack([np.cos(wx), -p[:K] * x[:, None] * np.sin(wx)])


def multiexp_fcn(x, p):
    K = p.size // 2
    return np.exp(-np.outer(x, p[K:])) @ p[:K]


def multiexp_jac(x, p):
    K = p.size // 2
    e = np.exp(-np.outer(x, p[K:]))
    return np.hstack([e, -p[:K] * x[:, None] * e])


def dense_cov(yerr, N):
    if isinstance(yerr, dict):
        cov = np.diag(np.asarray(yerr['sdev'], float) ** 2)
        for r0, c in yerr['blocks']:
            cov[r0:r0 + c.shape[0], r0:r0 + c.shape[0]] = c
        return cov
    yerr = np.asarray(yerr, float)
    return yerr if yerr.ndim == 2 else yerr


def oracle_normal(d, p, svdcut=1e-12):
    """(chi2, A = J^T J, g = J^T f) of the whitened problem at p from the oracle."""
    from oracle.chiv import Chiv
    N = d['ymean'].size
    pdf = ofit.build_pdf(d['ymean'], dense_cov(d['yerr'], N), d['prior'][0], d['prior'][1], svdcut=svdcut)
    x = d['x']
    f = pdf_residual(pdf, x, p, d.get('fcn', cosmix_fcn))
    J = pdf_jacobian(pdf, x, p, d.get('jac', cosmix_jac))
    return float(f @ f), J.T @ J, J.T @ f, f, J


def _whiten(pdf, delta):
    iw, w = pdf.i_invwgts[0]
    out = [w[:, None] * delta[iw] if delta.ndim == 2 else w * delta[iw]]
    for iw, W in pdf.i_invwgts[1:]:
        out.append(W @ delta[iw])
    return np.concatenate(out, axis=0)


def pdf_residual(pdf, x, p, fcn):
    delta = np.concatenate([fcn(x, p), p]) - pdf.mean
    return _whiten(pdf, delta)


def pdf_jacobian(pdf, x, p, jac):
    Jd = np.vstack([jac(x, p), np.eye(p.size)])
    return _whiten(pdf, Jd)


def oracle_fit(d, solver='cholesky', tol=1e-8, svdcut=1e-12, p0=None, fcn=cosmix_fcn, jac=cosmix_jac, maxit=1000):
    N = d['ymean'].size
    return ofit.nonlinear_fit(d['x'], d['ymean'], dense_cov(d['yerr'], N), fcn, prior_mean=d['prior'][0],
                              prior_err=d['prior'][1], p0=d['p0'] if p0 is None else p0, tol=tol,
                              svdcut=svdcut, jac=jac, solver=solver, maxit=maxit)


def relmax(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))
