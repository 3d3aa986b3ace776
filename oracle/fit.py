"""``nonlinear_fit`` problem setup and result reduction (ORACLE ONLY).

Restates the numerical content of src/lsqfit/__init__.py:455-737 without
gvar objects: data and prior arrive as (mean, sdev-or-cov) arrays.

  * ``_unpack_data``  (:1840-1901): concat(y, prior) -> PDF(svdcut); ``udata``
    drops data correlations (:1892-1893);
  * ``_unpack_p0``    (:1947-1948): default start = prior mean, or
    ``mean + 0.1*sdev`` where the mean is exactly 0;
  * ``nf = nchiv``, ``dof = nf - P`` (:574-575);
  * fitter call (:662-664) -> ``chi2 = sum f**2`` (:667), ``Q`` (:670);
  * ``logGBF = (-logdet(J^T J) - pdf.logdet - chi2 - dof log 2pi)/2``
    (:709-725), ``None`` without a prior (:711-712).
"""
import numpy as np
from scipy.special import gammaincc

from .chiv import Chiv
from .lm import gsl_multifit
from .pdf import PDF


def gammaQ(a, x):
    """src/lsqfit/_scipy.py:16-18."""
    return float(gammaincc(a, x))


def _spec(mean, err):
    """(mean, sdev[n]) or (mean, cov[n,n]) -> mean, sdev, dense-cov-or-None."""
    mean = np.asarray(mean, float).reshape(-1)
    err = np.asarray(err, float)
    if err.ndim == 2:
        return mean, np.sqrt(np.diag(err)), err
    return mean, err.reshape(-1) * np.ones_like(mean), None


def build_pdf(ymean, yerr, prior_mean=None, prior_err=None, svdcut=1e-12,
              udata=False, extra_cov=None, eps=None):
    """``extra_cov``: optional list of ((i, j), value) cross-covariances between
    entries of concat(y, prior) (data-prior correlations, test_lsqfit.py:1000-1017)."""
    from .pdf import find_blocks
    ymean, ysd, ycov = _spec(ymean, yerr)
    if udata:
        ycov = None
    if prior_mean is None:
        mean, sd = ymean, ysd
        covs = [(0, ycov)]
    else:
        pmean, psd, pcov = _spec(prior_mean, prior_err)
        mean = np.concatenate([ymean, pmean])
        sd = np.concatenate([ysd, psd])
        covs = [(0, ycov), (ymean.size, pcov)]
    if extra_cov:
        n = mean.size
        full = np.diag(sd ** 2)
        for off, c in covs:
            if c is not None:
                full[off:off + c.shape[0], off:off + c.shape[0]] = c
        for (i, j), v in extra_cov:
            full[i, j] = full[j, i] = v
        return PDF.from_dense(mean, full, svdcut=svdcut, eps=eps)
    blocks = []
    for off, c in covs:
        if c is None:
            continue
        for comp in find_blocks(c):
            if comp.size > 1:
                blocks.append((comp + off, c[np.ix_(comp, comp)]))
    return PDF(mean, sd, blocks, svdcut=svdcut, eps=eps)


def default_p0(prior_mean, prior_sdev):
    pm = np.asarray(prior_mean, float).reshape(-1)
    ps = np.asarray(prior_sdev, float).reshape(-1)
    return np.where(pm != 0.0, pm, pm + 0.1 * ps)


class FitResult:
    pass


def nonlinear_fit(x, ymean, yerr, fcn, prior_mean=None, prior_err=None, p0=None,
                  svdcut=1e-12, tol=1e-8, maxit=1000, udata=False, extra_cov=None,
                  jac=None, fitter='gsl_multifit', linear=None, eps=None, **fitterargs):
    """``fcn(x, p)`` must accept float arrays and ``oracle.dual.Dual`` arrays
    (or pass ``jac(x, p)`` returning d fcn / d p explicitly)."""
    from .dual import Dual
    pdf = build_pdf(ymean, yerr, prior_mean, prior_err, svdcut=svdcut,
                    udata=udata, extra_cov=extra_cov, eps=eps)
    noprior = prior_mean is None
    if p0 is None:
        if noprior:
            raise ValueError('neither p0 nor prior is specified')
        _, psd, _ = _spec(prior_mean, prior_err)
        p0 = default_p0(prior_mean, psd)
    p0 = np.asarray(p0, float).reshape(-1)

    def flatfcn(p):
        ans = fcn(p) if x is False else fcn(x, p)
        if isinstance(ans, Dual):
            return ans.reshape(-1)
        if isinstance(ans, (list, tuple)) and any(isinstance(a, Dual) for a in ans):
            from .dual import concatenate
            return concatenate([a if isinstance(a, Dual) else np.asarray(a, float).reshape(-1)
                                for a in ans])
        return np.asarray(ans, float).reshape(-1)

    chiv = Chiv(pdf, flatfcn, noprior)
    if jac is None:
        dchiv = chiv.jacobian
    else:
        def dchiv(p):
            Jf = np.asarray(jac(p) if x is False else jac(x, p), float)
            Jd = Jf if noprior else np.vstack([Jf, np.eye(p.size)])
            out = np.zeros((pdf.nchiv, p.size))
            iw, w = pdf.i_invwgts[0]
            i2 = len(iw)
            out[:i2] = w[:, None] * Jd[iw]
            for iw, W in pdf.i_invwgts[1:]:
                i1, i2 = i2, i2 + len(W)
                out[i1:i2] = W @ Jd[iw]
            return out
    nf = pdf.nchiv
    fit = FitResult()
    fit.pdf = pdf
    fit.p0 = p0
    fit.dof = nf - p0.size
    fit.svdn = pdf.nmod
    fit.nblocks = pdf.nblocks
    fit.svdcut = svdcut
    if maxit == 0:
        # src/lsqfit/__init__.py:683-706: no fit; parameters = prior (or p0 with infinite errors)
        if noprior:
            fit.pmean = np.array(p0)
            fit.psdev = np.full(p0.size, np.inf)
            fit.cov = np.diag(fit.psdev ** 2)
        else:
            pm, psd, pcov = _spec(prior_mean, prior_err)
            fit.pmean, fit.psdev = pm, psd
            fit.cov = pcov if pcov is not None else np.diag(psd ** 2)
        fit.chiv = chiv
        fit.error = None
        fit.residuals = chiv.residual(fit.pmean)
        fit.chi2 = float(np.sum(fit.residuals ** 2))
        fit.Q = gammaQ(fit.dof / 2., fit.chi2 / 2.)
        fit.nit, fit.tol, fit.stopping_criterion, fit.description = 0, tol, 0, ''
        if noprior:
            fit.logGBF = None
        else:   # :718-725 without J: logdet(cov)
            sign, ld = np.linalg.slogdet(fit.cov)
            fit.logGBF = 0.5 * (ld - pdf.logdet - fit.chi2 - fit.dof * np.log(2. * np.pi))
        return fit
    if fitter == 'scipy_least_squares':     # _scipy.py:115-181 (bounds: __init__.py:641-655)
        from .trf import scipy_least_squares
        if fitterargs.get('bounds') is not None:
            lo, hi = fitterargs['bounds']
            fitterargs['bounds'] = (np.reshape(lo, -1), np.reshape(hi, -1))
        lm = scipy_least_squares(p0, nf, chiv.residual, dchiv, tol=tol, maxit=maxit, **fitterargs)
    else:
        if linear is not None and len(linear) > 0:
            # src/lsqfit/__init__.py:738-787 (_varpro_fit): parameters the fit function is linear in.
            # The reference wraps the fit function so that every evaluation solves for them
            # exactly (variable projection); oracle/lm.py iterate_varpro does the same on the
            # normal equations.
            mask = np.zeros(p0.size, bool)
            mask[np.asarray(linear, int)] = True
            fitterargs = dict(fitterargs, undamped=mask)
        lm = gsl_multifit(p0, nf, chiv.residual, dchiv, tol=tol, maxit=maxit, **fitterargs)
    fit.lm = lm
    fit.chiv = chiv
    fit.error = lm.error
    fit.cov = lm.cov
    fit.chi2 = float(np.sum(lm.f ** 2))
    fit.J = lm.J
    fit.residuals = np.array(lm.f)
    fit.Q = gammaQ(fit.dof / 2., fit.chi2 / 2.)
    fit.nit = lm.nit
    fit.tol = lm.tol
    fit.stopping_criterion = lm.stopping_criterion
    fit.description = lm.description
    fit.pmean = np.array(lm.x)
    fit.psdev = np.sqrt(np.diag(lm.cov))
    if noprior:
        fit.logGBF = None
    else:
        sign, ld = np.linalg.slogdet(lm.J.T @ lm.J)
        fit.logdet_JtJ = float(ld)
        fit.logGBF = 0.5 * (-ld - pdf.logdet - fit.chi2 - fit.dof * np.log(2. * np.pi))
    return fit


def dp_dinputs(fit):
    """``D[a, i] = d pmean[a] / d buf[i]``, ``buf = concat(y, prior)`` in the
    caller's order: the matrix ``_getp`` builds (src/lsqfit/__init__.py:897-911)
    from ``chivw`` (src/lsqfit/_utilities.pyx:96-139), whose value is
    ``inv(C_reg) @ delta`` -- ``wgts**2 * delta`` for the 1x1 entries (:127-129),
    ``(sum_j w_j w_j^T) @ delta`` for each block (:130-134).  Its p-derivative
    contracted with ``cov`` (``mdotder``, :908) gives
    ``D = cov @ [J_f ; I]^T @ inv(C_reg)`` (doc/source/lsqfit.rst:105-117).

    ``fit.J`` holds the rows of ``W @ [J_f ; I]`` in chiv order (1x1 rows first,
    then the modes of each block), so ``D[:, iw] = cov @ J_rows^T @ W``.
    """
    pdf, J, cov = fit.pdf, fit.J, fit.cov
    D = np.zeros((cov.shape[0], pdf.mean.size))
    iw, w = pdf.i_invwgts[0]
    i2 = len(iw)
    if i2:
        D[:, iw] = cov @ (J[:i2] * w[:, None]).T
    for iw, W in pdf.i_invwgts[1:]:
        i1, i2 = i2, i2 + len(W)
        D[:, iw] = cov @ J[i1:i2].T @ W
    return D


def partial_sdev(D, grads, groups, cov_in):
    """Error budget (gvar.fmt_errorbudget as used in examples/simple.py:56-61):
    for output ``g`` with gradient ``grads[g]`` (d g / d p at pmean) and input group
    ``S`` (indices into buf), the partial variance is ``d_S C_SS d_S^T`` with
    ``d = grad @ D``.  ``cov_in`` is the (dense) covariance of buf."""
    out = {}
    for g, grad in grads.items():
        d = np.asarray(grad, float) @ D
        for name, idx in groups.items():
            idx = np.asarray(idx, int)
            out[g, name] = float(np.sqrt(max(d[idx] @ cov_in[np.ix_(idx, idx)] @ d[idx], 0.0)))
    return out


def dchi2(fit, p):
    """``chi**2(p) - fit.chi2`` (``_fit_dchi2.__call__``, src/lsqfit/__init__.py:1664-1669);
    ``p`` (P,) or (m, P) -- the latter is vegas_fit._chiv's lbatch layout
    (src/lsqfit/_extras.py:2467-2486)."""
    p = np.asarray(p, float)
    if p.ndim == 1:
        return float(np.sum(fit.chiv.residual(p) ** 2) - fit.chi2)
    return np.array([np.sum(fit.chiv.residual(q) ** 2) for q in p]) - fit.chi2


def pdf(fit, p):
    """``exp(-dchi2/2)`` (``_fit_pdf.__call__``, src/lsqfit/__init__.py:1811-1816)."""
    return np.exp(-0.5 * dchi2(fit, p))


def pdf_lognorm(fit):
    """``_fit_pdf.lognorm`` (src/lsqfit/__init__.py:1806-1809)."""
    return 0.5 * (fit.pdf.logdet + np.log(2 * np.pi) * (fit.dof + fit.pmean.size)) + fit.chi2 / 2
