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


_C3 = []


def config3_problem():
    """BASELINE config 3 at full size (one dense 8192 x 8192 data block, dense 1024 x 1024 prior): its generator alone takes 11 s
    (two 8192 x 16384 x 8192 products); three tests of the default suite use it -- made once per session.  Treat as read-only."""
    if not _C3:
        from lsqfit_amd import synth
        _C3.append(synth.make_cosmix(N=8192, P=1024, seed=20262, block=8192, prior_corr=True))
    return _C3[0]


def relmax(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def numpy_normal_equations(d):
    """An independent numpy / LAPACK restatement of the whitened normal equations of a cosmix problem (tests only): every
    covariance block whitened with its Cholesky factor -- any W with W^T W = inv(C) gives the same J^T J, J^T f, chi2
     -- -> (normal_eq(p) -> (A, g, chi2), chi2_fn(p), log det of the whole covariance)."""
    import scipy.linalg as sla
    x, ymean = np.asarray(d['x'], float), np.asarray(d['ymean'], float)
    yerr = d['yerr']
    sd = np.asarray(yerr['sdev'] if isinstance(yerr, dict) else yerr, float)
    blocks = yerr['blocks'] if isinstance(yerr, dict) else []
    in_block = np.zeros(x.size, bool)
    Ls = []
    logdet = 0.0
    for r0, c in blocks:
        L = sla.cholesky(np.asarray(c, float), lower=True)
        Ls.append((r0, L))
        in_block[r0:r0 + L.shape[0]] = True
        logdet += 2.0 * np.sum(np.log(np.diag(L)))
    logdet += 2.0 * np.sum(np.log(sd[~in_block]))
    pm, perr = np.asarray(d['prior'][0], float), np.asarray(d['prior'][1], float)
    P = pm.size
    if perr.ndim == 2:
        Lp = sla.cholesky(perr, lower=True)
        prec = sla.cho_solve((Lp, True), np.eye(P))
        logdet += 2.0 * np.sum(np.log(np.diag(Lp)))
    else:
        prec = np.diag(1.0 / perr ** 2)
        logdet += 2.0 * np.sum(np.log(perr))
    K = P // 2

    def whiten(v):
        out = v / (sd[:, None] if v.ndim == 2 else sd)
        for r0, L in Ls:
            B = L.shape[0]
            out[r0:r0 + B] = sla.solve_triangular(L, v[r0:r0 + B], lower=True)
        return out

    def chi2_fn(p):
        r = whiten(np.cos(np.outer(x, p[K:])) @ p[:K] - ymean)
        dp = p - pm
        return float(r @ r + dp @ prec @ dp)

    def normal_eq(p):
        wx = np.outer(x, p[K:])
        c, s = np.cos(wx), np.sin(wx)
        J = whiten(np.hstack([c, -p[:K] * x[:, None] * s]))
        r = whiten(c @ p[:K] - ymean)
        dp = p - pm
        return J.T @ J + prec, J.T @ r + prec @ dp, float(r @ r + dp @ prec @ dp)

    return normal_eq, chi2_fn, float(logdet)


def threaded_normal_equations(d):
    """the same sums from oracle/port.py (threaded trig, per-block whitening with inv(chol(C_b)), BLAS dsyrk): the full-size
    checks use it (an oracle iteration at (65536, 4096) takes ~6 s with it, ~15 s with the plain restatement above -- which
    the two agree with to 1e-15, tests/test_oracle_port.py)"""
    from oracle.port import CosmixPort
    port = CosmixPort(d)
    return port.normal_eq, port.chi2_fn, port.logdet


def check_fit_vs_normal_oracle(fit, d, p0, tol=1e-6, lm_tol=(1e-8, 1e-10, 1e-10), seed=17, restart=False, fast=False, first=None):
    """a converged device fit against the oracle's LM driver (oracle.lm.lm_normal: the restated gsl_multifit_nlinear trust
    / lm / nielsen / more / convergence, src/lsqfit/_gsl.pyx:563-706) run on numpy_normal_equations: p, chi2/dof, the diagonal
    and 64 random columns of cov, logGBF at ``tol``; -> the oracle's result.
    restart=False: the oracle walks the whole way from the same start ``p0``.  restart=True (where an oracle iteration costs
    ~15 s: config 4 at full size needs 15+ of them even from 1e-7 off the generating values): the oracle is started AT the
    device's answer and must stay there -- its own convergence test fires within two iterations and its end point is the
    device's to ``tol`` -- which is what 'the same converged fit' means; the trajectory itself is compared iteration for
    iteration at (4096, 512) (test_iteration_count_from_the_prior_mean_matches_oracle)."""
    from oracle import lm as olm
    normal_eq, chi2_fn, logdet_c = (threaded_normal_equations if fast else numpy_normal_equations)(d)
    if first is not None:           # the caller wants the oracle's first evaluation (point, A, g, chi2, log det C) as well
        inner = normal_eq

        def normal_eq(p):
            out = inner(p)
            if not first:
                first.extend([np.array(p, float), np.array(out[0]), np.array(out[1]), float(out[2]), logdet_c])
            return out
    ref = olm.lm_normal(fit.pmean if restart else p0, normal_eq, chi2_fn, tol=lm_tol, maxit=1000 if not restart else 3)
    if restart:
        assert ref.nit <= 2 and ref.stopping_criterion in (1, 2), (ref.nit, ref.stopping_criterion)
    P = ref.x.size
    assert relmax(fit.pmean, ref.x) < tol, relmax(fit.pmean, ref.x)
    chi2_ref = chi2_fn(ref.x)
    assert abs(fit.chi2 / fit.dof - chi2_ref / fit.dof) <= tol * chi2_ref / fit.dof
    assert relmax(np.diag(fit.cov), np.diag(ref.cov)) < tol
    cols = np.random.default_rng(seed).choice(P, size=min(64, P), replace=False)
    assert relmax(fit.cov[:, cols], ref.cov[:, cols]) < tol
    sign, ld = np.linalg.slogdet(ref.A)
    logGBF_ref = 0.5 * (-ld - logdet_c - chi2_ref - fit.dof * np.log(2 * np.pi))
    assert sign > 0 and abs(fit.logGBF - logGBF_ref) <= tol * abs(logGBF_ref), (fit.logGBF, logGBF_ref)
    assert fit.stopping_criterion == ref.stopping_criterion or {fit.stopping_criterion, ref.stopping_criterion} <= {1, 2}
    return ref
