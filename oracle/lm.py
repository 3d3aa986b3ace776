"""Trust-region Levenberg-Marquardt driver (ORACLE ONLY).

Restates what src/lsqfit/_gsl.pyx:563-723 obtains from GSL's
``gsl_multifit_nlinear`` with ``trs=lm`` (:622-623), scalers
more/levenberg/marquardt (:637-644), solvers qr/cholesky/svd (:646-653),
``factor_up=3, factor_down=2`` (:573-574,:656-657), the driver call
``gsl_multifit_nlinear_driver(maxit, xtol, gtol, ftol, ...)`` (:677), the
``info`` -> ``stopping_criterion`` map (:690-701), the error strings
(:686-687,:714-717) and ``gsl_multifit_nlinear_covar(J, 0.0, covar)``
(:704-706).

GSL is a third-party dependency ("v2.2.1 or greater", INSTALLATION.txt:6)
and is NOT under /root/reference; the algorithm below is restated from its
published sources (multifit_nlinear/{fdf,trust,lm,nielsen,scaling,
convergence,qr,cholesky,covar}.c):

  init     f=F(x); J=dF(x); g=J^T f; D from scaler; mu=1e-3*max_j(|J_j|/D_j)^2; nu=2
  iterate  solver.init(J); loop { solve (J^T J + mu D^2) v = -g ; f_t=F(x+v);
           rho=(1-(|f_t|/|f|)^2)/((|Jv|^2+2 mu |Dv|^2)/|f|^2)  (-1 if |f_t|>=|f|);
           rho>0: accept, J=dF, g, D update, mu*=max(1/3,1-(2rho-1)^3), nu=2
           else : mu*=nu, nu*=2, >15 consecutive rejections -> ENOPROG(27) }
  driver   ENOPROG on the first iteration -> info=27, EMAXITER(11);
           convergence: |dx_i| < xtol^2+xtol|x_i| for all i -> info 1;
           max_i|g_i max(x_i,1)| <= gtol max(|f|^2/2,1) -> info 2; ftol disabled.

Iteration counts are pinned nowhere in the reference; only the
converged (x, cov, f) and ``stopping_criterion`` are.

A second entry point, ``lm_normal``, runs the identical driver from the
normal equations (A=J^T J, g=J^T f, |f|^2) supplied by a callback; it is the
form a row-sharded evaluation reduces to (one sum over shards per Jacobian)
and is what the multi-process tests use.
"""
import numpy as np
import scipy.linalg as sla

GSL_SUCCESS, GSL_CONTINUE = 0, -2
GSL_EMAXITER, GSL_ENOPROG = 11, 27
_STRERROR = {11: 'exceeded max number of iterations',
             27: 'iteration is not making progress towards solution'}


def normalize_tol(tol):
    """_gsl.pyx:594-603."""
    shape = np.shape(tol)
    if shape == ():
        return (tol, 1e-10, 1e-10)
    if shape == (1,):
        return (tol[0], 1e-10, 1e-10)
    if shape == (2,):
        return (tol[0], tol[1], 1e-10)
    if shape != (3,):
        raise ValueError('tol must be number or a 1-, 2-, or 3-tuple')
    return tuple(tol)


# ---------------------------------------------------------------- linear algebra
class _DenseLin:
    """Holds J; solves the damped step with the chosen GSL solver."""

    def __init__(self, solver):
        self.solver = solver

    def set(self, J, f):
        self.J, self.f = J, f
        self.g = J.T @ f
        self.colnorm = np.sqrt(np.einsum('ij,ij->j', J, J))
        if self.solver == 'cholesky':
            self.A = J.T @ J

    def step(self, mu, diag):
        J, f = self.J, self.f
        P = J.shape[1]
        if self.solver == 'cholesky':
            M = self.A + mu * np.diag(diag ** 2)
            c = sla.cho_factor(M, lower=True)
            return -sla.cho_solve(c, self.g)
        aug = np.vstack([J, np.sqrt(mu) * np.diag(diag)])
        rhs = np.concatenate([f, np.zeros(P)])
        if self.solver == 'qr':
            Q, R = np.linalg.qr(aug)
            return -sla.solve_triangular(R, Q.T @ rhs)
        if self.solver == 'svd':
            return -np.linalg.lstsq(aug, rhs, rcond=None)[0]
        raise ValueError('unkown solver ' + str(self.solver))

    def norm_Jv2(self, v):
        w = self.J @ v
        return float(w @ w)

    def covar(self):
        """gsl_multifit_nlinear_covar(J, epsrel=0): pivoted QR, (R^T R)^-1."""
        J = self.J
        R, piv = sla.qr(J, mode='r', pivoting=True)
        P = J.shape[1]
        R = R[:P, :P]
        zero = np.nonzero(np.abs(np.diag(R)) <= 0.0)[0]   # tolr = epsrel*|R00| = 0
        k = int(zero[0]) if zero.size else P
        cov_p = np.zeros((P, P))
        if k > 0:
            Rinv = sla.solve_triangular(R[:k, :k], np.eye(k))
            cov_p[:k, :k] = Rinv @ Rinv.T
        cov = np.zeros((P, P))
        cov[np.ix_(piv, piv)] = cov_p
        return cov


class _NormalLin:
    """Same interface from (A, g) only: the 'cholesky' solver's algebra."""

    def set(self, A, g):
        self.A, self.g = A, g
        self.colnorm = np.sqrt(np.diag(A))

    def step(self, mu, diag):
        M = self.A + mu * np.diag(diag ** 2)
        c = sla.cho_factor(M, lower=True)
        return -sla.cho_solve(c, self.g)

    def norm_Jv2(self, v):
        return float(v @ (self.A @ v))

    def covar(self):
        c = sla.cho_factor(self.A, lower=True)
        return sla.cho_solve(c, np.eye(self.A.shape[0]))


# ---------------------------------------------------------------- scaling.c
def _scale_init(scaler, colnorm):
    if scaler == 'levenberg':
        return np.ones_like(colnorm)
    if scaler in ('more', 'marquardt'):
        d = colnorm.copy()
        d[d == 0.0] = 1.0
        return d
    raise ValueError('unkown scaler ' + str(scaler))


def _scale_update(scaler, colnorm, diag):
    if scaler == 'levenberg':
        return diag
    if scaler == 'more':
        return np.maximum(diag, colnorm)
    d = colnorm.copy()
    d[d == 0.0] = 1.0
    return d


class LMResult:
    pass


def _drive(x0, evaluate, eval_fnorm2, lin, tol, maxit, scaler, factor_up, factor_down):
    """Shared trust.c/fdf.c logic.  ``evaluate(x)`` refreshes ``lin`` with the
    Jacobian-level quantities at x and returns |f|^2; ``eval_fnorm2(x)`` is the
    cheap trial evaluation."""
    xtol, gtol, ftol = tol
    x = np.array(x0, float)
    P = x.size
    res = LMResult()
    res.nfev = res.njev = 0
    res.ntrial = 0

    # trust_init
    fnorm2 = evaluate(x)
    res.nfev += 1
    res.njev += 1
    diag = _scale_init(scaler, lin.colnorm)
    mu = 1e-3 * float(np.max(lin.colnorm / diag)) ** 2 if P else 0.0   # nielsen_init
    nu = 2
    delta = 0.3 * max(1.0, float(np.linalg.norm(diag * x)))
    dx = np.zeros(P)

    def iterate():
        nonlocal x, fnorm2, diag, mu, nu, delta, dx
        bad_steps = 0
        while True:
            v = lin.step(mu, diag)                        # lm_step
            dx = v
            x_trial = x + dx
            ft2 = eval_fnorm2(x_trial)
            res.nfev += 1
            res.ntrial += 1
            # trust_calc_rho
            normf, normf_trial = np.sqrt(fnorm2), np.sqrt(ft2)
            if not (normf_trial < normf):
                rho = -1.0
            else:
                u = normf_trial / normf
                actual = 1.0 - u * u
                un = np.sqrt(lin.norm_Jv2(v)) / normf     # lm_preduction
                vn = float(np.linalg.norm(diag * v)) / normf
                pred = un * un + 2.0 * mu * vn * vn
                rho = actual / pred if pred > 0.0 else -1.0
            if rho > 0.75:
                delta *= factor_up
            elif rho < 0.25:
                delta /= factor_down
            if rho > 0.0:
                fnorm2 = evaluate(x_trial)                # J <- J(x+dx), g
                res.njev += 1
                x = x_trial
                diag = _scale_update(scaler, lin.colnorm, diag)
                b = 2.0 * rho - 1.0                       # nielsen_accept
                mu *= max(0.333333333333333, 1.0 - b * b * b)
                nu = 2
                return GSL_SUCCESS
            mu *= nu                                      # nielsen_reject
            nu <<= 1
            bad_steps += 1
            if bad_steps > 15:
                return GSL_ENOPROG

    def test():
        if np.all(np.abs(dx) < xtol * xtol + xtol * np.abs(x)):     # test_delta
            return GSL_SUCCESS, 1
        gnorm = float(np.max(np.abs(np.maximum(x, 1.0) * lin.g))) if P else 0.0
        phi = 0.5 * fnorm2
        if gnorm <= gtol * max(phi, 1.0):
            return GSL_SUCCESS, 2
        return GSL_CONTINUE, 0

    # gsl_multifit_nlinear_driver
    it = 0
    niter = 0
    info = 0
    status = GSL_CONTINUE
    early = False
    while True:
        status = iterate()
        niter += 1
        if status == GSL_ENOPROG and it == 0:
            info = GSL_ENOPROG
            status = GSL_EMAXITER
            early = True
            break
        it += 1
        status, info = test()
        if not (status == GSL_CONTINUE and it < maxit):
            break
    if not early and it >= maxit and status != GSL_SUCCESS:
        status = GSL_EMAXITER

    res.x = x
    res.fnorm2 = fnorm2
    res.nit = niter
    res.status = status
    res.info = info
    res.mu = mu
    res.diag = diag
    # _gsl.pyx:686-701,:714-717
    res.error = None
    if status:
        res.error = (status, _STRERROR.get(status, 'gsl error %d' % status))
    if 0 <= info <= 3:
        res.stopping_criterion = info
    elif info == 27:
        res.stopping_criterion = 4
    else:
        res.stopping_criterion = 0
    if status == 11 and res.nit < maxit:
        res.error = "gsl_multifit can't improve on starting value; may have converged already."
    if info == 0 and res.error is None:
        res.error = "gsl_multifit didn't converge in {} iterations".format(maxit)
    return res


def gsl_multifit(x0, n, f, df, tol=(1e-5, 0.0, 0.0), maxit=1000, alg='lm',
                 solver='qr', scaler='more', factor_up=3.0, factor_down=2.0):
    """Counterpart of ``lsqfit.gsl_multifit`` with an explicit Jacobian callback
    ``df`` in place of the reference's GVar trick.  Returns an object with the
    attributes nonlinear_fit reads (__init__.py:665-679)."""
    if alg != 'lm':
        raise ValueError('oracle restates alg="lm" only (got %r)' % (alg,))
    tol = normalize_tol(tol)
    lin = _DenseLin(solver)
    cache = {}

    def evaluate(x):
        fv = np.asarray(f(x), float)
        if fv.shape != (n,):
            raise ValueError('fit function returned %s, expected (%d,)' % (fv.shape, n))
        J = np.asarray(df(x), float)
        lin.set(J, fv)
        return float(fv @ fv)

    def eval_fnorm2(x):
        fv = np.asarray(f(x), float)
        cache['f'] = fv
        return float(fv @ fv)

    res = _drive(x0, evaluate, eval_fnorm2, lin, tol, maxit, scaler, factor_up, factor_down)
    res.tol = tol
    res.f = lin.f
    res.J = lin.J
    res.cov = lin.covar()
    res.description = 'methods = {}/{}/{}'.format(alg, scaler, solver)
    res.results = None
    return res


def lm_normal(x0, normal_eq, chi2_fn, tol=(1e-5, 0.0, 0.0), maxit=1000,
              scaler='more', factor_up=3.0, factor_down=2.0):
    """Same driver fed by ``normal_eq(x) -> (A, g, chi2)`` and ``chi2_fn(x)``."""
    tol = normalize_tol(tol)
    lin = _NormalLin()

    def evaluate(x):
        A, g, c2 = normal_eq(x)
        lin.set(np.asarray(A, float), np.asarray(g, float))
        return float(c2)

    res = _drive(x0, evaluate, lambda x: float(chi2_fn(x)), lin, tol, maxit,
                 scaler, factor_up, factor_down)
    res.tol = tol
    res.A = lin.A
    res.g = lin.g
    res.cov = lin.covar()
    res.description = 'methods = lm/{}/cholesky'.format(scaler)
    res.results = None
    return res
