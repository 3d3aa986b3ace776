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

Other trust-region sub-problem solvers (``alg`` = lmaccel / dogleg / ddogleg /
subspace2D, _gsl.pyx:622-635), restated from
multifit_nlinear/{lm,dogleg,subspace2D,fdfvv}.c:

  lmaccel    velocity v as in lm; fvv = (2/h)((F(x+hv)-F(x))/h - J v), h = h_fvv = 0.02
             (no analytic fvv is ever supplied: fdf.fvv = NULL, _gsl.pyx:662);
             acceleration a: (J^T J + mu D^2) a = -J^T fvv; dx = v + a/2;
             the step is rejected when |a|/|v| > avmax (trust_eval_step); the predicted
             reduction is lm's, evaluated on dx.
  dogleg     dx_sd = -alpha D^-2 g, alpha = |D^-1 g|^2/|J D^-2 g|^2; dx_gn from mu = 0;
             |D dx_sd| >= delta: dx_sd scaled to the boundary; |D dx_gn| <= delta: dx_gn;
             else dx_sd + beta (dx_gn - dx_sd) on the boundary.
  ddogleg    as dogleg with the Gauss-Newton leg shortened by t = 1 - 0.8 (1 - c),
             c = |D^-1 g|^4 / (|J D^-2 g|^2 |g^T dx_gn|); t |D dx_gn| <= delta:
             dx_gn scaled to the boundary.
  subspace2D exact minimiser of the quadratic model over span(D dx_sd, D dx_gn) on the
             trust-region boundary (GSL: roots of a quartic; here the equivalent secular
             equation of the 2x2 problem); dx_gn inside: dx_gn; parallel legs: dx_sd scaled.
  dogleg-family predicted reduction: -(|J dx|^2 + 2 g.dx)/|f|^2; delta is updated by
  factor_up/factor_down exactly as for lm (where it is inert).

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

    def step_subset(self, mask):
        """argmin over the masked parameters alone of |J da + f|^2 (the others held)"""
        return -np.linalg.lstsq(self.J[:, mask], self.f, rcond=None)[0]

    def norm_Jv2(self, v):
        w = self.J @ v
        return float(w @ w)

    def step_rhs(self, mu, diag, fvec):
        """min |J a + fvec|^2 + mu |D a|^2 (the acceleration solve of lm_step)."""
        keep_f, keep_g = self.f, self.g
        self.f, self.g = fvec, self.J.T @ fvec
        try:
            return self.step(mu, diag)
        finally:
            self.f, self.g = keep_f, keep_g

    def fvv(self, x, v, h, eval_fvec):
        """fdfvv.c: finite-difference second directional derivative."""
        fp = eval_fvec(x + h * v)
        return (2.0 / h) * ((fp - self.f) / h - self.J @ v)

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

    def step_subset(self, mask):
        return -np.linalg.solve(self.A[np.ix_(mask, mask)], self.g[mask])

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


def _solve_tr_2d(B, g, delta):
    """argmin g.q + q.B.q/2 subject to |q| = delta for a 2x2 positive semi-definite B with
    the unconstrained minimiser outside the ball: q = -(B + lam I)^-1 g, lam > 0 from the
    secular equation (bisection + Newton on 1/|q(lam)|)."""
    w, V = np.linalg.eigh(B)
    gt = V.T @ g

    def qnorm(lam):
        return float(np.sqrt(np.sum((gt / (w + lam)) ** 2)))
    lo, hi = 0.0, max(1.0, float(np.linalg.norm(g)) / delta)
    while qnorm(hi) > delta:
        hi *= 2.0
    for _ in range(200):
        mid = 0.5 * (lo + hi)
        if qnorm(mid) > delta:
            lo = mid
        else:
            hi = mid
        if hi - lo <= 1e-16 * hi:
            break
    lam = 0.5 * (lo + hi)
    return V @ (-gt / (w + lam))


TRACE = None      # a list: _drive appends one record per trial step (developer tool)


def _drive(x0, evaluate, eval_fnorm2, lin, tol, maxit, scaler, factor_up, factor_down,
           alg='lm', avmax=0.75, h_fvv=0.02, eval_fvec=None, undamped=None, stop=None):
    """Shared trust.c/fdf.c logic.  ``evaluate(x)`` refreshes ``lin`` with the
    Jacobian-level quantities at x and returns |f|^2; ``eval_fnorm2(x)`` is the
    cheap trial evaluation.  ``stop()`` (bench.py's time box), asked after every iteration,
    ends the run as ``maxit`` would."""
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
    lin_mask = None
    if undamped is not None and np.any(undamped):
        # nonlinear_fit's ``linear=`` (src/lsqfit/__init__.py:738-787, _varpro_fit): variable
        # projection.  The residual is linear in the masked parameters a, so the fitter works on
        # phi(theta) = min_a chi2(a, theta): EVERY evaluation first solves A_aa da = -g_a exactly
        # (the reference: lstsq inside the wrapped fit function), and the step in theta is the LM
        # step of the projected functional, (S + mu D_theta^2) dtheta = -g_theta with the Schur
        # complement S = A_tt - A_ta A_aa^-1 A_at -- obtained by solving the full system with
        # the a-block of the damping matrix set to zero (Kaufman's form of the Golub-Pereyra
        # Jacobian; the reference differentiates through lstsq and keeps the second term too:
        # same minimum, slightly different iterates).
        if alg != 'lm':
            raise ValueError("linear parameters need alg='lm'")
        lin_mask = np.asarray(undamped, bool)

        def project(xx, f2):
            """-> (x with a re-solved, chi2 there); lin must hold the evaluation at xx"""
            da = lin.step_subset(lin_mask)
            xx = xx.copy()
            xx[lin_mask] += da
            return xx, f2 + float(lin.g[lin_mask] @ da)

        x, _ = project(x, fnorm2)
        fnorm2 = evaluate(x)
        res.nfev += 1
        res.njev += 1
    diag = _scale_init(scaler, lin.colnorm)
    if lin_mask is not None:
        diag = np.where(lin_mask, 0.0, diag)
    damped = diag > 0
    mu = 1e-3 * float(np.max(lin.colnorm[damped] / diag[damped])) ** 2 if np.any(damped) else 0.0   # nielsen_init
    nu = 2
    delta = 0.3 * max(1.0, float(np.linalg.norm(diag * x)))
    dx = np.zeros(P)

    if alg not in ('lm', 'lmaccel', 'dogleg', 'ddogleg', 'subspace2D'):
        raise ValueError('unkown algorithm ' + str(alg))
    if alg == 'lmaccel' and (eval_fvec is None or not hasattr(lin, 'fvv')):
        raise ValueError('lmaccel needs residual-level access (dense driver only)')
    leg = {}

    def dogleg_preloop():
        # dogleg_preloop / subspace2D_preloop: steepest-descent leg; Gauss-Newton leg lazily
        g = lin.g
        w1 = g / diag
        leg['norm_Dinvg'] = float(np.linalg.norm(w1))
        w2 = w1 / diag
        leg['norm_JDinv2g'] = float(np.sqrt(lin.norm_Jv2(w2)))
        uu = leg['norm_Dinvg'] / leg['norm_JDinv2g']
        leg['dx_sd'] = -(uu * uu) * w2
        leg['norm_Dsd'] = float(np.linalg.norm(diag * leg['dx_sd']))
        leg['norm_Dgn'] = -1.0
        leg.pop('sub', None)

    def calc_gn():
        if leg['norm_Dgn'] < 0.0:
            leg['dx_gn'] = lin.step(0.0, diag)
            leg['norm_Dgn'] = float(np.linalg.norm(diag * leg['dx_gn']))

    def dogleg_beta(t):
        w = t * leg['dx_gn'] - leg['dx_sd']
        a = float(np.sum((diag * w) ** 2))
        b = 2.0 * float(leg['dx_sd'] @ (diag * diag * w))
        c = (leg['norm_Dsd'] + delta) * (leg['norm_Dsd'] - delta)
        disc = np.sqrt(b * b - 4.0 * a * c)
        return (-2.0 * c) / (b + disc) if b > 0.0 else (-b + disc) / (2.0 * a)

    def trs_step():
        """-> (dx, avratio)"""
        if alg in ('lm', 'lmaccel'):
            v = lin.step(mu, diag)                        # lm_step
            if alg == 'lm':
                return v, 0.0
            a = lin.step_rhs(mu, diag, lin.fvv(x, v, h_fvv, eval_fvec))
            res.nfev += 1
            return v + 0.5 * a, float(np.linalg.norm(a) / np.linalg.norm(v))
        if alg == 'subspace2D':
            calc_gn()
            if leg['norm_Dgn'] <= delta:
                return leg['dx_gn'], 0.0
            if 'sub' not in leg:
                W = np.column_stack([diag * leg['dx_sd'] / leg['norm_Dsd'],
                                     diag * leg['dx_gn'] / leg['norm_Dgn']])
                Q, R, _ = sla.qr(W, mode='economic', pivoting=True)
                rank = int(np.sum(np.abs(np.diag(R)) > np.finfo(float).eps * 2 * abs(R[0, 0]))) if R[0, 0] != 0 else 0
                if rank == 2:
                    DQ = Q / diag[:, None]
                    subB = np.array([[lin_quad(DQ[:, i], DQ[:, j]) for j in range(2)] for i in range(2)])
                    leg['sub'] = (Q, DQ.T @ lin.g, subB)
                else:
                    leg['sub'] = None
            if leg['sub'] is None:
                return leg['dx_sd'] * (delta / leg['norm_Dsd']), 0.0
            Q, subg, subB = leg['sub']
            return (Q @ _solve_tr_2d(subB, subg, delta)) / diag, 0.0
        # dogleg / ddogleg
        if leg['norm_Dsd'] >= delta:
            return leg['dx_sd'] * (delta / leg['norm_Dsd']), 0.0
        calc_gn()
        if leg['norm_Dgn'] <= delta:
            return leg['dx_gn'], 0.0
        t = 1.0
        if alg == 'ddogleg':
            uu = (leg['norm_Dinvg'] / leg['norm_JDinv2g']) ** 2
            gd = float(lin.g @ leg['dx_gn'])
            c = uu * (leg['norm_Dinvg'] / abs(gd)) * leg['norm_Dinvg']
            t = 1.0 - 0.8 * (1.0 - c)
            if t * leg['norm_Dgn'] <= delta:
                return leg['dx_gn'] * (delta / leg['norm_Dgn']), 0.0
        beta = dogleg_beta(t)
        return leg['dx_sd'] + beta * (t * leg['dx_gn'] - leg['dx_sd']), 0.0

    def lin_quad(a, b):
        # a^T J^T J b from the norm interface (polarisation)
        if a is b:
            return lin.norm_Jv2(a)
        return 0.25 * (lin.norm_Jv2(a + b) - lin.norm_Jv2(a - b))

    def preduction(step, normf):
        if alg in ('lm', 'lmaccel'):
            un = np.sqrt(lin.norm_Jv2(step)) / normf      # lm_preduction
            vn = float(np.linalg.norm(diag * step)) / normf
            return un * un + 2.0 * mu * vn * vn
        # quadratic_preduction
        return -(lin.norm_Jv2(step) + 2.0 * float(lin.g @ step)) / (normf * normf)

    def iterate():
        nonlocal x, fnorm2, diag, mu, nu, delta, dx
        bad_steps = 0
        if alg in ('dogleg', 'ddogleg', 'subspace2D'):
            dogleg_preloop()
        while True:
            dx, avratio = trs_step()
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
                pred = preduction(dx, normf)
                rho = actual / pred if pred > 0.0 else -1.0
            if TRACE is not None:     # (developer hook, tools/trace_cosmix.py: one entry per trial step)
                TRACE.append(dict(it=res.njev, mu=float(mu), rho=float(rho), chi2=float(fnorm2), chi2_trial=float(ft2),
                                  Ddx=float(np.linalg.norm(diag * dx)), dx=float(np.linalg.norm(dx))))
            if rho > 0.75:
                delta *= factor_up
            elif rho < 0.25:
                delta /= factor_down
            # trust_eval_step: geodesic acceleration must stay small next to the velocity
            if rho > 0.0 and not (alg == 'lmaccel' and avratio > avmax):
                fnorm2 = evaluate(x_trial)                # J <- J(x+dx), g
                res.njev += 1
                x = x_trial
                diag = _scale_update(scaler, lin.colnorm, diag)
                b = 2.0 * rho - 1.0                       # nielsen_accept
                mu *= max(0.333333333333333, 1.0 - b * b * b)
                nu = 2
                return GSL_SUCCESS
            mu = float(mu) * nu                           # nielsen_reject
            nu <<= 1
            bad_steps += 1
            if bad_steps > 15:
                return GSL_ENOPROG

    def iterate_varpro():
        nonlocal x, fnorm2, diag, mu, nu, dx
        bad_steps = 0
        while True:
            v = lin.step(mu, diag)                        # a-block undamped: Schur-complement step
            normf = np.sqrt(fnorm2)
            pred = preduction(v, normf)
            keep = dict(lin.__dict__)
            with np.errstate(all='ignore'):
                ft2 = evaluate(x + v)                     # full evaluation at the trial point ...
            res.nfev += 1
            res.njev += 1
            res.ntrial += 1
            finite = np.isfinite(ft2) and np.all(np.isfinite(lin.g)) and np.all(np.isfinite(lin.colnorm))
            x_trial = x + v
            if finite:
                x_trial, ft2 = project(x_trial, ft2)      # ... then the exact linear solve there
            dx = x_trial - x                              # what the convergence test sees (trial steps too)
            normf_trial = np.sqrt(max(ft2, 0.0)) if finite else np.inf
            rho = -1.0
            if normf_trial < normf:
                u = normf_trial / normf
                rho = (1.0 - u * u) / pred if pred > 0.0 else -1.0
            if rho > 0.0:
                fnorm2 = evaluate(x_trial)                # J, g at the projected point
                res.nfev += 1
                res.njev += 1
                x = x_trial
                diag = np.where(lin_mask, 0.0, _scale_update(scaler, lin.colnorm, np.where(lin_mask, 1.0, diag)))
                b = 2.0 * rho - 1.0
                mu *= max(0.333333333333333, 1.0 - b * b * b)
                nu = 2
                return GSL_SUCCESS
            lin.__dict__.update(keep)                     # back to the current point's J, g
            mu = float(mu) * nu
            nu <<= 1
            bad_steps += 1
            if bad_steps > 15:
                return GSL_ENOPROG

    if lin_mask is not None:
        iterate = iterate_varpro

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
        if not (status == GSL_CONTINUE and it < maxit) or (stop is not None and stop()):
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
                 solver='qr', scaler='more', factor_up=3.0, factor_down=2.0, avmax=0.75, undamped=None):
    """Counterpart of ``lsqfit.gsl_multifit`` with an explicit Jacobian callback
    ``df`` in place of the reference's GVar trick.  Returns an object with the
    attributes nonlinear_fit reads (__init__.py:665-679)."""
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

    res = _drive(x0, evaluate, eval_fnorm2, lin, tol, maxit, scaler, factor_up, factor_down,
                 alg=alg, avmax=avmax, eval_fvec=lambda xx: np.asarray(f(xx), float), undamped=undamped)
    res.tol = tol
    res.f = lin.f
    res.J = lin.J
    res.cov = lin.covar()
    res.description = 'methods = {}/{}/{}'.format(alg, scaler, solver)
    if alg == 'lmaccel':
        res.description += '    avmax = {}'.format(avmax)        # _gsl.pyx:617-618
    res.results = None
    return res


def lm_normal(x0, normal_eq, chi2_fn, tol=(1e-5, 0.0, 0.0), maxit=1000,
              scaler='more', factor_up=3.0, factor_down=2.0, alg='lm', stop=None, lin=None):
    """Same driver fed by ``normal_eq(x) -> (A, g, chi2)`` and ``chi2_fn(x)``.  ``lin``: a ``_NormalLin`` (or a subclass: the
    bench's CPU baseline times its factorisations) -- same algebra."""
    tol = normalize_tol(tol)
    lin = _NormalLin() if lin is None else lin

    def evaluate(x):
        A, g, c2 = normal_eq(x)
        lin.set(np.asarray(A, float), np.asarray(g, float))
        return float(c2)

    res = _drive(x0, evaluate, lambda x: float(chi2_fn(x)), lin, tol, maxit,
                 scaler, factor_up, factor_down, alg=alg, stop=stop)
    res.tol = tol
    res.A = lin.A
    res.g = lin.g
    res.cov = lin.covar()
    res.description = 'methods = {}/{}/cholesky'.format(alg, scaler)
    res.results = None
    return res
