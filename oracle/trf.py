"""Trust Region Reflective least squares with box bounds (ORACLE ONLY).

What src/lsqfit/_scipy.py:115-181 obtains from ``scipy.optimize.least_squares``
with ``method='trf'`` (the plugin's default, :135-139) and ``bounds``
(flattened by the caller, src/lsqfit/__init__.py:641-655): the fit point, the
evaluation count reported as ``nit`` (:161), the covariance from the
thresholded SVD of the final Jacobian (:165-169) and the status map (:176-181).

scipy is a third-party dependency (``scipy>=1.13``, pyproject.toml:6) and is not
under /root/reference; the algorithm is restated from its published source
(scipy 1.15.3, optimize/_lsq/{least_squares,trf,common}.py: Branch, Coleman
and Li's reflective trust-region method with the exact (SVD) sub-problem
solver).  scipy itself IS importable in this image and on the GPU box, so this
restatement is pinned directly: tests/test_oracle_trf.py runs both on the same
problems and demands the same iterates (identical nfev, x to rounding).

  outer   v, dv = Coleman-Li scaling (distance to the bound the gradient points at);
          d = sqrt(v) * x_scale;  g_h = d g;  C = diag(g dv x_scale);
          stop when |g v|_inf < gtol
  inner   p_h = argmin of the quadratic model of (J d | sqrt C) inside |p_h| <= Delta
          (Gauss-Newton point when it is inside; else Newton on the secular equation
          |p_h(alpha)| = Delta, relative accuracy 0.01, at most 10 iterations, warm alpha);
          if x + d p_h leaves the box: best of {step cut back from the bound, step
          reflected at the bound, scaled gradient step}, all kept strictly inside by
          theta = max(0.995, 1 - |g v|_inf);
          ratio = actual / predicted reduction: < 0.25 shrinks Delta to |step_h|/4,
          > 0.75 on the boundary doubles it; ftol / xtol tests after every trial
  status  0 max_nfev, 1 gtol, 2 ftol, 3 xtol, 4 ftol and xtol

x_scale: 1.0 (scipy's default, what the reference passes on), an array, or 'jac'.
loss / f_scale (passed through by the reference, src/lsqfit/_scipy.py:76-79,:147-153): scipy's robust losses
rho(z), z = (f / f_scale)^2 -- 'linear', 'soft_l1', 'huber', 'cauchy', 'arctan' (optimize/_lsq/least_squares.py) --
enter through scale_for_robust_loss_function (optimize/_lsq/common.py): row i of J is multiplied by
sqrt(max(eps, rho' + 2 rho'' f_i^2)), f_i by rho' over that; cost = 0.5 f_scale^2 sum rho(z).
"""
import numpy as np

EPS = np.finfo(float).eps
LOSSES = ('linear', 'soft_l1', 'huber', 'cauchy', 'arctan')


def loss_rho(f, loss, f_scale, cost_only=False):
    """scipy's construct_loss_function: -> cost (cost_only) or rho = [rho, rho', rho''] of z = (f / f_scale)^2, the value
    scaled by f_scale^2 and the second derivative by 1 / f_scale^2."""
    z = (f / f_scale) ** 2
    if loss == 'linear':
        r0, r1, r2 = z, np.ones_like(z), np.zeros_like(z)
    elif loss == 'huber':
        big = z > 1
        r0 = np.where(big, 2 * np.sqrt(np.where(big, z, 1.0)) - 1, z)
        r1 = np.where(big, np.where(big, z, 1.0) ** -0.5, 1.0)
        r2 = np.where(big, -0.5 * np.where(big, z, 1.0) ** -1.5, 0.0)
    elif loss == 'soft_l1':
        t = 1 + z
        r0, r1, r2 = 2 * (t ** 0.5 - 1), t ** -0.5, -0.5 * t ** -1.5
    elif loss == 'cauchy':
        t = 1 + z
        r0, r1, r2 = np.log1p(z), 1 / t, -1 / t ** 2
    elif loss == 'arctan':
        t = 1 + z ** 2
        r0, r1, r2 = np.arctan(z), 1 / t, -2 * z / t ** 2
    else:
        raise ValueError('`loss` must be one of %s.' % (LOSSES,))
    if cost_only:
        return 0.5 * f_scale ** 2 * np.sum(r0)
    return np.array([r0 * f_scale ** 2, r1, r2 / f_scale ** 2])


def scale_for_loss(J, f, rho):
    """scipy's scale_for_robust_loss_function -> (J scaled, f scaled)."""
    js = rho[1] + 2 * rho[2] * f ** 2
    js[js < EPS] = EPS
    js **= 0.5
    return J * js[:, None], f * rho[1] / js


def cl_scaling(x, g, lb, ub):
    v = np.ones_like(x)
    dv = np.zeros_like(x)
    up = (g < 0) & np.isfinite(ub)
    v[up] = ub[up] - x[up]
    dv[up] = -1.0
    lo = (g > 0) & np.isfinite(lb)
    v[lo] = x[lo] - lb[lo]
    dv[lo] = 1.0
    return v, dv


def strictly_feasible(x, lb, ub, rstep):
    """Points on (rstep = 0) or within rstep of a bound are moved just inside it; a point
    within reach of both bounds belongs to the nearer one (the upper one on a tie)."""
    out = x.copy()
    if rstep == 0:
        lo, hi = x <= lb, x >= ub
        lo = lo & ~hi
        out[lo] = np.nextafter(lb[lo], ub[lo])
        out[hi] = np.nextafter(ub[hi], lb[hi])
    else:
        dlo, dhi = x - lb, ub - x
        lo = np.isfinite(lb) & (dlo <= np.minimum(dhi, rstep * np.maximum(1.0, np.abs(lb))))
        hi = np.isfinite(ub) & (dhi <= np.minimum(dlo, rstep * np.maximum(1.0, np.abs(ub))))
        lo = lo & ~hi
        out[lo] = lb[lo] + rstep * np.maximum(1.0, np.abs(lb[lo]))
        out[hi] = ub[hi] - rstep * np.maximum(1.0, np.abs(ub[hi]))
    tight = (out < lb) | (out > ub)
    out[tight] = 0.5 * (lb[tight] + ub[tight])
    return out


def to_bound(x, s, lb, ub):
    """Largest t with x + t s inside the box, and which coordinates stop it (+-1)."""
    steps = np.full(x.shape, np.inf)
    nz = s != 0
    with np.errstate(over='ignore', invalid='ignore'):
        steps[nz] = np.maximum((lb - x)[nz] / s[nz], (ub - x)[nz] / s[nz])
    t = np.min(steps)
    return t, (steps == t) * np.sign(s).astype(int)


def line_tr_exit(x, s, Delta):
    """Larger root t of |x + t s| = Delta for x inside the trust region."""
    a = s @ s
    b = x @ s
    c = x @ x - Delta ** 2
    disc = np.sqrt(b * b - a * c)
    q = -(b + np.copysign(disc, b))
    return max(q / a, c / q)


def quad_min_1d(a, b, lo, hi, c=0.0):
    """min over [lo, hi] of a t^2 + b t + c -> (t, value)."""
    ts = [lo, hi]
    if a != 0:
        t0 = -0.5 * b / a
        if lo < t0 < hi:
            ts.append(t0)
    ts = np.asarray(ts)
    ys = ts * (a * ts + b) + c
    k = int(np.argmin(ys))
    return ts[k], ys[k]


class _Model:
    """Quadratic model in the scaled variables: m(s) = g_h.s + (|J_h s|^2 + s.C s)/2."""

    def __init__(self, Jh, gh, C):
        self.Jh, self.gh, self.C = Jh, gh, C

    def curv(self, a, b):
        return (self.Jh @ a) @ (self.Jh @ b) + (a * self.C) @ b

    def value(self, s):
        return 0.5 * self.curv(s, s) + self.gh @ s

    def along(self, s, s0=None):
        """coefficients of t -> m(s0 + t s)"""
        a = 0.5 * self.curv(s, s)
        b = self.gh @ s
        if s0 is None:
            return a, b
        return a, b + self.curv(s0, s), self.value(s0)


def tr_subproblem(n, m, uf, s, V, Delta, alpha0):
    """min |J_aug p + f_aug| with |p| <= Delta from the SVD J_aug = U diag(s) V^T, uf = U^T f_aug.
    -> (p, alpha, Newton iterations)"""
    suf = s * uf
    full_rank = m >= n and s[-1] > EPS * m * s[0]
    if full_rank:
        p = -V @ (uf / s)
        if np.linalg.norm(p) <= Delta:
            return p, 0.0, 0

    def phi(alpha):
        den = s ** 2 + alpha
        pn = np.linalg.norm(suf / den)
        return pn - Delta, -np.sum(suf ** 2 / den ** 3) / pn

    hi = np.linalg.norm(suf) / Delta
    if full_rank:
        f0, d0 = phi(0.0)
        lo = -f0 / d0
    else:
        lo = 0.0
    restart = lambda: max(0.001 * hi, (lo * hi) ** 0.5)
    alpha = restart() if (not full_rank and alpha0 == 0) else alpha0
    it = 0
    for it in range(10):
        if alpha < lo or alpha > hi:
            alpha = restart()
        ph, dph = phi(alpha)
        if ph < 0:
            hi = alpha
        ratio = ph / dph
        lo = max(lo, alpha - ratio)
        alpha -= (ph + Delta) * ratio / Delta
        if abs(ph) < 0.01 * Delta:
            break
    p = -V @ (suf / (s ** 2 + alpha))
    p *= Delta / np.linalg.norm(p)
    return p, alpha, it + 1


def choose_step(x, M, p, p_h, d, Delta, lb, ub, theta):
    """-> (step, step_h, predicted reduction)"""
    xn = x + p
    if np.all((xn >= lb) & (xn <= ub)):
        return p, p_h, -M.value(p_h)
    t_hit, hits = to_bound(x, p, lb, ub)
    r_h = p_h.copy()
    r_h[hits.astype(bool)] *= -1.0           # reflected direction
    r = d * r_h
    p = p * t_hit
    p_h = p_h * t_hit
    x_hit = x + p
    t_tr = line_tr_exit(p_h, r_h, Delta)
    t_box, _ = to_bound(x_hit, r, lb, ub)
    t_r = min(t_box, t_tr)
    if t_r > 0:
        r_lo = (1 - theta) * t_hit / t_r
        r_hi = theta * t_box if t_r == t_box else t_tr
    else:
        r_lo, r_hi = 0.0, -1.0
    if r_lo <= r_hi:
        a, b, c = M.along(r_h, s0=p_h)
        t, r_value = quad_min_1d(a, b, r_lo, r_hi, c=c)
        r_h = p_h + t * r_h
        r = r_h * d
    else:
        r_value = np.inf
    p = p * theta                             # strictly interior
    p_h = p_h * theta
    p_value = M.value(p_h)
    ag_h = -M.gh
    ag = d * ag_h
    t_tr = Delta / np.linalg.norm(ag_h)
    t_box, _ = to_bound(x, ag, lb, ub)
    t_max = theta * t_box if t_box < t_tr else t_tr
    a, b = M.along(ag_h)
    t, ag_value = quad_min_1d(a, b, 0.0, t_max)
    ag_h = ag_h * t
    ag = ag * t
    if p_value < r_value and p_value < ag_value:
        return p, p_h, -p_value
    if r_value < p_value and r_value < ag_value:
        return r, r_h, -r_value
    return ag, ag_h, -ag_value


class TRFResult:
    pass


def trf(fun, jac, x0, bounds=None, xtol=1e-8, gtol=1e-8, ftol=1e-8, max_nfev=None, x_scale=1.0, loss='linear', f_scale=1.0):
    """Counterpart of ``least_squares(fun, x0, jac, bounds, 'trf', ftol, xtol, gtol, x_scale, loss, f_scale,
    max_nfev=...)`` for dense Jacobians."""
    robust = loss != 'linear'
    if loss not in LOSSES:
        raise ValueError('`loss` must be one of %s.' % (LOSSES,))
    x0 = np.atleast_1d(np.asarray(x0, float))
    n = x0.size
    if bounds is None:
        lb, ub = np.full(n, -np.inf), np.full(n, np.inf)
    else:
        lb = np.broadcast_to(np.asarray(bounds[0], float), (n,)).copy()
        ub = np.broadcast_to(np.asarray(bounds[1], float), (n,)).copy()
    if np.any(lb >= ub):
        raise ValueError('Each lower bound must be strictly less than each upper bound.')
    if not np.all((x0 >= lb) & (x0 <= ub)):
        raise ValueError('Initial guess is outside of provided bounds')
    if ftol < EPS and xtol < EPS and gtol < EPS:
        raise ValueError('At least one of the tolerances must be higher than machine epsilon')
    x = strictly_feasible(x0, lb, ub, 1e-10)
    f = np.atleast_1d(np.asarray(fun(x), float))
    if not np.all(np.isfinite(f)):
        raise ValueError('Residuals are not finite in the initial point.')
    nfev = 1
    J = np.atleast_2d(np.asarray(jac(x), float))
    njev = 1
    m = f.size
    f_true = f
    if robust:
        rho = loss_rho(f, loss, f_scale)
        cost = 0.5 * np.sum(rho[0])
        J, f = scale_for_loss(J, f, rho)
    else:
        cost = 0.5 * (f @ f)
    g = J.T @ f
    jac_scale = isinstance(x_scale, str) and x_scale == 'jac'

    def jscale(J, old=None):
        si = np.sum(J ** 2, axis=0) ** 0.5
        if old is None:
            si[si == 0] = 1.0
        else:
            si = np.maximum(si, old)
        return 1.0 / si, si

    if jac_scale:
        scale, scale_inv = jscale(J)
    else:
        scale = np.broadcast_to(np.asarray(x_scale, float), (n,)).copy()
        scale_inv = 1.0 / scale
    v, dv = cl_scaling(x, g, lb, ub)
    v[dv != 0] *= scale_inv[dv != 0]
    Delta = np.linalg.norm(x * scale_inv / v ** 0.5)
    if Delta == 0:
        Delta = 1.0
    if max_nfev is None:
        max_nfev = 100 * n
    alpha = 0.0
    status = None
    g_norm = None
    f_aug = np.zeros(m + n)
    J_aug = np.empty((m + n, n))
    while True:
        v, dv = cl_scaling(x, g, lb, ub)
        g_norm = np.linalg.norm(g * v, ord=np.inf)
        if g_norm < gtol:
            status = 1
        if status is not None or nfev == max_nfev:
            break
        v[dv != 0] *= scale_inv[dv != 0]
        d = v ** 0.5 * scale
        C = g * dv * scale
        g_h = d * g
        f_aug[:m] = f
        J_aug[:m] = J * d
        J_aug[m:] = np.diag(C ** 0.5)
        U, s, Vt = np.linalg.svd(J_aug, full_matrices=False)
        uf = U.T @ f_aug
        M = _Model(J_aug[:m], g_h, C)
        theta = max(0.995, 1 - g_norm)
        actual = -1.0
        while actual <= 0 and nfev < max_nfev:
            p_h, alpha, _ = tr_subproblem(n, m, uf, s, Vt.T, Delta, alpha)
            step, step_h, predicted = choose_step(x, M, d * p_h, p_h, d, Delta, lb, ub, theta)
            x_new = strictly_feasible(x + step, lb, ub, 0)
            f_new = np.atleast_1d(np.asarray(fun(x_new), float))
            nfev += 1
            sh_norm = np.linalg.norm(step_h)
            if not np.all(np.isfinite(f_new)):
                Delta = 0.25 * sh_norm
                continue
            cost_new = loss_rho(f_new, loss, f_scale, cost_only=True) if robust else 0.5 * (f_new @ f_new)
            actual = cost - cost_new
            if predicted > 0:
                ratio = actual / predicted
            elif predicted == actual == 0:
                ratio = 1.0
            else:
                ratio = 0.0
            Delta_new = Delta
            if ratio < 0.25:
                Delta_new = 0.25 * sh_norm
            elif ratio > 0.75 and sh_norm > 0.95 * Delta:
                Delta_new = 2.0 * Delta
            f_ok = actual < ftol * cost and ratio > 0.25
            x_ok = np.linalg.norm(step) < xtol * (xtol + np.linalg.norm(x))
            status = 4 if (f_ok and x_ok) else 2 if f_ok else 3 if x_ok else None
            if status is not None:
                break
            alpha *= Delta / Delta_new
            Delta = Delta_new
        if actual > 0:
            x, f, cost = x_new, f_new, cost_new
            f_true = f
            J = np.atleast_2d(np.asarray(jac(x), float))
            njev += 1
            if robust:
                J, f = scale_for_loss(J, f, loss_rho(f, loss, f_scale))
            g = J.T @ f
            if jac_scale:
                scale, scale_inv = jscale(J, scale_inv)
    res = TRFResult()
    res.x, res.cost, res.fun, res.jac, res.grad = x, cost, f_true, J, g
    res.optimality, res.nfev, res.njev = g_norm, nfev, njev
    res.status = 0 if status is None else status
    return res


def _dogleg_in_box(x, newton, g, a, b, tr, lb, ub):
    """Dogleg step inside the intersection of the box with the rectangular trust region |s_i| <= tr_i.
    -> (step, which original bound each coordinate lands on (-1, 0, 1), trust region hit?)"""
    lo_c, hi_c = lb - x, ub - x
    lo_t, hi_t = np.maximum(lo_c, -tr), np.minimum(hi_c, tr)
    lands = np.zeros(x.shape, dtype=int)
    if np.all((newton >= lo_t) & (newton <= hi_t)):
        return newton, lands, False
    zero = np.zeros_like(x)
    t_max, _ = to_bound(zero, -g, lo_t, hi_t)
    t, _ = quad_min_1d(a, b, 0.0, t_max)
    cauchy = -t * g
    diff = newton - cauchy
    t, hits = to_bound(cauchy, diff, lo_t, hi_t)
    lands[(hits < 0) & (lo_t == lo_c)] = -1
    lands[(hits > 0) & (hi_t == hi_c)] = 1
    tr_hit = bool(np.any(((hits < 0) & (lo_t == -tr)) | ((hits > 0) & (hi_t == tr))))
    return cauchy + t * diff, lands, tr_hit


def dogbox(fun, jac, x0, bounds=None, xtol=1e-8, gtol=1e-8, ftol=1e-8, max_nfev=None, x_scale=1.0, loss='linear', f_scale=1.0):
    """Counterpart of ``least_squares(..., method='dogbox')`` (dense Jacobian): dogleg
    steps in a rectangular trust region, variables that reach a wall with the gradient pushing
    outwards leave the active problem (optimize/_lsq/dogbox.py)."""
    x0 = np.atleast_1d(np.asarray(x0, float))
    n = x0.size
    if bounds is None:
        lb, ub = np.full(n, -np.inf), np.full(n, np.inf)
    else:
        lb = np.broadcast_to(np.asarray(bounds[0], float), (n,)).copy()
        ub = np.broadcast_to(np.asarray(bounds[1], float), (n,)).copy()
    if np.any(lb >= ub):
        raise ValueError('Each lower bound must be strictly less than each upper bound.')
    if not np.all((x0 >= lb) & (x0 <= ub)):
        raise ValueError('Initial guess is outside of provided bounds')
    if ftol < EPS and xtol < EPS and gtol < EPS:
        raise ValueError('At least one of the tolerances must be higher than machine epsilon')
    x = x0.copy()
    f = np.atleast_1d(np.asarray(fun(x), float))
    if not np.all(np.isfinite(f)):
        raise ValueError('Residuals are not finite in the initial point.')
    nfev = njev = 1
    J = np.atleast_2d(np.asarray(jac(x), float))
    robust = loss != 'linear'
    if loss not in LOSSES:
        raise ValueError('`loss` must be one of %s.' % (LOSSES,))
    f_true = f
    if robust:
        rho = loss_rho(f, loss, f_scale)
        cost = 0.5 * np.sum(rho[0])
        J, f = scale_for_loss(J, f, rho)
    else:
        cost = 0.5 * (f @ f)
    g = J.T @ f
    jac_scale = isinstance(x_scale, str) and x_scale == 'jac'

    def jscale(J, old=None):
        si = np.sum(J ** 2, axis=0) ** 0.5
        if old is None:
            si[si == 0] = 1.0
        else:
            si = np.maximum(si, old)
        return 1.0 / si, si

    if jac_scale:
        scale, scale_inv = jscale(J)
    else:
        scale = np.broadcast_to(np.asarray(x_scale, float), (n,)).copy()
        scale_inv = 1.0 / scale
    Delta = np.linalg.norm(x0 * scale_inv, ord=np.inf)
    if Delta == 0:
        Delta = 1.0
    on_bound = np.zeros(n, dtype=int)
    on_bound[x0 == lb] = -1
    on_bound[x0 == ub] = 1
    if max_nfev is None:
        max_nfev = 100 * n
    status = None
    g_norm = None
    while True:
        active = on_bound * g < 0
        free = ~active
        g_full = g.copy()
        g = g.copy()
        g[active] = 0.0
        g_norm = np.linalg.norm(g, ord=np.inf)
        if g_norm < gtol:
            status = 1
        if status is not None or nfev == max_nfev:
            break
        Jf, gf = J[:, free], g[free]
        newton = np.linalg.lstsq(Jf, -f, rcond=-1)[0]
        v = Jf @ gf
        a, b = 0.5 * (v @ v), -(gf @ gf)           # model along -g_free
        actual = -1.0
        while actual <= 0 and nfev < max_nfev:
            sf, lands, tr_hit = _dogleg_in_box(x[free], newton, gf, a, b, Delta * scale[free], lb[free], ub[free])
            step = np.zeros(n)
            step[free] = sf
            Js = Jf @ sf
            predicted = -(0.5 * (Js @ Js) + gf @ sf)
            x_new = np.clip(x + step, lb, ub)
            f_new = np.atleast_1d(np.asarray(fun(x_new), float))
            nfev += 1
            sh_norm = np.linalg.norm(step * scale_inv, ord=np.inf)
            if not np.all(np.isfinite(f_new)):
                Delta = 0.25 * sh_norm
                continue
            cost_new = loss_rho(f_new, loss, f_scale, cost_only=True) if robust else 0.5 * (f_new @ f_new)
            actual = cost - cost_new
            if predicted > 0:
                ratio = actual / predicted
            elif predicted == actual == 0:
                ratio = 1.0
            else:
                ratio = 0.0
            if ratio < 0.25:
                Delta = 0.25 * sh_norm
            elif ratio > 0.75 and tr_hit:
                Delta *= 2.0
            f_ok = actual < ftol * cost and ratio > 0.25
            x_ok = np.linalg.norm(step) < xtol * (xtol + np.linalg.norm(x))
            status = 4 if (f_ok and x_ok) else 2 if f_ok else 3 if x_ok else None
            if status is not None:
                break
        if actual > 0:
            on_bound[free] = lands
            x = x_new
            x[on_bound == -1] = lb[on_bound == -1]
            x[on_bound == 1] = ub[on_bound == 1]
            f, cost = f_new, cost_new
            f_true = f
            J = np.atleast_2d(np.asarray(jac(x), float))
            njev += 1
            if robust:
                J, f = scale_for_loss(J, f, loss_rho(f, loss, f_scale))
            g = J.T @ f
            if jac_scale:
                scale, scale_inv = jscale(J, scale_inv)
        else:
            g = g_full
    res = TRFResult()
    res.x, res.cost, res.fun, res.jac, res.grad = x, cost, f_true, J, g_full
    res.optimality, res.nfev, res.njev = g_norm, nfev, njev
    res.status = 0 if status is None else status
    res.active_mask = on_bound
    return res


def scipy_least_squares(x0, n, f, df, tol=(1e-8, 1e-8, 1e-8), maxit=1000, method=None, bounds=None,
                        x_scale=1.0, loss='linear', f_scale=1.0):
    """Counterpart of ``lsqfit.scipy_least_squares`` (src/lsqfit/_scipy.py:115-181) for
    method 'trf', with an explicit Jacobian callback in place of the GVar trick."""
    from .lm import normalize_tol
    if method not in (None, 'trf', 'dogbox', 'lm'):
        raise ValueError("`method` must be 'trf', 'dogbox' or 'lm'.")
    tol = normalize_tol(tol)
    if method == 'lm':
        from .minpack import least_squares_lm
        if bounds is not None and not (np.all(np.isneginf(bounds[0])) and np.all(np.isposinf(bounds[1]))):
            raise ValueError("Method 'lm' doesn't support bounds.")
        if loss != 'linear':
            raise ValueError("method='lm' supports only 'linear' loss function.")
        fit = least_squares_lm(f, df, x0, xtol=tol[0], gtol=tol[1], ftol=tol[2], max_nfev=maxit, x_scale=x_scale)
    else:
        fit = (dogbox if method == 'dogbox' else trf)(f, df, x0, bounds=bounds, xtol=tol[0], gtol=tol[1],
                                                      ftol=tol[2], max_nfev=maxit, x_scale=x_scale, loss=loss, f_scale=f_scale)
    res = TRFResult()
    res.tol = tol
    res.description = 'method = {}'.format('trf' if method is None else method)      # :134-139
    res.x = fit.x
    res.f = np.asarray(f(res.x), float)
    res.J = np.asarray(df(res.x), float)
    res.nit = fit.nfev
    res.results = fit
    _, s, Vt = np.linalg.svd(fit.jac, full_matrices=False)          # :165-169
    keep = s > EPS * max(fit.jac.shape) * s[0]
    s, Vt = s[keep], Vt[keep]
    res.cov = (Vt.T / s ** 2) @ Vt
    res.error = None
    res.stopping_criterion = {0: 0, 1: 2, 2: 3, 3: 1, 4: 1}[fit.status]  # :176-181
    return res
