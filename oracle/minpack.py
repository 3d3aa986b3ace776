"""MINPACK's Levenberg-Marquardt driver ``lmder`` (ORACLE ONLY).

What src/lsqfit/_scipy.py:115-181 obtains from ``scipy.optimize.least_squares``
with ``method='lm'`` (:64-67; tests/test_lsqfit.py:1768-1772,:1824-1826):
scipy hands the problem to MINPACK's ``lmder`` with ``factor = 100``,
``diag = 1/x_scale`` (``mode = 2``) or MINPACK's own column-norm scaling for
``x_scale='jac'`` (``mode = 1``), and maps MINPACK's ``info`` 1, 2, 3, 4, 5 to its
status 2, 3, 4, 1, 0 (scipy optimize/_lsq/least_squares.py ``call_minpack``).
Bounds are not supported by this method.

MINPACK is compiled into scipy (third party, ``scipy>=1.13``, pyproject.toml:6);
its algorithm is restated here from the published Fortran (lmder.f, lmpar.f;
More, Garbow, Hillstrom 1980) with dense linear algebra in place of the
pivoted-QR bookkeeping: ``qrfac/qrsolv`` only ever deliver the least-squares
solution of ``[J; sqrt(par) D] x = [f; 0]`` and triangular solves with its R
factor (both functions of ``A = J^T J``, ``g = J^T f`` and ``D``, which is the form the
device uses; the oracle keeps an unpivoted QR of the stacked matrix).
scipy itself IS importable in this image and on the GPU box, so the restatement is
pinned directly on it (tests/test_oracle_trf.py: same nfev, njev, status, x).

  outer   J, column norms; first pass: D (mode 1), delta = factor |D x| (factor if 0);
          gnorm = max_j |g_j| / (|J_j| |f|) <= gtol -> info 4;  D = max(D, |J_j|) (mode 1)
  inner   lmpar: p = (A + par D^2)^-1 g with |D p| within 10 % of delta (par = 0 when the
          Gauss-Newton step already is), at most 10 Newton steps on the secular equation
          with More's bounds parl / paru;  x_t = x - p;  first pass: delta = min(delta, |D p|)
          actred = 1 - (|f_t|/|f|)^2 (-1 if |f_t| >= 10 |f|);
          prered = (|J p|^2 + 2 par |D p|^2)/|f|^2;  dirder = -(|J p|^2 + par |D p|^2)/|f|^2
          ratio <= 1/4: delta = temp min(delta, 10 |D p|), par /= temp  (temp from actred, dirder)
          par = 0 or ratio >= 3/4: delta = 2 |D p|, par /= 2
          ratio >= 1e-4: accept;  info 1 (ftol) / 2 (xtol: delta <= xtol |D x|) / 3 (both);
          info 5 (maxfev) / 6, 7, 8 (tolerances below machine precision); repeat while ratio < 1e-4
"""
import numpy as np

EPSMCH = np.finfo(float).eps
DWARF = np.finfo(float).tiny


class _Normal:
    """x(par) = argmin |J x - f|^2 + par |D x|^2 and the quadratic forms lmpar needs, from a QR
    factorisation of the stacked matrix (no squaring of the condition number, like MINPACK)."""

    def __init__(self, J, f):
        self.J, self.f = J, f
        self.g = J.T @ f

    def solve(self, par, D):
        """-> (x, q -> q.(J^T J + par D^2)^-1 q) ; x is None when the matrix is singular"""
        n = self.J.shape[1]
        if par == 0:
            M, rhs = self.J, self.f
        else:
            M = np.vstack([self.J, np.sqrt(par) * np.diag(D)])
            rhs = np.concatenate([self.f, np.zeros(n)])
        Q, R = np.linalg.qr(M)
        dr = np.abs(np.diag(R))
        if M.shape[0] < n or dr.min() == 0 or dr.min() <= EPSMCH * dr.max():
            return None, None
        x = np.linalg.solve(R, Q.T @ rhs)
        return x, (lambda q: float(np.sum(np.linalg.solve(R.T, q) ** 2)))


def lmpar(nm, D, delta, par):
    """-> (par, x) with |D x| ~ delta (lmpar.f)."""
    x, form = nm.solve(0.0, D)
    full_rank = x is not None
    if full_rank:
        dxnorm = np.linalg.norm(D * x)
        fp = dxnorm - delta
        if fp <= 0.1 * delta:
            return 0.0, x
        q = D * (D * x) / dxnorm
        parl = (fp / delta) / form(q)
    else:
        # lmpar.f takes a truncated least-squares direction here; with a singular A the normal
        # equations offer none, so the search starts from the upper bound instead
        dxnorm, fp, parl = np.inf, np.inf, 0.0
    gnorm = np.linalg.norm(nm.g / D)
    paru = gnorm / delta
    if paru == 0:
        paru = DWARF / min(delta, 0.1)
    par = min(max(par, parl), paru)
    if par == 0:
        par = gnorm / dxnorm
    for it in range(1, 11):
        if par == 0:
            par = max(DWARF, 0.001 * paru)
        x, form = nm.solve(par, D)
        dxnorm = np.linalg.norm(D * x)
        prev, fp = fp, dxnorm - delta
        if abs(fp) <= 0.1 * delta or (parl == 0 and fp <= prev and prev < 0) or it == 10:
            break
        q = D * (D * x) / dxnorm
        parc = (fp / delta) / form(q)
        if fp > 0:
            parl = max(parl, par)
        if fp < 0:
            paru = min(paru, par)
        par = max(parl, par + parc)
    return par, x


class LMDerResult:
    pass


def lmder(fun, jac, x0, ftol=1e-8, xtol=1e-8, gtol=1e-8, maxfev=None, diag=None, factor=100.0):
    """-> result with x, fvec, nfev, njev, info (MINPACK numbering).  diag None: mode 1."""
    x = np.atleast_1d(np.asarray(x0, float)).copy()
    n = x.size
    mode = 1 if diag is None else 2
    D = np.ones(n) if diag is None else np.asarray(diag, float).copy()
    if maxfev is None:
        maxfev = 100 * (n + 1)
    res = LMDerResult()
    info = 0
    if n <= 0 or ftol < 0 or xtol < 0 or gtol < 0 or maxfev <= 0 or factor <= 0 or np.any(D <= 0):
        res.x, res.fvec, res.nfev, res.njev, res.info = x, None, 0, 0, 0
        return res
    f = np.atleast_1d(np.asarray(fun(x), float))
    nfev, njev = 1, 0
    fnorm = np.linalg.norm(f)
    par = 0.0
    it = 1
    delta = xnorm = 0.0
    while True:
        J = np.atleast_2d(np.asarray(jac(x), float))
        njev += 1
        nm = _Normal(J, f)
        acnorm = np.sqrt(np.sum(J * J, axis=0))
        if it == 1:
            if mode == 1:
                D = np.where(acnorm == 0, 1.0, acnorm)
            xnorm = np.linalg.norm(D * x)
            delta = factor * xnorm
            if delta == 0:
                delta = factor
        gnorm = 0.0
        if fnorm != 0:
            ok = acnorm != 0
            if np.any(ok):
                gnorm = float(np.max(np.abs(nm.g[ok] / fnorm / acnorm[ok])))
        if gnorm <= gtol:
            info = 4
            break
        if mode == 1:
            D = np.maximum(D, acnorm)
        while True:
            par, p = lmpar(nm, D, delta, par)
            xt = x - p
            pnorm = np.linalg.norm(D * p)
            if it == 1:
                delta = min(delta, pnorm)
            ft = np.atleast_1d(np.asarray(fun(xt), float))
            nfev += 1
            fnorm1 = np.linalg.norm(ft)
            actred = -1.0
            if 0.1 * fnorm1 < fnorm:
                actred = 1.0 - (fnorm1 / fnorm) ** 2
            t1 = np.linalg.norm(J @ p) / fnorm
            t2 = np.sqrt(par) * pnorm / fnorm
            prered = t1 * t1 + t2 * t2 / 0.5
            dirder = -(t1 * t1 + t2 * t2)
            ratio = actred / prered if prered != 0 else 0.0
            if ratio <= 0.25:
                temp = 0.5 if actred >= 0 else 0.5 * dirder / (dirder + 0.5 * actred)
                if 0.1 * fnorm1 >= fnorm or temp < 0.1:
                    temp = 0.1
                delta = temp * min(delta, pnorm / 0.1)
                par = par / temp
            elif par == 0 or ratio >= 0.75:
                delta = pnorm / 0.5
                par = 0.5 * par
            if ratio >= 1e-4:
                x, f = xt, ft
                xnorm = np.linalg.norm(D * x)
                fnorm = fnorm1
                it += 1
            small = abs(actred) <= ftol and prered <= ftol and 0.5 * ratio <= 1
            if small:
                info = 1
            if delta <= xtol * xnorm:
                info = 2
            if small and info == 2:
                info = 3
            if info != 0:
                break
            if nfev >= maxfev:
                info = 5
            if abs(actred) <= EPSMCH and prered <= EPSMCH and 0.5 * ratio <= 1:
                info = 6
            if delta <= EPSMCH * xnorm:
                info = 7
            if gnorm <= EPSMCH:
                info = 8
            if info != 0:
                break
            if ratio >= 1e-4:
                break
        if info != 0:
            break
    res.x, res.fvec, res.nfev, res.njev, res.info = x, f, nfev, njev, info
    return res


_TO_SCIPY = {0: -1, 1: 2, 2: 3, 3: 4, 4: 1, 5: 0}


def least_squares_lm(fun, jac, x0, xtol=1e-8, gtol=1e-8, ftol=1e-8, max_nfev=None, x_scale=1.0):
    """Counterpart of ``least_squares(fun, x0, jac, method='lm', ...)`` (call_minpack)."""
    x0 = np.atleast_1d(np.asarray(x0, float))
    n = x0.size
    if ftol < EPSMCH or xtol < EPSMCH or gtol < EPSMCH:
        raise ValueError("All tolerances must be higher than machine epsilon for method 'lm'.")
    if isinstance(x_scale, str) and x_scale == 'jac':
        diag = None
    else:
        diag = 1.0 / np.broadcast_to(np.asarray(x_scale, float), (n,))
    r = lmder(fun, jac, x0, ftol=ftol, xtol=xtol, gtol=gtol, maxfev=100 * n if max_nfev is None else max_nfev,
              diag=diag, factor=100.0)
    if r.info not in _TO_SCIPY:
        raise RuntimeError('MINPACK info %d' % r.info)      # 6, 7, 8: scipy raises a KeyError here
    out = LMDerResult()
    out.x, out.fun = r.x, r.fvec
    out.jac = np.atleast_2d(np.asarray(jac(r.x), float))
    out.cost = 0.5 * float(r.fvec @ r.fvec)
    out.grad = out.jac.T @ r.fvec
    out.optimality = float(np.linalg.norm(out.grad, ord=np.inf))
    out.nfev, out.njev, out.status = r.nfev, r.njev, _TO_SCIPY[r.info]
    return out
