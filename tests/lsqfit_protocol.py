"""A gvar-free stand-in for what ``lsqfit.nonlinear_fit`` hands a fitter plugin (tests only).

lsqfit and gvar cannot be imported in this image (``import gvar`` fails), so ``register(lsqfit)`` has never run
against the real package.  This module reproduces, in independent code, the PROTOCOL on lsqfit's side of the plugin call

    fit = FITTERS[name](p0, nf, self._chiv, tol=tol, maxit=maxit, **fitterargs)          (src/lsqfit/__init__.py:662-664)

as far as a plugin can observe it -- the objects, container types and numpy calls the plugin's ``f`` goes through:

  * ``p0 = self.p0.flatten()``: a contiguous float64 vector (:566); ``nf = yp_pdf.nchiv`` (:574);
  * ``flatfcn``: the user's ``fcn`` re-wrapped so that inputs and outputs are flat (``_unpack_fcn`` and the four
    ``flatfcn_aa / ad / da / dd`` forms, :1997-2042): array parameters arrive as ``p.reshape(pshape)``, dictionary
    parameters as a BufferDict whose buffer is REPLACED by the flat vector (``po.buf = p`` -- an OBJECT array when the plugin
    differentiates), dictionary outputs are copied key by key into a BufferDict built over ``buf = y.size * [None]`` and
    returned as its ``.flat``; ``x is False`` means ``fcn(p)``;
  * ``chiv.__call__`` (src/lsqfit/_utilities.pyx:58-94): ``numpy.concatenate((fcn(p), p)) - mean`` (no ``p`` without a
    prior), the 1 x 1 blocks as ``numpy.multiply(wgts, delta[iw])`` with ``wgts`` a typed memoryview, every correlated block
    as ``numpy.dot(wgt, delta[iw])`` with ``wgt`` a 2-d memoryview, results assigned into ``numpy.zeros(nw, object)`` when
    ``mixed`` (the branch a plugin that is not gvar asks for: ``isinstance(delta[0], gvar.GVar)`` is False for its numbers);
  * user functions written against ``gvar``'s function table: ``gv.exp(x)`` and friends dispatch to ``x.exp()`` when the
    argument has such a method and to ``numpy.exp(x)`` otherwise -- for an object array numpy's loop then calls every
    element's ``.exp()`` (what :class:`GV` below does).

The whitening inputs (``yp_pdf.mean``, ``.nchiv``, ``.i_invwgts``) come from the oracle's restatement of ``gvar.PDF``
(oracle/pdf.py) -- test infrastructure on both sides of the seam; the PRODUCT under test is what sits behind
``mi355x_lm(p0, nf, chiv)``.  Nothing here is imported by ``lsqfit_amd``.
"""
import functools

import numpy as np


class BufferDict:
    """The slice of ``gvar.BufferDict`` the protocol uses: an ordered mapping of names to scalars / arrays stored in ONE flat
    buffer; ``BufferDict(other, buf=new_buffer)`` keeps ``other``'s layout over a new buffer; assigning ``.buf`` swaps the
    storage (any dtype, object included) without touching the layout; ``d[k]`` is a view of the buffer reshaped to the
    entry's shape (a scalar for shape ()); ``.flat`` / ``.size`` / ``.shape is None``."""
    shape = None

    def __init__(self, other=None, buf=None):
        self._layout = {}            # key -> (slice, shape)
        self._buf = np.zeros(0, float)
        if other is not None:
            if isinstance(other, BufferDict):
                self._layout = dict(other._layout)
                self._buf = np.array(other._buf)
            else:
                for k in other:
                    self[k] = other[k]
            if buf is not None:
                self.buf = buf

    @property
    def buf(self):
        return self._buf

    @buf.setter
    def buf(self, b):
        b = b if isinstance(b, np.ndarray) else np.array(b, dtype=object if any(v is None for v in b) else None)
        if b.ndim != 1 or b.size != self.size:
            raise ValueError('buffer of size %d expected' % self.size)
        self._buf = b

    @property
    def size(self):
        return sum(int(np.prod(shp, dtype=int)) for _, shp in self._layout.values())

    @property
    def flat(self):
        return self._buf.flat

    def flatten(self):
        return np.array(self._buf)

    def keys(self):
        return self._layout.keys()

    def __iter__(self):
        return iter(self._layout)

    def __contains__(self, k):
        return k in self._layout

    def __len__(self):
        return len(self._layout)

    def __getitem__(self, k):
        sl, shp = self._layout[k]
        return self._buf[sl.start] if shp == () else self._buf[sl].reshape(shp)

    def __setitem__(self, k, v):
        if k in self._layout:
            sl, shp = self._layout[k]
            if shp == ():
                self._buf[sl.start] = v
            else:
                self._buf[sl] = np.asarray(v, dtype=self._buf.dtype if self._buf.dtype != object else object).reshape(-1)
            return
        v = np.asarray(v)
        n0 = self._buf.size
        self._layout[k] = (slice(n0, n0 + v.size), v.shape)
        self._buf = np.concatenate([self._buf, v.reshape(-1).astype(self._buf.dtype if self._buf.dtype == object or v.dtype != object else object)])


# ---- _unpack_fcn and the four flat forms (src/lsqfit/__init__.py:1997-2042), restated -------------------------------------
def _call(fcn, x, po):
    return fcn(po) if x is False else fcn(x, po)


def _flat_of(ans):
    return ans.flat if hasattr(ans, 'flat') else np.array(ans).flat


def _array_params_array_out(p, x, fcn, pshape):
    return _flat_of(_call(fcn, x, p.reshape(pshape)))


def _dict_params_array_out(p, x, fcn, po):
    po.buf = p
    return _flat_of(_call(fcn, x, po))


def _array_params_dict_out(p, x, fcn, pshape, yo):
    fxp = _call(fcn, x, p.reshape(pshape))
    for k in yo:
        yo[k] = fxp[k]
    return yo.flat


def _dict_params_dict_out(p, x, fcn, po, yo):
    po.buf = p
    fxp = _call(fcn, x, po)
    for k in yo:
        yo[k] = fxp[k]
    return yo.flat


def unpack_fcn(fcn, p0, y, x):
    """``p0`` / ``y``: numpy arrays (``.shape`` not None) or :class:`BufferDict` (``.shape is None``) of parameter start values /
    data means; -> the flat function the fitter's ``chiv`` calls."""
    if getattr(y, 'shape', None) is not None:
        if getattr(p0, 'shape', None) is not None:
            return functools.partial(_array_params_array_out, x=x, fcn=fcn, pshape=p0.shape)
        po = BufferDict(p0, buf=np.zeros(p0.size, float))
        return functools.partial(_dict_params_array_out, x=x, fcn=fcn, po=po)
    yo = BufferDict(y, buf=y.size * [None])
    if getattr(p0, 'shape', None) is not None:
        return functools.partial(_array_params_dict_out, x=x, fcn=fcn, pshape=p0.shape, yo=yo)
    po = BufferDict(p0, buf=np.zeros(p0.size, float))
    return functools.partial(_dict_params_dict_out, x=x, fcn=fcn, po=po, yo=yo)


class Chiv:
    """``chi**2 = sum(chiv(p)**2)``: the callable lsqfit hands the plugin (src/lsqfit/_utilities.pyx:50-94).  ``fd``: the
    whitening (``mean``, ``nchiv``, ``i_invwgts``); weights are held as memoryviews, as the Cython class types them."""

    def __init__(self, fd, fcn, noprior):
        self.mean = np.asarray(fd.mean, float)
        self.nw = int(fd.nchiv)
        self.inv_wgts = [(np.asarray(iw, np.intp), memoryview(np.ascontiguousarray(w, float))) for iw, w in fd.i_invwgts]
        self.fcn = fcn
        self.noprior = noprior
        self.calls = []              # (dtype of p, mixed) per call: what the plugin did with it

    def __call__(self, p, mixed=False):
        self.calls.append((np.asarray(p).dtype, mixed))
        if self.noprior:
            delta = self.fcn(p) - self.mean
        else:
            delta = np.concatenate((self.fcn(p), p)) - self.mean
        ans = np.zeros(self.nw, object if mixed else float)
        iw, wgts = self.inv_wgts[0]
        i1, i2 = 0, len(iw)
        if i2 > 0:
            ans[i1:i2] = np.multiply(wgts, delta[iw])
        for iw, wgt in self.inv_wgts[1:]:
            i1 = i2
            i2 += len(wgt)
            ans[i1:i2] = np.dot(wgt, delta[iw])
        return ans


class GV:
    """``gv``: the function table user code is written against.  Each function tries the argument's own method first and
    falls back on numpy's ufunc (whose object loop calls the elements' methods)."""

    @staticmethod
    def _make(name):
        def f(x):
            try:
                return getattr(x, name)()
            except AttributeError:
                return getattr(np, name)(x)
        f.__name__ = name
        return staticmethod(f)


for _n in ('exp', 'log', 'sqrt', 'sin', 'cos', 'tan', 'arcsin', 'arccos', 'arctan', 'sinh', 'cosh', 'tanh'):
    setattr(GV, _n, GV._make(_n))
gv = GV


def fitter_call(fcn, x, ymean, yerr, prior_mean=None, prior_err=None, p0=None, svdcut=1e-12):
    """What ``nonlinear_fit.__init__`` prepares before it calls the plugin (:539-575): -> (p0 flat, nf, chiv, pdf).
    ``ymean`` / ``prior_mean``: arrays or BufferDicts (dictionary data / parameters); errors: sdev arrays or covariance
    matrices of the FLATTENED vectors."""
    from oracle import fit as ofit
    yflat = np.asarray(ymean.flatten() if isinstance(ymean, BufferDict) else ymean, float).reshape(-1)
    pflat = None if prior_mean is None else np.asarray(prior_mean.flatten() if isinstance(prior_mean, BufferDict) else prior_mean,
                                                       float).reshape(-1)
    pdf = ofit.build_pdf(yflat, yerr, pflat, prior_err, svdcut=svdcut)
    if p0 is None:
        p0 = prior_mean if isinstance(prior_mean, BufferDict) else np.asarray(prior_mean, float)
        flat0 = ofit.default_p0(pflat, np.sqrt(np.diag(prior_err)) if np.ndim(prior_err) == 2 else np.asarray(prior_err, float))
        p0 = BufferDict(p0, buf=flat0) if isinstance(p0, BufferDict) else flat0.reshape(np.shape(p0))
    yshape = ymean if isinstance(ymean, BufferDict) else np.asarray(ymean, float)
    flatfcn = unpack_fcn(fcn, p0, yshape, x)
    chiv = Chiv(pdf, flatfcn, noprior=prior_mean is None)
    p0flat = np.ascontiguousarray(p0.flatten() if isinstance(p0, BufferDict) else np.asarray(p0, float).reshape(-1))
    return p0flat, int(pdf.nchiv), chiv, pdf


def reduce(fit, pdf, P):
    """What the caller computes from the plugin's attributes (src/lsqfit/__init__.py:665-679,709-725): -> dict(chi2, dof, Q,
    logGBF, pmean, psdev, cov, nit, stopping_criterion, description)."""
    from oracle.fit import gammaQ
    assert fit.error is None, fit.error
    f = np.asarray(fit.f, float)
    chi2 = float(np.sum(f ** 2))
    dof = int(pdf.nchiv) - P
    J = np.asarray(fit.J, float)
    sign, ld = np.linalg.slogdet(J.T.dot(J))
    return dict(chi2=chi2, dof=dof, Q=float(gammaQ(dof / 2., chi2 / 2.)),
                logGBF=0.5 * (-ld - pdf.logdet - chi2 - dof * np.log(2. * np.pi)),
                pmean=np.asarray(fit.x, float).reshape(-1), cov=np.asarray(fit.cov, float), psdev=np.sqrt(np.diag(fit.cov)),
                nit=fit.nit, stopping_criterion=fit.stopping_criterion, description=getattr(fit, 'description', ''),
                tol=fit.tol, results=fit.results)


# ---- the three reference examples, written the way the reference writes them ---------------------------------------------
def simple_example():
    """examples/simple.py:28-47: dictionary data (two 2 x 2 covariance blocks + a scalar), dictionary parameters, ``gv.exp``."""
    y = BufferDict()
    y['data1'] = np.array([1.376, 2.010])
    y['data2'] = np.array([1.329, 1.582])
    y['b/a'] = np.array(2.0)
    ycov = np.zeros((5, 5))
    ycov[:2, :2] = [[0.0047, 0.01], [0.01, 0.056]]
    ycov[2:4, 2:4] = [[0.0047, 0.0067], [0.0067, 0.0136]]
    ycov[4, 4] = 0.5 ** 2
    x = BufferDict()
    x['data1'] = np.array([0.1, 1.0])
    x['data2'] = np.array([0.1, 0.5])
    prior = BufferDict()
    prior['a'] = np.array(0.5)
    prior['b'] = np.array(0.5)

    def fcn(x, p):
        ans = {}
        for k in ['data1', 'data2']:
            ans[k] = gv.exp(p['a'] + x[k] * p['b'])
        ans['b/a'] = p['b'] / p['a']
        return ans
    return dict(fcn=fcn, x=x, ymean=y, yerr=ycov, prior_mean=prior, prior_err=np.array([0.5, 0.5]))


def p_corr_example(k):
    """examples/p-corr.py:44-61: array data and parameters, ``p[1] = 20 p[0] + 0.0(1)`` correlated with ``p[0]``."""
    from oracle import gvar_lite
    ym, ys = gvar_lite.parse_array(k['y'])
    pcov = np.eye(4)
    pcov[1, 1] = 400. + 0.1 ** 2
    pcov[0, 1] = pcov[1, 0] = 20.

    def fcn(x, p):
        return (p[0] * (x ** 2 + p[1] * x)) / (x ** 2 + x * p[2] + p[3])
    return dict(fcn=fcn, x=np.array(k['x']), ymean=ym, yerr=ys, prior_mean=np.zeros(4), prior_err=pcov)


def x_err_example(k):
    """examples/x-err.py:44-71: ``fcn(p)`` with no x (``x is False``), dictionary parameters ``p['b']`` (unpacked into four
    names) and ``p['x']`` (the 15 abscissae are fit parameters), array data, ``gv.exp`` on an array."""
    from oracle import gvar_lite
    xm, xs = gvar_lite.parse_array(k['x'])
    ym, ys = gvar_lite.parse_array(k['y'])
    bm, bs = gvar_lite.parse_array(k['prior_b'])
    prior = BufferDict()
    prior['b'] = bm
    prior['x'] = xm

    def fcn(p):
        b0, b1, b2, b3 = p['b']
        x = p['x']
        return b0 / ((1. + gv.exp(b1 - b2 * x)) ** (1. / b3))
    return dict(fcn=fcn, x=False, ymean=ym, yerr=ys, prior_mean=prior, prior_err=np.concatenate([bs, xs]))


def y_vs_x_example(k, nexp):
    """examples/y-vs-x.py:20-67: array data with a dense 8 x 8 covariance whose smallest correlation eigenvalue the svdcut
    raises (``svdcut/n = 1e-12/1``: gvar whitens that block in its eigen basis), dictionary parameters ``p['a']``, ``p['E']``,
    the fit function a Python ``sum`` over ``zip(a, E)`` of ``ai * np.exp(-Ei * x)``."""
    prior = BufferDict()
    prior['a'] = np.full(nexp, 0.5)
    prior['E'] = np.arange(1, nexp + 1.0)

    def fcn(x, p):
        a = p['a']
        E = p['E']
        return sum(ai * np.exp(-Ei * x) for ai, Ei in zip(a, E))
    return dict(fcn=fcn, x=np.array(k['x']), ymean=np.array(k['ymean']), yerr=np.array(k['ycov']), prior_mean=prior,
                prior_err=np.full(2 * nexp, 0.4))


def nist_example(name, nist):
    """examples/nist.py (one function per StRD problem, e.g. misra1a :102-120): ``fcn(x, b)`` unpacks ``b1, b2, ... = b`` and
    evaluates the certified formula with gvar's functions; priors ``0 +- 200 |b|``, start 2, ``tol = 1e-10``."""
    from tests.helpers import nist_problem
    pr = nist_problem(name, nist)
    cols = pr['columns'][1:]
    code = compile(pr['expr'], '<nist:%s>' % name, 'eval')
    ns = {'exp': gv.exp, 'log': gv.log, 'sqrt': gv.sqrt, 'sin': gv.sin, 'cos': gv.cos, 'arctan': gv.arctan, 'pi': np.pi}

    def fcn(x, b):
        env = dict(ns)
        env.update(x if isinstance(x, dict) else {cols[0]: x})
        for i in range(pr['P']):
            env['b%d' % (i + 1)] = b[i]
        return eval(code, {'__builtins__': {}}, env)
    x = {c: pr['x'][c] for c in cols} if len(cols) > 1 else pr['x'][cols[0]]
    return dict(fcn=fcn, x=x, ymean=pr['y'], yerr=pr['ysd'], prior_mean=pr['prior_mean'], prior_err=pr['prior_sd'], p0=pr['p0']), pr
