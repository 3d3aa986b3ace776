"""Evidence maximisation over fits that share shape.

Counterpart of ``lsqfit.empbayes_fit`` (src/lsqfit/_extras.py:30-185): pick the
hyper-parameter ``z`` that maximises ``logGBF`` of ``nonlinear_fit(**fitargs(z))``.
The reference walks a scalar minimiser (``_multiminex``) through one whole fit
per function value.  Here the search is organised around what the device does
well -- many same-shape fits in lockstep:

  * :class:`EvidenceSurface` turns a LIST of z values into one call of the
    batched hipGraph engine (:class:`lsqfit_amd.BatchedFits`) whenever the
    fits differ only in their diagonal prior; anything else (correlated
    priors, ``cross=``, ``linear=``, bounds, ...) is evaluated fit by fit;
  * :func:`simplex_search` is a Nelder-Mead search written for such an
    evaluator: the whole starting simplex is one batch, and every iteration
    evaluates its four candidate points (reflection, expansion, both
    contractions) speculatively as one batch, then applies the usual rules --
    the iterates are those of the textbook method with the standard
    coefficients (1, 2, 1/2, 1/2), at one device round trip per iteration;
  * ``prior_width_sweep`` is BASELINE.json's config-5 shape: every width of a
    list in ONE batch.
"""
import warnings
from collections import namedtuple

import numpy as np

from .fit import nonlinear_fit

SweepFit = namedtuple('SweepFit', 'width pmean psdev chi2 dof Q logGBF nit stopping_criterion')

_BATCHABLE = {'data', 'model', 'prior', 'p0', 'tol', 'maxit', 'svdcut'}


def _same_x(a, b):
    if a is b:
        return True
    if hasattr(a, 'keys') or hasattr(b, 'keys'):
        return hasattr(a, 'keys') and hasattr(b, 'keys') and list(a) == list(b) and all(_same_x(a[k], b[k]) for k in a)
    if a is False or b is False or a is None or b is None:
        return False
    return _same(a, b)


def _same(a, b):
    if a is b:
        return True
    try:
        a, b = np.asarray(a, float), np.asarray(b, float)
    except (TypeError, ValueError):
        return False
    return a.shape == b.shape and bool(np.array_equal(a, b))


class EvidenceSurface:
    """z values -> ``-(logGBF + plausibility)``, many at a time.

    ``fitargs(z)`` has the reference's meaning: the keyword dict of ``nonlinear_fit`` for this z,
    or a pair (dict, log prior plausibility of z).  The best point seen so far (``zbest``) and
    its parameters (the warm start of every later fit, _extras.py:160-161) are remembered."""

    def __init__(self, fitargs, p0=None, scalar=True, fitter=nonlinear_fit):
        self.fitargs, self.scalar = fitargs, scalar
        self.fitter = fitter          # anything but lsqfit_amd.nonlinear_fit is called fit by fit (never batched)
        self.warm = None if p0 is None else np.array(p0, float).reshape(-1)
        self.zbest, self.fbest = None, np.inf
        self.nfits = self.nbatches = 0
        self._engine = self._engine_key = None
        self._traced = None

    def unpack(self, zrow, canonical=True):
        out = self.fitargs(float(zrow[0]) if self.scalar else zrow)
        a, pl = (dict(out), 0.0) if hasattr(out, 'keys') else (dict(out[0]), float(out[1]))
        return (self._canonical(a) if canonical and self.fitter is nonlinear_fit else a), pl

    def _canonical(self, a):
        """The reference's own form of the arguments -- ``fcn=`` a Python function, array or dictionary parameters
        (examples/empbayes.py:31-34) -- as the batch engine wants them: the function is recorded ONCE
        (lsqfit_amd.trace) and every z reuses that model object; data and prior flattened in the traced order."""
        if a.get('model') is not None or a.get('fcn') is None or 'data' not in a or a.get('prior') is None:
            return a
        from .trace import flatten_mean_err, trace
        dd = a['data'] if len(a['data']) == 3 else (False,) + tuple(a['data'])
        xx, ym, ye = dd
        prior = a['prior']
        if not isinstance(prior, tuple) or len(prior) != 2:
            return a
        c = self._traced
        layout = (tuple((k, np.shape(v)) for k, v in prior[0].items()) if hasattr(prior[0], 'keys') else np.shape(prior[0]))
        if c is None or c[0] is not a['fcn'] or not _same_x(c[1], xx) or c[7] != layout:
            tr = trace(a['fcn'], xx, prior[0], y=ym)
            ymf, yef = flatten_mean_err(ym, ye)
            c = self._traced = (a['fcn'], xx, tr, ym, ye, ymf, yef, layout)
        tr = c[2]
        if c[3] is ym and c[4] is ye:
            ymf, yef = c[5], c[6]
        else:
            ymf, yef = flatten_mean_err(ym, ye)
            if _same(ymf, c[5]) and _same(yef, c[6]):
                ymf, yef = c[5], c[6]
        b = {k: v for k, v in a.items() if k != 'fcn'}
        b['model'] = tr.model
        b['data'] = (tr.x, ymf, yef)
        b['prior'] = flatten_mean_err(prior[0], prior[1])
        if b.get('p0') is not None:
            b['p0'] = tr.pack_params(b['p0'])
        return b

    # -- one lockstep batch --------------------------------------------------------------------
    def _engine_for(self, args, n):
        from .batched import BatchedFits
        x, ym, yerr = args['data']
        key = (args['model'], x, ym, yerr, args.get('svdcut', False))
        k0 = self._engine_key
        if (self._engine is not None and self._engine.B == n and k0[0] is key[0]
                and all(_same(u, v) for u, v in zip(k0[1:4], key[1:4])) and k0[4] == key[4]):
            return self._engine
        if self._engine is not None:
            self._engine.close()
        pm = np.zeros((n, args['model'].n_param))
        kw = {} if key[4] is False else dict(svdcut=key[4])
        self._engine = BatchedFits(args['model'], x, ym, yerr, pm, np.ones_like(pm), **kw)
        self._engine_key = key
        return self._engine

    def _batch(self, items):
        """items: [(args, plausibility)] that differ in their diagonal priors only -> values or None."""
        a0 = items[0][0]
        pri = []
        for a, _ in items:
            if set(a) - _BATCHABLE or 'data' not in a or a.get('prior') is None:
                return None
            if a['model'] is not a0['model'] or not all(_same(u, v) for u, v in zip(a['data'], a0['data'])):
                return None
            if any(a.get(k) != a0.get(k) for k in ('tol', 'maxit', 'svdcut')):
                return None
            pm, pe = a['prior']
            if hasattr(pe, 'keys'):
                return None
            pe = np.asarray(pe, float)
            if pe.ndim > 1:
                return None
            pri.append((np.asarray(pm, float).reshape(-1), np.broadcast_to(pe, np.shape(pm)).reshape(-1)))
        eng = self._engine_for(a0, len(items))
        eng.set_priors(np.array([p[0] for p in pri]), np.array([p[1] for p in pri]))
        p0 = self.warm if self.warm is not None else a0.get('p0')
        kw = {k: a0[k] for k in ('tol', 'maxit') if a0.get(k) is not None}
        out = eng.run(p0=p0, **kw)
        self.nbatches += 1
        vals = -np.asarray(out['logGBF'], float) - np.array([pl for _, pl in items])
        return vals, out['pmean']

    def __call__(self, zs):
        zs = np.atleast_2d(np.asarray(zs, float))
        items = [self.unpack(z) for z in zs]
        got = self._batch(items) if len(items) > 1 and self.fitter is nonlinear_fit else None
        if got is None:
            vals, pmeans = np.empty(len(items)), []
            for i, (a, pl) in enumerate(items):
                if self.warm is not None and 'p0' not in a:
                    a['p0'] = self.warm
                fit = self.fitter(**a)
                g = np.nan if fit.logGBF is None else fit.logGBF
                vals[i] = -g - pl
                pmeans.append(fit.pmean)
        else:
            vals, pmeans = got
        self.nfits += len(items)
        vals = np.where(np.isfinite(vals), vals, np.inf)      # a null logGBF is never the optimum
        i = int(np.argmin(vals))
        if vals[i] < self.fbest:
            self.fbest, self.zbest, self.warm = float(vals[i]), zs[i].copy(), np.array(pmeans[i], float)
        return vals

    def close(self):
        if self._engine is not None:
            self._engine.close()
            self._engine = None


def simplex_search(fmany, z0, tol=1e-4, maxit=1000):
    """Nelder-Mead on ``fmany(points[m, n]) -> values[m]`` with speculative candidate batches.
    Stops when the simplex and its values are both within ``tol`` of the best vertex."""
    z0 = np.asarray(z0, float).reshape(-1)
    n = z0.size
    sim = np.tile(z0, (n + 1, 1))
    for k in range(n):
        sim[k + 1, k] = 1.05 * z0[k] if z0[k] != 0.0 else 2.5e-4
    val = np.asarray(fmany(sim), float)
    for _ in range(int(maxit)):
        order = np.argsort(val, kind='stable')
        sim, val = sim[order], val[order]
        if np.max(np.abs(sim[1:] - sim[0])) <= tol and np.max(np.abs(val[1:] - val[0])) <= tol:
            break
        cen = sim[:-1].mean(axis=0)
        step = cen - sim[-1]
        cand = np.array([cen + step, cen + 2.0 * step, cen + 0.5 * step, cen - 0.5 * step])
        fr, fe, foc, fic = fmany(cand)
        if fr < val[0]:
            pick = (cand[1], fe) if fe < fr else (cand[0], fr)
        elif fr < val[-2]:
            pick = (cand[0], fr)
        elif fr < val[-1] and foc <= fr:
            pick = (cand[2], foc)
        elif fr >= val[-1] and fic < val[-1]:
            pick = (cand[3], fic)
        else:
            pick = None
        if pick is None:                                      # shrink towards the best vertex
            sim[1:] = sim[0] + 0.5 * (sim[1:] - sim[0])
            val[1:] = fmany(sim[1:])
        else:
            sim[-1], val[-1] = pick
    k = int(np.argmin(val))
    return sim[k], float(val[k])


def empbayes_fit(z0, fitargs, p0=None, tol=1e-4, maxit=1000, fitter=nonlinear_fit):
    """-> (fit, z): the fit at the z that maximises ``logGBF`` (+ plausibility), and that z,
    laid out like ``z0`` (number or array).  ``tol`` / ``maxit`` steer the simplex search as the
    reference's ``minargs`` steer its minimiser (src/lsqfit/_scipy.py:224-227).  ``fitter`` is any callable
    with ``nonlinear_fit``'s keyword interface whose result has ``logGBF`` and ``pmean`` (the reference accepts one,
    _extras.py:30-41,:163); only ``lsqfit_amd.nonlinear_fit`` itself is evaluated in lockstep batches, another
    fitter fit by fit."""
    surface = EvidenceSurface(fitargs, p0=p0, scalar=np.shape(z0) == (), fitter=fitter)
    try:
        simplex_search(surface, np.atleast_1d(np.asarray(z0, float)), tol=tol, maxit=maxit)
        if surface.zbest is None:
            raise ValueError('empbayes_fit: logGBF is undefined at every z that was tried')
        if surface.nfits and not np.isfinite(surface.fbest):
            warnings.warn('empbayes_fit: null logGBF')
        args, _ = surface.unpack(surface.zbest, canonical=False)    # (the caller's own form: fit.p comes back in its layout)
        args.setdefault('p0', surface.warm)
        z = float(surface.zbest[0]) if surface.scalar else surface.zbest
        return fitter(**args), z
    finally:
        surface.close()


def prior_width_sweep(*args, p0=None, **runkw):
    """``prior_width_sweep(data, model, prior_mean, widths)``: fit the same data under the priors
    ``prior_mean +- widths[j]`` (each width a scalar or a P-vector) as ONE lockstep batch on the device.
    -> [SweepFit] in the order of ``widths`` (a namedtuple: width pmean psdev chi2 dof Q logGBF nit
    stopping_criterion).  Round 1's form ``prior_width_sweep(problem, data, model, prior_mean, widths)`` -- a
    resident ``DeviceProblem`` first, a list of fit objects back -- is still accepted: the problem is not
    needed any more and is ignored; the return value is the list of SweepFit records either way."""
    from .batched import BatchedFits
    if len(args) == 5:
        warnings.warn('prior_width_sweep(problem, data, ...): the leading DeviceProblem is ignored since round 2 '
                      '(the sweep runs as one batch); it returns SweepFit records', DeprecationWarning, stacklevel=2)
        args = args[1:]
    if len(args) != 4:
        raise TypeError('prior_width_sweep(data, model, prior_mean, widths, p0=None, **run_options)')
    data, model, prior_mean, widths = args
    x, ymean, yerr = data
    pm = np.asarray(prior_mean, float).reshape(-1)
    sd = np.array([np.broadcast_to(np.asarray(w, float), pm.shape) for w in widths])
    eng = BatchedFits(model, x, ymean, yerr, np.tile(pm, (len(sd), 1)), sd)
    try:
        out = eng.run(p0=p0, **runkw)
        psd = out.get('psdev')
        if psd is None:
            psd = np.sqrt(np.array([np.diag(eng.cov(b)) for b in range(len(sd))]))
        return [SweepFit(widths[b], out['pmean'][b], psd[b], float(out['chi2'][b]), out['dof'],
                         float(out['Q'][b]), float(out['logGBF'][b]), int(out['nit'][b]),
                         int(out['stopping_criterion'][b])) for b in range(len(sd))]
    finally:
        eng.close()
