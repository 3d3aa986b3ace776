"""Simulated and bootstrap refits: host mirror of
``nonlinear_fit.simulated_fit_iter`` / ``simulated_data_iter`` /
``bootstrapped_fit_iter`` (src/lsqfit/__init__.py:1391-1469,1470-1543,1548-1642).

The reference loops over whole Python fits, one per copy of the data.  Every copy
has the structure of the original fit (same model, x, covariance, whitening; only
the data means -- and optionally the prior means -- change), so here the copies run
as ONE lockstep batch on the device (:class:`lsqfit_amd.BatchedFits`, C ABI
``lsqamdb_*``): correlated data blocks and a correlated (dense) prior are shared by
the copies, data and prior MEANS are per copy.

Random numbers: ``numpy.random.Generator(PCG64(seed))`` (the reference draws from
gvar's global RNG through ``gvar.bootstrap_iter``: mean + a N(0, C_regulated) deviate).
"""
import numpy as np

from .batched import BatchedFits


def _data_order(wh, a):
    """Rows in the whitening's own order (interleaved covariance components are made contiguous by a permutation) ->
    the caller's row order."""
    perm = getattr(wh, 'perm', None)
    if perm is None:
        return a
    out = np.empty_like(a)
    out[..., perm] = a
    return out


def simulated_data(fit, n, pexact=None, add_priornoise=False, seed=0):
    """-> (ymeans[n, N], prior_means[n, P] or None): ``simulated_data_iter``
    (__init__.py:1519-1543): data means = fcn(pexact) + noise with the covariance of
    ``fit.y``; prior means move only with ``add_priornoise``."""
    wh = fit.whitening
    rng = np.random.Generator(np.random.PCG64(seed))
    pexact = fit.pmean if pexact is None else np.asarray(pexact, float)
    f = fit.problem.fcn(pexact)
    ymeans = f[None, :] + _data_order(wh, wh.draw_data(rng, n))
    if not wh.has_prior:
        return ymeans, None
    pm = np.broadcast_to(wh.prior_mean, (n, wh.prior_mean.size)).copy()
    if add_priornoise:
        pm += wh.draw_prior(rng, n)
    return ymeans, pm


def bootstrap_data(fit, n, seed=0):
    """-> (ymeans, prior_means): ``bootstrapped_fit_iter`` with ``datalist=None``
    (__init__.py:1615-1626): both the data and the prior means are redrawn around
    their original values."""
    wh = fit.whitening
    rng = np.random.Generator(np.random.PCG64(seed))
    ymeans = _data_order(wh, wh.ymean[None, :] + wh.draw_data(rng, n))
    if not wh.has_prior:
        return ymeans, None
    return ymeans, wh.prior_mean[None, :] + wh.draw_prior(rng, n)


class ResampledFits(dict):
    """Results of n refits: arrays with a leading copy index (``pmean[n, P]``, ``chi2``,
    ``dof``, ``Q``, ``nit``, ``stopping_criterion``, ``psdev`` ...), plus the inputs
    ``ymeans`` / ``prior_means`` and ``engine`` ('batched')."""
    __getattr__ = dict.__getitem__


_BATCHED_ARGS = {'alg', 'scaler', 'factor_up', 'factor_down', 'solver', 'avmax'}


def _batchable(fit):
    """the lockstep engine runs plain LM: a fit made with another fitter, another algorithm, a robust loss, bounds or
    ``x_scale`` is refitted copy by copy with ITS fitter and arguments (src/lsqfit/__init__.py:1457-1459,:1603-1604)"""
    args = getattr(fit, 'fitterargs', {})
    # the batch engine factors the normal equations (solver = cholesky) with the default avmax: a fit made with solver='qr' or
    # another avmax keeps ITS arguments, i.e. goes copy by copy; scaler / factor_up / factor_down are forwarded (refit)
    return (fit.fitter == 'mi355x_lm' and args.get('alg', 'lm') == 'lm' and set(args) <= _BATCHED_ARGS and not getattr(fit, 'linear', None)
            and args.get('solver', 'cholesky') == 'cholesky' and float(args.get('avmax', 0.75)) == 0.75)


def _sequential_copies(fit, ymeans, prior_means, p0, tol=None, maxit=None, covariance=True):
    """the copies one after the other on the fit's resident device problem, through ``nonlinear_fit`` with the original
    fit's fitter and fitter arguments"""
    from .fit import nonlinear_fit
    wh, pr = fit.whitening, fit.problem
    n = ymeans.shape[0]
    tol = fit.tol if tol is None else tol
    maxit = fit.maxit if maxit is None else maxit
    keys = ('pmean', 'psdev', 'chi2', 'Q', 'nit', 'stopping_criterion', 'logGBF')
    out = {k: [] for k in keys}
    cov = []
    y0 = wh.ymean.copy()
    pm0 = wh.prior_mean.copy() if wh.has_prior else None
    perm = getattr(wh, 'perm', None)
    prec = None
    if wh.has_prior:
        prec = getattr(wh, 'prior_prec_dev', None)
        prec = wh.prior_prec if prec is None else prec
    try:
        for k in range(n):
            yk = np.asarray(ymeans[k], float)
            wh.ymean = yk if perm is None else yk[perm]
            pr.set_ymean(yk)
            prior_k = None
            if wh.has_prior:
                wh.prior_mean = np.asarray(prior_means[k], float)
                pr.set_prior(wh.prior_mean, prec)
                prior_k = (wh.prior_mean, None)
            f = nonlinear_fit(data=(fit.problem_x, yk, None), model=fit.model, prior=prior_k, p0=np.asarray(p0, float).reshape(n, -1)[k]
                              if np.ndim(p0) == 2 else p0, problem=pr, tol=tol, maxit=maxit, fitter=fit.fitter,
                              linear=getattr(fit, 'linear', None) or None, **getattr(fit, 'fitterargs', {}))
            for key in keys:
                out[key].append(getattr(f, key))
            cov.append(f.cov)
    finally:
        wh.ymean = y0
        pr.set_ymean(y0 if perm is None else _data_order(wh, y0))
        if wh.has_prior:
            wh.prior_mean = pm0
            pr.set_prior(pm0, prec)
    res = ResampledFits({k: np.array(v) for k, v in out.items()})
    if covariance:
        res['cov'] = np.array(cov)
    res['dof'] = np.full(n, fit.dof)
    res['engine'] = 'sequential (%s)' % fit.fitter
    res['ymeans'] = ymeans
    res['prior_means'] = prior_means
    res['rounds'] = n
    return res


def refit(fit, ymeans, prior_means, p0, tol=None, maxit=None, covariance=True):
    """Refit ``fit``'s problem for every row of ``ymeans`` (and ``prior_means``)."""
    if not _batchable(fit):
        return _sequential_copies(fit, ymeans, prior_means, p0, tol, maxit, covariance)
    wh = fit.whitening
    n, P = ymeans.shape[0], fit.pmean.size
    tol = fit.tol if tol is None else tol
    maxit = fit.maxit if maxit is None else maxit
    p0 = np.broadcast_to(np.asarray(p0, float), (n, P))
    dense = wh.has_prior and wh.prior_dense
    x, ym = fit.problem_x, ymeans
    perm = getattr(wh, 'perm', None)
    if perm is not None:             # the engine works in the whitening's row order
        x = np.asarray(x, float).reshape(wh.n_data, fit.model.n_x)[perm]
        ym = np.ascontiguousarray(np.asarray(ymeans, float)[:, perm])
    bf = BatchedFits(fit.model, x, ym, None,
                     prior_means if wh.has_prior else None,
                     None if (dense or not wh.has_prior) else wh.prior_sdev, whitening=wh, n_fits=n,
                     prior_prec=wh.prior_prec if dense else None,
                     prior_logdet=(wh.logdet - wh.logdet_data) if dense else None, rows_permuted=perm is not None)
    args = getattr(fit, 'fitterargs', {})      # the copies are fitted with the original fit's arguments (src/lsqfit/__init__.py:1457-1459)
    out = bf.run(p0=p0, tol=tol, maxit=maxit, covariance=covariance, scaler=args.get('scaler', 'more'),
                 factor_up=float(args.get('factor_up', 3.0)), factor_down=float(args.get('factor_down', 2.0)))
    bf.close()
    res = ResampledFits(out)
    res['engine'] = 'batched'
    res['ymeans'] = ymeans
    res['prior_means'] = prior_means
    return res


# ---- data correlated with the prior (nonlinear_fit(..., cross=...): concat(y, prior) whitened as one vector) ---------------
# What the reference's iterators do with such data follows from gvar.bootstrap_iter, which returns NEW gvars carrying the
# covariance of what it was handed (src/lsqfit/__init__.py:1519-1543,:1607-1626):
#   bootstrapped_fit_iter           hands it y AND the prior together: every copy keeps the full joint covariance;
#   simulated_fit_iter, priornoise  likewise (yp = [y, prior]);
#   simulated_fit_iter, no noise    hands it y alone: the copies' data are no longer correlated with the (unchanged) prior.
# The first two are refits of the SAME joint problem with new means -- data rows and the prior entries that travel as rows
# alike -- run one after the other on the fit's resident device problem (the batched engine has no parameter rows); the
# third is an ordinary resampled fit of y with its own covariance and the prior with its own, one lockstep batch.
def _joint_copies(fit, zmeans, p0, tol=None, maxit=None):
    from .fit import nonlinear_fit
    wh, pr = fit.whitening, fit.problem
    n, P, N = zmeans.shape[0], fit.pmean.size, fit.whitening.n_model
    tol = fit.tol if tol is None else tol
    maxit = fit.maxit if maxit is None else maxit
    keys = ('pmean', 'psdev', 'chi2', 'Q', 'nit', 'stopping_criterion', 'logGBF')
    out = {k: [] for k in keys}
    cov = []
    z0 = wh.ymean.copy()
    back = np.empty(N + P, int)
    back[wh.row_src] = np.arange(N + P)              # caller's index -> joint row
    try:
        for k in range(n):
            wh.ymean = zmeans[k]                     # (the whitening's means, in ITS row order: what maxit = 0 / default p0 read)
            pr.set_ymean(zmeans[k])
            prior_k = zmeans[k][back[N:]]
            f = nonlinear_fit(data=(fit.problem_x, zmeans[k][back[:N]], None), model=fit.model,
                              prior=(prior_k, wh.prior_cov_host), p0=p0, problem=pr, tol=tol, maxit=maxit,
                              fitter=fit.fitter, **getattr(fit, 'fitterargs', {}))      # (src/lsqfit/__init__.py:1457-1459,:1603-1604)
            for key in keys:
                out[key].append(getattr(f, key))
            cov.append(f.cov)
    finally:
        wh.ymean = z0
        pr.set_ymean(z0)
    res = ResampledFits({k: np.array(v) for k, v in out.items()})
    res['cov'] = np.array(cov)
    res['dof'] = np.full(n, fit.dof)
    res['engine'] = 'sequential (joint rows)'
    res['ymeans'] = zmeans[:, back[:N]]
    res['prior_means'] = zmeans[:, back[N:]]
    res['rounds'] = n
    return res


def _joint_draws(fit, n, seed, centre):
    """(n, N + P) means in the joint whitening's row order: centre + deviates with the regulated joint covariance"""
    wh = fit.whitening
    rng = np.random.Generator(np.random.PCG64(seed))
    return centre[None, :] + wh.draw_data(rng, n)


def simulated_fits(fit, n, pexact=None, add_priornoise=False, seed=0, **kw):
    """n simulated copies of ``fit`` refitted from ``p0 = pexact`` (``simulated_fit_iter``,
    __init__.py:1453-1469).  ``result.pexact`` holds the generating parameters."""
    pexact = fit.pmean if pexact is None else np.asarray(pexact, float)
    wh = fit.whitening
    if getattr(wh, 'joint', False):
        f = fit.problem.fcn(pexact)                  # joint row order; rows that are parameters hold p_j
        if add_priornoise or getattr(wh, 'rows_only', False):
            centre = np.where(wh.row_param >= 0, wh.ymean, f)       # data rows: fcn(pexact); prior rows: the prior means
            zm = _joint_draws(fit, n, seed, centre)
            if not add_priornoise:       # (rows_only: data and prior are uncorrelated, so leaving the prior rows where they are IS a draw of y alone)
                zm[:, wh.row_param >= 0] = centre[wh.row_param >= 0]
            res = _joint_copies(fit, zm, pexact, **kw)
        else:
            # the copies' data are new, independent of the prior: an ordinary fit of (y + noise_y, prior) without cross terms
            from .fit import nonlinear_fit
            N = wh.n_model
            back = np.empty(wh.n_data, int)
            back[wh.row_src] = np.arange(wh.n_data)
            plain = nonlinear_fit(data=(fit.problem_x, f[back[:N]], wh.data_cov_host), model=fit.model,
                                  prior=(wh.prior_mean_host, wh.prior_cov_host), p0=pexact, svdcut=fit.svdcut, tol=fit.tol, maxit=0)
            rng = np.random.Generator(np.random.PCG64(seed))
            ymeans = f[back[:N]][None, :] + _data_order(plain.whitening, plain.whitening.draw_data(rng, n))
            pm = np.broadcast_to(wh.prior_mean_host, (n, pexact.size)).copy()
            res = refit(plain, ymeans, pm, pexact, **dict(dict(tol=fit.tol, maxit=fit.maxit), **kw))
        res['pexact'] = pexact
        return res
    ymeans, pm = simulated_data(fit, n, pexact, add_priornoise, seed)
    res = refit(fit, ymeans, pm, pexact, **kw)
    res['pexact'] = pexact
    return res


def bootstrapped_fits(fit, n, seed=0, **kw):
    """n bootstrap copies of ``fit`` refitted from ``p0 = fit.pmean``
    (``bootstrapped_fit_iter``, __init__.py:1607-1626)."""
    if getattr(fit.whitening, 'joint', False):
        return _joint_copies(fit, _joint_draws(fit, n, seed, fit.whitening.ymean), fit.pmean, **kw)
    ymeans, pm = bootstrap_data(fit, n, seed)
    return refit(fit, ymeans, pm, fit.pmean, **kw)
