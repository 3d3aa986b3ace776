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


def refit(fit, ymeans, prior_means, p0, tol=None, maxit=None, covariance=True):
    """Refit ``fit``'s problem for every row of ``ymeans`` (and ``prior_means``)."""
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
    out = bf.run(p0=p0, tol=tol, maxit=maxit, covariance=covariance)
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
                              prior=(prior_k, wh.prior_cov_host), p0=p0, problem=pr, tol=tol, maxit=maxit)
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
        if add_priornoise:
            centre = np.where(wh.row_param >= 0, wh.ymean, f)       # data rows: fcn(pexact); prior rows: the prior means
            res = _joint_copies(fit, _joint_draws(fit, n, seed, centre), pexact, **kw)
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
