"""Hyper-parameter sweeps over fits that share shape: host mirror of
``lsqfit.empbayes_fit`` (src/lsqfit/_extras.py:30-185).

``empbayes_fit(z0, fitargs)`` maximises ``logGBF`` of
``nonlinear_fit(**fitargs(z))`` over ``z`` with a Nelder-Mead search (the
reference's ``_multiminex``: scipy ``minimize(method='Nelder-Mead')``,
src/lsqfit/_scipy.py:224-227), warm-starting every fit from the previous
``pmean`` (:160-161,:173).  The scalar search is host work; each evaluation is one
device fit.  ``prior_width_sweep`` is the BASELINE.json config-5 shape: many fits
of one problem that differ only in the prior, run back to back on ONE resident
``DeviceProblem`` (only the P-vector / P x P prior precision is re-uploaded).
A batched, hipGraph-captured engine for this row is the next step.
"""
import numpy as np

from .fit import nonlinear_fit


def empbayes_fit(z0, fitargs, p0=None, tol=1e-4, maxit=1000, fitter=nonlinear_fit):
    """-> (fit, z).  ``fitargs(z)`` returns the keyword dict for ``nonlinear_fit`` (or a
    tuple ``(dict, plausibility)``); z has the layout of z0 (number or array)."""
    from scipy.optimize import minimize
    scalar = np.shape(z0) == ()
    z0buf = np.array([z0], float) if scalar else np.asarray(z0, float)
    save = dict(lastz=None, lastp0=p0)

    def convert(zbuf):
        return float(zbuf[0]) if scalar else zbuf

    def minfcn(zbuf):
        z = convert(zbuf)
        args = fitargs(z)
        plaus = 0.0
        if not hasattr(args, 'keys'):
            args, plaus = args
        if save['lastp0'] is not None and 'p0' not in args:
            args = dict(args, p0=save['lastp0'])
        fit = fitter(**args)
        if fit.logGBF is None or np.isnan(fit.logGBF):
            raise ValueError('logGBF undefined - nan')
        save['lastz'] = z
        save['lastp0'] = fit.pmean
        return -fit.logGBF - plaus

    try:
        res = minimize(minfcn, z0buf, tol=tol, options=dict(maxiter=maxit), method='Nelder-Mead')
        z = convert(res.x)
    except ValueError:
        print('*** empbayes_fit warning: null logGBF')
        z = save['lastz']
    args = fitargs(z)
    if not hasattr(args, 'keys'):
        args, _ = args
    if save['lastp0'] is not None and 'p0' not in args:
        args = dict(args, p0=save['lastp0'])
    return fitter(**args), z


def prior_width_sweep(problem, data, model, prior_mean, widths, p0=None, **fitkw):
    """Fit the same data under priors ``prior_mean +- widths[j]`` (each a scalar or a
    P-vector), warm-started, on one resident DeviceProblem.  -> list of fits."""
    from .whiten import Whitening
    fits = []
    last = p0
    pm = np.asarray(prior_mean, float)
    for w in widths:
        sd = np.broadcast_to(np.asarray(w, float), pm.shape).copy()
        wh = problem.wh
        # same data whitening, new (diagonal) prior: refresh the host mirror and the device copy
        wh.prior_mean, wh.prior_prec, wh.prior_dense = pm, 1.0 / sd ** 2, False
        wh.prior_W, wh.prior_sdev = ('diag', 1.0 / sd), sd
        wh.logdet = wh.logdet_data + 2.0 * float(np.sum(np.log(sd)))
        problem.set_prior(pm, wh.prior_prec)
        fit = nonlinear_fit(data=data, model=model, prior=(pm, sd), p0=last, problem=problem, **fitkw)
        fits.append(fit)
        last = fit.pmean
    return fits
