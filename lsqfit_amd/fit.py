"""``nonlinear_fit``: problem setup and result reduction around the fitter plugin.

Host mirror of the numerical content of src/lsqfit/__init__.py:455-737 for the
device path, without gvar objects: data and prior arrive as (mean, error)
arrays, the fit function as a :class:`lsqfit_amd.Model`.

  :539-561   whitening of concat(y, prior)         -> :class:`Whitening`
  :562-565   default p0 (prior mean; mean + sdev/10 where the mean is 0, :1947-1948)
  :574-575   nf = nchiv, dof = nf - P
  :662-664   fit = FITTERS[fitter](p0, nf, chiv, tol=tol, maxit=maxit, **fitterargs)
  :665-682   error, cov, chi2, Q, nit, tol, stopping_criterion, pmean, psdev
  :709-725   logGBF = (-logdet(J^T J) - pdf.logdet - chi2 - dof log 2pi) / 2
"""
import time

import numpy as np

from .fitter import DeviceProblem, mi355x_lm, mi355x_trf
from .whiten import Whitening

FITTERS = {'mi355x_lm': mi355x_lm, 'mi355x_trf': mi355x_trf}
DEFAULTS = dict(tol=1e-8, svdcut=1e-12, maxit=1000, fitter='mi355x_lm')   # __init__.py:100-107


def gammaQ(a, x):
    """Regularised upper incomplete gamma function (src/lsqfit/_scipy.py:16-18)."""
    from scipy.special import gammaincc
    return float(gammaincc(a, x))


def default_p0(prior_mean, prior_sdev):
    pm = np.asarray(prior_mean, float).reshape(-1)
    ps = np.asarray(prior_sdev, float).reshape(-1)
    return np.where(pm != 0.0, pm, pm + 0.1 * ps)


class _Chiv:
    """What the reference hands the plugin as ``f``: p -> whitened residual.  On this
    path it only exists so the plugin call has the reference's shape."""

    def __init__(self, problem):
        self.problem = problem

    def __call__(self, p):
        self.problem.normal(p)
        return self.problem.get_f_data()


class nonlinear_fit(object):
    """data = (x, ymean, yerr) with yerr an sdev vector, a covariance matrix, or
    dict(sdev=..., blocks=[(row0, cov), ...]); prior = (mean, err) likewise;
    model = :class:`lsqfit_amd.Model`."""

    def __init__(self, data=None, model=None, prior=None, p0=None, svdcut=False, tol=None,
                 maxit=None, udata=None, fitter=None, problem=None, linear=None, cross=None, eps=False,
                 noise=False, rng=None, fcn=None, **fitterargs):
        if data is None and udata is None:
            raise ValueError('neither data nor udata is specified')
        if model is None and fcn is None:
            raise ValueError('no fit function (fcn or model) specified')
        if p0 is None and prior is None:
            raise ValueError('neither p0 nor prior is specified')
        self.fcn = fcn
        self.traced = None
        if model is None:
            # the reference's own call form: an ordinary Python function of (x, p) -- or of p alone when the data carry no x
            # (data = (y, yerr) / x False, src/lsqfit/__init__.py:2013-2016) -- with array or dictionary parameters and
            # outputs.  It is CALLED once on tracer arrays and the recording compiled for the device (lsqfit_amd.trace)
            from .trace import flatten_mean_err, trace
            dd = udata if data is None else data
            if len(dd) == 2:
                dd = (False,) + tuple(dd)
            xx, ymean_in, yerr_in = dd
            template = prior[0] if prior is not None else p0
            self.traced = trace(fcn, xx, template, y=ymean_in)
            model = self.traced.model
            ymean_f, yerr_f = flatten_mean_err(ymean_in, yerr_in)
            dd = (self.traced.x, ymean_f, yerr_f)
            if data is None:
                udata = dd
            else:
                data = dd
            if prior is not None:
                prior = flatten_mean_err(prior[0], prior[1])
            if p0 is not None:
                p0 = self.traced.pack_params(p0)
            if fitterargs.get('bounds') is not None:          # (lower, upper) shaped like the parameters, __init__.py:641-655
                lower, upper = fitterargs['bounds']
                fitterargs['bounds'] = (self.traced.pack_params(lower), self.traced.pack_params(upper))
        # src/lsqfit/__init__.py:471-479: neither given -> the default svdcut; eps alone -> no svdcut
        if svdcut is False and eps is False:
            svdcut, eps = DEFAULTS['svdcut'], None
        elif svdcut is False:
            svdcut = None
        elif eps is False:
            eps = None
        tol = DEFAULTS['tol'] if tol is None else tol
        maxit = DEFAULTS['maxit'] if maxit is None else maxit
        self.fitter = DEFAULTS['fitter'] if fitter is None else fitter
        if self.fitter not in FITTERS:
            raise ValueError('unknown fitter: ' + str(self.fitter))
        clock = time.perf_counter
        t0 = clock()
        uncorrelated = data is None
        x, ymean, yerr = udata if uncorrelated else data
        pm, perr = (None, None) if prior is None else prior
        # scipy_least_squares' robust losses (src/lsqfit/_scipy.py:76-79,:147-153) act on every ELEMENT of the whitened residual
        # vector: the prior then travels as rows and every block is whitened in gvar's (eigen) basis (whiten.rows_whitening)
        loss = fitterargs.get('loss', 'linear')
        robust = self.fitter == 'mi355x_trf' and isinstance(loss, str) and loss != 'linear'
        if problem is None:
            if robust:
                if cross is not None or eps is not None or np.any(noise):
                    raise NotImplementedError('a robust loss with cross= / eps / noise')
                from .whiten import rows_whitening
                wh = rows_whitening(ymean, yerr, pm, perr, svdcut=svdcut, udata=uncorrelated)
                problem = DeviceProblem(model, x, wh)
            elif cross is not None:
                # data correlated with the prior (examples/y-noerr.py): ``cross`` is the N x P
                # covariance between y and the prior; concat(y, prior) is whitened as one vector
                # (src/lsqfit/__init__.py:1892-1900)
                if prior is None or uncorrelated:
                    raise ValueError('cross needs data= and prior=')
                from .whiten import joint_whitening
                wh = joint_whitening(ymean, yerr, pm, perr, cross, svdcut=svdcut, eps=eps, noise=noise, rng=rng)
            else:
                wh = Whitening(ymean, yerr, pm, perr, svdcut=svdcut, eps=eps, udata=uncorrelated, noise=noise,
                               rng=rng)
            if not robust:
                problem = DeviceProblem(model, x, wh)
        else:
            wh = problem.wh
            if robust and (wh.has_prior or getattr(wh, 'engine', None) != 'eig'):
                raise ValueError('a robust loss needs a problem built for it (prior as rows, blocks in their eigen basis)')
        self.problem = problem
        self.problem_x = x
        self.whitening = wh
        self.svdcut = svdcut
        self.eps = getattr(wh, 'eps', None)
        self.noise = getattr(wh, 'noise', (False, False))
        self.svdn = wh.nmod
        self.nblocks = wh.nblocks
        self.model = model
        if p0 is None:
            p0 = default_p0(pm, wh.prior_sdev)
        self.p0 = np.array(p0, float).reshape(-1)
        nf = wh.nchiv
        self.dof = nf - self.p0.size
        self._chiv = _Chiv(problem)
        t1 = clock()
        if maxit == 0:
            # src/lsqfit/__init__.py:683-706: no fit -- parameters are the prior (or p0 with infinite
            # errors); chi2 is still evaluated on the device
            self.fitter_results = None
            self.fitterargs = dict(fitterargs)
            self.error = None
            if prior is None:
                self.pmean = self.p0.copy()
                self.psdev = np.full(self.p0.size, np.inf)
                self.cov = np.diag(self.psdev ** 2)
            elif getattr(wh, 'joint', False):      # data correlated with the prior: the joint whitening kept the prior's own block
                self.pmean = np.array(wh.prior_mean_host)
                self.psdev = np.array(wh.prior_sdev)
                self.cov = np.array(wh.prior_cov_host)
            else:
                self.pmean = np.array(wh.prior_mean)
                self.psdev = np.array(wh.prior_sdev)
                perr = np.asarray(perr, float)
                self.cov = perr.copy() if perr.ndim == 2 else np.diag(self.psdev ** 2)
            self.chi2 = problem.chi2(self.pmean)
            self.Q = gammaQ(self.dof / 2., self.chi2 / 2.)
            self.nit, self.tol, self.maxit, self.stopping_criterion, self.description = 0, tol, 0, 0, ''
            if prior is None:
                self.logGBF = None
            else:                                   # :718-725 without J: logdet(cov)
                sign, ld = np.linalg.slogdet(self.cov)
                self.logGBF = 0.5 * (ld - wh.logdet - self.chi2 - self.dof * np.log(2. * np.pi))
            self.time = clock() - t0
            self.time_setup, self.time_fit = t1 - t0, self.time - (t1 - t0)
            return
        if fitterargs.get('bounds') is not None:   # __init__.py:641-655: flattened like p0
            if self.fitter != 'mi355x_trf':
                raise ValueError("bounds need fitter='mi355x_trf'")
            lower, upper = fitterargs['bounds']
            fitterargs['bounds'] = (np.reshape(lower, -1), np.reshape(upper, -1))
        # __init__.py:656-661,:738-787: parameters the fit function is linear in.  The reference
        # projects them out of the function the plugin sees at every evaluation; so does the device
        # (include/lsqfit_amd.h, lsqamd_set_linear; api.hip iterate_varpro)
        self.linear = [] if linear is None else [int(i) for i in np.reshape(linear, -1)]
        if self.linear and (self.fitter != 'mi355x_lm' or fitterargs.get('alg', 'lm') != 'lm'):
            raise ValueError("linear= needs fitter='mi355x_lm' with alg='lm'")
        problem.set_linear(self.linear)
        try:
            fit = FITTERS[self.fitter](self.p0, nf, self._chiv, tol=tol, maxit=maxit, problem=problem,
                                       **fitterargs)
        finally:
            if self.linear:
                problem.set_linear(None)
        self.fitter_results = fit
        self.fitterargs = dict(fitterargs)          # what resampled copies are refitted with (src/lsqfit/__init__.py:1457-1459)
        self.error = fit.error
        self.cov = fit.cov
        self.chi2 = fit.chi2                       # = sum(fit.f**2), computed on the device
        self.Q = gammaQ(self.dof / 2., self.chi2 / 2.)
        self.nit = fit.nit
        self.tol = fit.tol
        self.maxit = maxit
        self.stopping_criterion = fit.stopping_criterion
        self.description = fit.description
        self.pmean = np.array(fit.x)
        self.psdev = np.sqrt(np.diag(fit.cov))
        if prior is None:
            self.logGBF = None
        else:
            self.logGBF = 0.5 * (-fit.logdet_jtj - wh.logdet - self.chi2 - self.dof * np.log(2. * np.pi))
        self.time = clock() - t0
        self.time_setup = t1 - t0
        self.time_fit = self.time - self.time_setup

    @property
    def residuals(self):
        return self.fitter_results.f

    @property
    def p(self):
        """best-fit parameter MEANS in the shape the fit function takes them (array or dictionary; traced fits only --
        ``pmean`` is the flat vector, ``psdev`` / ``cov`` its errors)"""
        return self.pmean if self.traced is None else self.traced.unpack_params(self.pmean)

    @property
    def J(self):
        return self.fitter_results.J

    # -- fit.p with its input correlations ----------------------------------
    def _no_joint(self):
        if getattr(self.whitening, 'joint', False):
            raise NotImplementedError('not available for fits with data-prior cross-correlations')

    def dp_dinputs(self, G=None):
        """``D[a, i] = d pmean[a] / d buf[i]`` for ``buf = concat(y, prior)``: the matrix
        ``_getp`` (src/lsqfit/__init__.py:897-911) turns into the derivatives of ``fit.p``
        (``p[a].der = sum_i D[a,i] buf[i].der``).  ``cov_p = D C D^T``
        (doc/source/lsqfit.rst:105-117).  Computed on the device from the resident whitened
        Jacobian; ``G`` (m x P) returns ``G @ D`` for m derived quantities instead."""
        D = self.problem.dpdy(G)
        wh = self.whitening
        if self.problem.N != wh.n_data and (getattr(wh, 'joint', False) or getattr(wh, 'perm', None) is not None):
            return D                             # a shard of reordered rows: columns in the whitening's order (problem.rows)
        if getattr(wh, 'joint', False):          # device rows are the permuted joint vector
            out = np.empty_like(D)
            out[:, wh.row_src] = D
            return out
        if getattr(wh, 'perm', None) is not None:    # interleaved covariance components: data columns back in order
            out = np.array(D)
            out[:, wh.perm] = D[:, :wh.n_data]
            return out
        return D

    def partial_sdev(self, grads, groups, cov_in):
        """Error budget in the sense of ``gvar.fmt_errorbudget(outputs, inputs)`` as used by
        examples/simple.py:56-61: ``grads`` maps an output name to its gradient d g / d p at
        ``pmean``; ``groups`` maps an input-group name to indices into ``buf``; ``cov_in`` is
        the covariance of ``buf`` (dense, or a 1-d array of variances).  Returns
        {(output, group): partial standard deviation}."""
        names = list(grads)
        GD = self.dp_dinputs(np.array([np.asarray(grads[g], float) for g in names]))
        cov_in = np.asarray(cov_in, float)
        out = {}
        for k, g in enumerate(names):
            for name, idx in groups.items():
                idx = np.asarray(idx, int)
                d = GD[k, idx]
                var = float(d @ cov_in[np.ix_(idx, idx)] @ d) if cov_in.ndim == 2 else float(np.sum(d * d * cov_in[idx]))
                out[g, name] = float(np.sqrt(max(var, 0.0)))
        return out

    # -- simulated / bootstrap copies ------------------------------------------
    def simulated_fits(self, n, pexact=None, add_priornoise=False, seed=0, **kw):
        """``simulated_fit_iter`` (src/lsqfit/__init__.py:1391-1469) as one device batch."""
        from .resample import simulated_fits
        return simulated_fits(self, n, pexact, add_priornoise, seed, **kw)

    def bootstrapped_fits(self, n, seed=0, **kw):
        """``bootstrapped_fit_iter`` (src/lsqfit/__init__.py:1548-1642) as one device batch."""
        from .resample import bootstrapped_fits
        return bootstrapped_fits(self, n, seed, **kw)

    # -- chi2(p) / pdf(p) at arbitrary points -----------------------------------
    def dchi2(self, p):
        """``chi**2(p) - fit.chi2`` (``_fit_dchi2``, src/lsqfit/__init__.py:1648-1670); ``p`` of
        shape (P,) -> float, (m, P) -> array of m values evaluated in one device pass (the
        lbatch layout of ``vegas_fit._chiv``, src/lsqfit/_extras.py:2467-2486)."""
        p = np.asarray(p, float)
        c = self.problem.chi2_points(p.reshape(-1, self.pmean.size)) - self.chi2
        return float(c[0]) if p.ndim == 1 else c

    def pdf(self, p):
        """``exp(-(chi**2(p) - fit.chi2)/2)`` (``_fit_pdf``, src/lsqfit/__init__.py:1803-1816)."""
        return np.exp(-0.5 * self.dchi2(p))

    @property
    def pdf_lognorm(self):
        """``fit.pdf.lognorm`` (src/lsqfit/__init__.py:1806-1809)."""
        return 0.5 * (self.whitening.logdet + np.log(2 * np.pi) * (self.dof + self.pmean.size)) + self.chi2 / 2
