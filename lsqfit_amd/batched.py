"""``BatchedFits``: many fits of one shape in lockstep on the device (C ABI ``lsqamdb_*``).

Host counterpart of running ``lsqfit.nonlinear_fit`` in a Python loop over priors /
starting points (the inner loop of ``lsqfit.empbayes_fit``, src/lsqfit/_extras.py:153-174):
the fits share the model, ``x`` and the data covariance (sdev vector, covariance matrix or
``dict(sdev=, blocks=)`` -- whitened exactly as in :class:`lsqfit_amd.Whitening`) and differ in
their diagonal priors, ``p0`` and -- ``ymean`` of shape (n_fits, N) -- the data means
(simulated / bootstrap refits, src/lsqfit/__init__.py:1391-1469,1548-1642).  Result arrays
have a leading batch dimension.
"""
import ctypes as C
import time

import numpy as np

from . import _lib
from .fit import gammaQ
from .fitter import _SCALERS, normalize_tol
from .models import MODEL_IDENTITY, MODEL_TAPE


class BatchedFits:
    def __init__(self, model, x, ymean, ysdev, prior_mean, prior_sdev, device=None, svdcut=1e-12,
                 whitening=None, n_fits=None, prior_prec=None, prior_logdet=None, rows_permuted=False):
        """``prior_sdev``: per-fit diagonal priors ([n_fits, P] or broadcastable).  Alternatively
        ``prior_prec`` (P x P) + ``prior_logdet``: ONE correlated prior covariance shared by all
        fits (its inverse and log-determinant, e.g. from :class:`Whitening`), per-fit means only."""
        import torch
        from .whiten import Whitening
        if not torch.cuda.is_available():
            raise RuntimeError('lsqfit_amd: no MI355X visible; the batched fitter has no CPU path')
        self.lib = lib = _lib.load()
        self.model = model
        ymean = np.ascontiguousarray(ymean, np.float64)
        ymeans = ymean if ymean.ndim == 2 else None
        ymean = ymean[0] if ymean.ndim == 2 else ymean.reshape(-1)
        if whitening is None:
            yerr = ysdev if isinstance(ysdev, dict) else np.asarray(ysdev, np.float64)
            if not isinstance(yerr, dict) and yerr.ndim < 2:
                yerr = np.broadcast_to(yerr, ymean.shape)
            whitening = Whitening(ymean, yerr, svdcut=svdcut)
        self.wh = wh = whitening
        # covariance components that interleave: the whitening made them contiguous by a row permutation; x and every vector of
        # data means follow it (resample.refit hands both over already reordered: rows_permuted=True)
        self._perm = None if rows_permuted else getattr(wh, 'perm', None)
        if self._perm is not None:
            if model.kind == MODEL_IDENTITY:
                raise ValueError('identity model with interleaved covariance components: reorder the data')
            x = np.asarray(x, np.float64).reshape(ymean.size, model.n_x)[self._perm]
            ymean = np.ascontiguousarray(ymean[self._perm])
        row0, size, modes, tri, wt = wh.block_arrays()
        self.has_prior = prior_mean is not None
        self.prior_dense = prior_prec is not None
        if self.has_prior:
            pm = np.ascontiguousarray(prior_mean, np.float64)
            if pm.ndim != 2 or pm.shape[1] != model.n_param:
                raise ValueError('prior_mean must be [n_fits, n_param]')
            if self.prior_dense:
                ps = None
                self.prior_prec = np.ascontiguousarray(prior_prec, np.float64)
                if self.prior_prec.shape != (model.n_param, model.n_param):
                    raise ValueError('prior_prec must be [n_param, n_param]')
                self.prior_logdet = float(prior_logdet) if prior_logdet is not None else \
                    -float(np.linalg.slogdet(self.prior_prec)[1])
            else:
                ps = np.ascontiguousarray(np.broadcast_to(np.asarray(prior_sdev, np.float64), pm.shape))
                if np.any(ps <= 0):
                    raise ValueError('some priors have zero standard deviations')
            n_fits = pm.shape[0]
        else:
            pm = ps = None
            if n_fits is None:
                n_fits = ymeans.shape[0] if ymeans is not None else None
            if n_fits is None:
                raise ValueError('n_fits is needed when there is no prior')
        self.B, self.P, self.N = int(n_fits), model.n_param, ymean.size
        if ymeans is not None and ymeans.shape[0] != self.B:
            raise ValueError('ymean must be [N] or [n_fits, N]')
        self.prior_mean, self.prior_sdev = pm, ps
        cfg = _lib.Config(abi_version=_lib.ABI_VERSION, model=model.kind, n_data=self.N, n_param=self.P,
                          n_x=model.n_x, has_prior=int(self.has_prior), prior_dense=int(self.prior_dense), n_blocks=len(size),
                          max_block=int(size.max()) if len(size) else 0, sum_block_sq=int(np.sum(size * size)),
                          want_jacobian_out=0, n_batch=self.B)
        t0 = time.perf_counter()
        nbytes = lib.lsqamdb_workspace_bytes(C.byref(cfg), self.B)
        if nbytes == 0:
            raise ValueError('lsqfit_amd: unsupported batched problem shape/model')
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.workspace = torch.empty(nbytes + 512, dtype=torch.uint8, device=self.device)
        base = self.workspace.data_ptr()
        off = (-base) % 256
        h = C.c_void_p()
        rc = lib.lsqamdb_create(C.byref(cfg), self.B, C.c_void_p(base + off), nbytes,
                                C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream), C.byref(h))
        if rc != 0:
            raise RuntimeError('lsqfit_amd: lsqamdb_create failed (%s)' % _lib.ERRORS.get(rc, rc))
        self.h = h
        if model.kind != MODEL_IDENTITY:
            xa = np.ascontiguousarray(np.asarray(x, np.float64).reshape(self.N, model.n_x))
            self._check(lib.lsqamdb_set_x(h, _lib.dptr(xa), self.N, model.n_x), 'set_x')
        if model.kind == MODEL_TAPE:
            code = np.ascontiguousarray(model.tape, np.int32)
            consts = np.ascontiguousarray(model.consts, np.float64)
            self._check(lib.lsqamdb_set_tape(h, code.ctypes.data_as(C.POINTER(C.c_int32)), code.size,
                                             _lib.dptr(consts), consts.size), 'set_tape')
        wd = np.ascontiguousarray(wh.wdiag)
        self._check(lib.lsqamdb_set_data(h, _lib.dptr(ymean), _lib.dptr(wd)), 'set_data')
        if len(size):
            self._check(lib.lsqamdb_set_blocks(
                h, len(size), row0.ctypes.data_as(C.POINTER(C.c_int64)), size.ctypes.data_as(C.POINTER(C.c_int64)),
                modes.ctypes.data_as(C.POINTER(C.c_int64)), tri.ctypes.data_as(C.POINTER(C.c_int32)),
                _lib.anyptr(wt)), 'set_blocks')
        if ymeans is not None:
            self.set_data_means(ymeans)
        if self.has_prior and self.prior_dense:
            self.set_prior_means(pm)
        elif self.has_prior:
            self.set_priors(pm, ps)
        self.t_setup = time.perf_counter() - t0

    def _check(self, rc, what):
        if rc < 0:
            msg = self.lib.lsqamdb_last_error(self.h)
            raise RuntimeError('lsqfit_amd: batched %s failed (%s): %s' % (
                what, _lib.ERRORS.get(rc, rc), msg.decode() if msg else ''))

    def timing(self, on=True):
        """HIP-event timers around the batched J^T J launch and the batched factorisation (rounds then run eagerly)."""
        self._check(self.lib.lsqamdb_timing_enable(self.h, int(bool(on))), 'timing_enable')

    def timings(self):
        """-> {'syrk': (total ms, launches), 'cholesky': (...)} since ``timing(True)``"""
        out = {}
        for name, which in (('syrk', 3), ('cholesky', 6)):
            ms, n = C.c_double(0.0), C.c_int64(0)
            self._check(self.lib.lsqamdb_timing_get(self.h, which, C.byref(ms), C.byref(n)), 'timing_get')
            out[name] = (ms.value, n.value)
        return out

    def set_data_means(self, ymeans):
        """New data means, one row per fit (same covariance): shape (n_fits, N)."""
        ymeans = np.ascontiguousarray(ymeans, np.float64)
        if ymeans.shape != (self.B, self.N):
            raise ValueError('ymeans must be [n_fits, N]')
        if self._perm is not None:
            ymeans = np.ascontiguousarray(ymeans[:, self._perm])
        self._check(self.lib.lsqamdb_set_data_means(self.h, _lib.dptr(ymeans)), 'set_data_means')

    def set_prior_means(self, mean):
        """Dense shared prior: new per-fit means (the shared precision is re-sent with them)."""
        mean = np.ascontiguousarray(mean, np.float64)
        self.prior_mean = mean
        self._check(self.lib.lsqamdb_set_priors(self.h, _lib.dptr(mean), _lib.anyptr(self.prior_prec)), 'set_priors')

    def set_priors(self, mean, sdev):
        if self.prior_dense:
            raise ValueError('this batch has a dense shared prior: use set_prior_means')
        mean = np.ascontiguousarray(mean, np.float64)
        sdev = np.ascontiguousarray(np.broadcast_to(np.asarray(sdev, np.float64), mean.shape))
        prec = np.ascontiguousarray(1.0 / sdev ** 2)
        self.prior_mean, self.prior_sdev = mean, sdev
        self._check(self.lib.lsqamdb_set_priors(self.h, _lib.dptr(mean), _lib.anyptr(prec)), 'set_priors')

    def run(self, p0=None, tol=1e-8, maxit=1000, scaler='more', factor_up=3.0, factor_down=2.0,
            use_graph=True, covariance=True):
        """-> dict of arrays: pmean[B,P], chi2, dof, Q, logGBF, nit, stopping_criterion, status,
        nfev (+ psdev[B,P] and cov(b) when ``covariance``)."""
        B, P = self.B, self.P
        if p0 is None and not self.has_prior:
            raise ValueError('neither p0 nor prior is specified')
        if p0 is None and self.prior_dense:
            p0 = self.prior_mean
        elif p0 is None:
            p0 = np.where(self.prior_mean != 0.0, self.prior_mean, self.prior_mean + 0.1 * self.prior_sdev)
        p0 = np.ascontiguousarray(np.broadcast_to(np.asarray(p0, np.float64), (B, P)))
        xtol, gtol, ftol = normalize_tol(tol)
        opt = _lib.Options(xtol=xtol, gtol=gtol, ftol=ftol, maxit=int(maxit), scaler=_SCALERS[scaler], solver=0,
                           trs=0, factor_up=factor_up, factor_down=factor_down, avmax=0.75)
        self._check(self.lib.lsqamdb_set_options(self.h, C.byref(opt)), 'set_options')
        summ = (_lib.Summary * B)()
        t0 = time.perf_counter()
        self._check(self.lib.lsqamdb_run(self.h, _lib.dptr(p0), summ, int(bool(use_graph))), 'run')
        t_run = time.perf_counter() - t0
        x = np.empty(B * P)
        self._check(self.lib.lsqamdb_get_x(self.h, _lib.dptr(x), x.size), 'get_x')
        out = dict(pmean=x.reshape(B, P),
                   chi2=np.array([s.chi2 for s in summ]), nit=np.array([s.nit for s in summ]),
                   nfev=np.array([s.nfev for s in summ]), njev=np.array([s.njev for s in summ]), status=np.array([s.status for s in summ]),
                   stopping_criterion=np.array([s.stopping_criterion for s in summ]),
                   rounds=int(self.lib.lsqamdb_rounds(self.h)), graph_rounds=int(summ[0].t_setup_ms),
                   time=t_run, device_ms=float(summ[0].t_run_ms))
        # nf - P: (kept data modes + P) - P with a prior, kept data modes - P without
        dof = self.wh.nchiv_data - (0 if self.has_prior else P)
        out['dof'] = dof
        from scipy.special import gammaincc
        out['Q'] = np.asarray(gammaincc(dof / 2., out['chi2'] / 2.), float)           # (gammaQ, all fits at once)
        if covariance:
            ld = np.empty(B)
            self._check(self.lib.lsqamdb_covariance(self.h, _lib.dptr(ld), B), 'covariance')
            out['logdet_jtj'] = ld
            if self.has_prior:
                logdet_c = self.wh.logdet_data + (self.prior_logdet if self.prior_dense else
                                                  2.0 * np.sum(np.log(self.prior_sdev), axis=1))
                out['logGBF'] = 0.5 * (-ld - logdet_c - out['chi2'] - dof * np.log(2. * np.pi))
            else:
                out['logGBF'] = None              # src/lsqfit/__init__.py:711-712
            out['psdev'] = np.sqrt(np.einsum('bii->bi', self.cov_all())) if B * P * P <= 1 << 26 else None
        return out

    def cov(self, b):
        out = np.empty(self.P * self.P)
        self._check(self.lib.lsqamdb_get_cov(self.h, int(b), _lib.dptr(out), out.size), 'get_cov')
        return out.reshape(self.P, self.P)

    def cov_all(self):
        """-> [B, P, P]: every fit's covariance in one copy."""
        out = np.empty(self.B * self.P * self.P)
        self._check(self.lib.lsqamdb_get_cov_all(self.h, _lib.dptr(out), out.size), 'get_cov_all')
        return out.reshape(self.B, self.P, self.P)

    def close(self):
        if getattr(self, 'h', None):
            self.lib.lsqamdb_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
