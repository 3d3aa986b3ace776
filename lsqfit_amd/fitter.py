"""``mi355x_lm``: the fitter plugin (host mirror of ``lsqfit.gsl_multifit``).

Same constructor contract as the reference's plugins
(src/lsqfit/__init__.py:662-664):

    fit = FITTERS[name](p0, nf, chiv, tol=tol, maxit=maxit, **fitterargs)

and the same result attributes (``x cov f J nit tol stopping_criterion error
description results``, read at :665-679).  The callable ``f`` the reference
passes is ignored -- a device cannot call Python; the problem itself arrives
through the ``problem=`` fitter argument (a :class:`DeviceProblem`), using the
reference's own pass-through of unknown keyword arguments (:505-515).
"""
import ctypes as C
import time
import warnings

import numpy as np

from . import _lib
from .models import MODEL_IDENTITY, MODEL_TAPE, Model
from .whiten import Whitening

_STREAMS = {}     # device index -> idle streams of closed problems (creating one costs tens of microseconds: a small fit's worth)

_SCALERS = dict(more=0, levenberg=1, marquardt=2)
_SOLVERS = dict(cholesky=0, qr=1, svd=1)                               # _gsl.pyx:646-653; svd -> the QR-grade route
_ALGS = dict(lm=0, lmaccel=1, dogleg=2, ddogleg=3, subspace2D=4)      # _gsl.pyx:622-635
_BOUNDED = dict(trf=5, dogbox=6, minpack_lm=7)                        # LSQAMD_TRS_TRF, _DOGBOX, _MINPACK_LM


def _check(lib, h, rc, what):
    if rc < 0:
        msg = lib.lsqamd_last_error(h)
        raise RuntimeError('lsqfit_amd: %s failed (%s): %s' % (
            what, _lib.ERRORS.get(rc, rc), msg.decode() if msg else ''))
    return rc


def describe(alg, scaler, solver, avmax=0.75):
    """``description`` of an mi355x_lm fit (src/lsqfit/_gsl.pyx:611-618 prints ``methods = alg/scaler/solver``): names what RAN.
    gsl's ``'svd'`` solver (:650-651) has no counterpart here -- such a fit runs the QR-grade route (qr.hip) and says so."""
    ran = 'qr' if solver == 'svd' else solver
    d = 'methods = {}/{}/{}'.format(alg, scaler, ran)
    if solver == 'svd':
        d += "    (solver 'svd' runs as 'qr')"
    if alg == 'lmaccel':
        d += '    avmax = {}'.format(avmax)          # _gsl.pyx:617-618
    return d


def normalize_tol(tol):
    """src/lsqfit/_gsl.pyx:594-603."""
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


def _cov_failed(fit, summary):
    """A rank-deficient J^T J at the end point must not look like a clean fit."""
    fit.cov_dropped = max(0, int(summary.cov_status))   # > 0: rank-deficient Jacobian, the reference's truncated inverse
    if summary.cov_status == -9:      # LSQAMD_EINACCURATE: delivered, but not to the accuracy the route promises
        warnings.warn('lsqfit_amd: the orthogonalisation behind the covariance did not converge '
                      '(nearly rank-deficient Jacobian); covariance and logGBF may be inaccurate')
    elif summary.cov_status < 0:
        msg = 'J^T J is not positive definite at the solution: covariance and logGBF are undefined'
        warnings.warn('lsqfit_amd: ' + msg)
        if fit.error is None:
            fit.error = msg


def _problem_from_residual(x0, n, f):
    """No ``problem=``: the plugin was called exactly as lsqfit calls its fitters, ``FITTERS[name](p0, nf, chiv, ...)``
    (src/lsqfit/__init__.py:662-664).  ``f`` -- lsqfit's ``chiv``: p -> whitened residual, prior rows included -- is called
    ONCE on tracer numbers (the way ``_c_df`` calls it on GVars, src/lsqfit/_gsl.pyx:748-750) and the recording becomes the
    device model; the residual is already whitened, so the device sees unit weights and no separate prior."""
    if f is None or not callable(f):
        raise ValueError("the MI355X fitters need the residual function f (as lsqfit hands it over) or problem=DeviceProblem(...)")
    from .trace import trace_residual
    from .whiten import Whitening
    P = int(np.size(x0))
    tr = trace_residual(f, P)
    if n is not None and int(n) != tr.n_rows:
        raise ValueError('n = %d but the residual function returned %d values' % (int(n), tr.n_rows))
    wh = Whitening(np.zeros(tr.n_rows), np.ones(tr.n_rows), svdcut=None)
    return DeviceProblem(tr.model, tr.x, wh)


class DeviceProblem:
    """Model + data + whitening resident on one GPU (one C-ABI handle).

    ``rows=(a, b)`` restricts the data rows to a shard (whole covariance blocks);
    the prior stays replicated."""

    def __init__(self, model, x, whitening, rows=None, device=None, reduce_hook=None,
                 adds_prior=True):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError('lsqfit_amd: no MI355X visible (torch.cuda.is_available() is False); '
                               'the fitter has no CPU path')
        self.lib = _lib.load()
        self.model = model
        self.wh = whitening
        a, b = (0, whitening.n_data) if rows is None else rows
        joint = getattr(whitening, 'joint', False)     # concat(y, prior) whitened as one vector
        if joint:
            if model.kind == MODEL_IDENTITY:
                raise ValueError('data-prior cross-correlations: x-dependent models only')
            xe = np.zeros((whitening.n_data, model.n_x))
            xe[whitening.model_rows] = np.asarray(x, np.float64).reshape(whitening.n_model, model.n_x)[
                whitening.row_src[whitening.model_rows]]
            x = xe
        self.perm = None if joint else getattr(whitening, 'perm', None)
        if self.perm is not None:
            # interleaved covariance components: the whitening reordered the rows; x follows (a shard is a range of the
            # REORDERED rows: whole components, see dist.shard_rows)
            if model.kind == MODEL_IDENTITY:
                raise ValueError('identity model with interleaved covariance components: reorder the data')
            x = np.asarray(x, np.float64).reshape(whitening.n_data, model.n_x)[self.perm]
        self.rows = (a, b)
        N = b - a
        P = model.n_param
        if whitening.has_prior and whitening.prior_mean.size != P:
            raise ValueError('prior has %d entries, model has %d parameters'
                             % (whitening.prior_mean.size, P))
        if model.kind == MODEL_IDENTITY and whitening.n_data != P:
            raise ValueError('identity model needs len(y) == len(p)')
        row0, size, modes, tri, wt = whitening.block_arrays((a, b))
        cfg = _lib.Config(abi_version=_lib.ABI_VERSION, model=model.kind, n_data=N, n_param=P,
                          n_x=model.n_x, has_prior=int(whitening.has_prior),
                          prior_dense=int(whitening.prior_dense), n_blocks=len(size),
                          max_block=int(size.max()) if len(size) else 0,
                          sum_block_sq=int(np.sum(size * size)), want_jacobian_out=1, n_batch=1,
                          tape_len=0 if model.tape is None else int(len(model.tape)))
        self.cfg = cfg
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        t0 = time.perf_counter()
        nbytes = self.lib.lsqamd_workspace_bytes(C.byref(cfg))
        if nbytes == 0:
            raise ValueError('lsqfit_amd: unsupported problem shape/model')
        self.workspace = torch.empty(nbytes + 512, dtype=torch.uint8, device=self.device)
        base = self.workspace.data_ptr()
        self._ws_off = (-base) % 256
        # a stream of its own: the caller's current stream may be the legacy default stream, which cannot be
        # captured (the LM step of small problems is replayed from graphs); every entry point of the
        # library synchronises its stream before handing results back, so callers see no difference
        idle = _STREAMS.get(self.device.index)
        self.stream = idle.pop() if idle else torch.cuda.Stream(device=self.device)
        # device-made whitening weights are complete when the whitening hands them over (made on the package's side stream
        # and waited for: _lib.side_stream); work the CALLER queued on a stream of its own is ordered before ours.  The legacy
        # default stream is never touched -- recording an event on it collides with any graph capture another thread has
        # open (include/lsqfit_amd.h, "threads")
        self._after_callers_stream()
        h = C.c_void_p()
        rc = self.lib.lsqamd_create(C.byref(cfg), C.c_void_p(base + self._ws_off), nbytes,
                                    C.c_void_p(self.stream.cuda_stream), C.byref(h))
        if rc != 0:
            raise RuntimeError('lsqfit_amd: lsqamd_create failed (%s)' % _lib.ERRORS.get(rc, rc))
        self.h = h
        self.N, self.P = N, P
        lib = self.lib
        if model.kind != MODEL_IDENTITY:
            xa = np.ascontiguousarray(np.asarray(x, np.float64).reshape(whitening.n_data, model.n_x)[a:b])
            _check(lib, h, lib.lsqamd_set_x(h, _lib.dptr(xa), N, model.n_x), 'set_x')
        if model.kind == MODEL_TAPE and getattr(model, 'programs', None):
            # one formula per row range (models.piecewise): the ranges clipped to this handle's rows
            if joint or self.perm is not None:
                raise ValueError('piecewise models: data whose covariance ties rows of different ranges into '
                                 'interleaved components (or to the prior) is not supported')
            if sum(n for n, _ in model.programs) != whitening.n_data:
                raise ValueError('piecewise model covers %d rows, the data has %d'
                                 % (sum(n for n, _ in model.programs), whitening.n_data))
            prow0, codes, r = [0], [], 0
            for n, code in model.programs:
                lo, hi = max(r, a), min(r + n, b)
                r += n
                if hi > lo:
                    codes.append(np.asarray(code, np.int32))
                    prow0.append(hi - a)
            if not codes:                                   # a shard without rows: any formula, no rows
                codes, prow0 = [np.asarray(model.programs[0][1], np.int32)], [0, 0]
            code = np.ascontiguousarray(np.concatenate(codes), np.int32)
            off = np.ascontiguousarray(np.concatenate([[0], np.cumsum([c.size for c in codes])]), np.int32)
            prow0 = np.ascontiguousarray(prow0, np.int64)
            consts = np.ascontiguousarray(model.consts, np.float64)
            _check(lib, h, lib.lsqamd_set_tape_programs(
                h, len(codes), prow0.ctypes.data_as(C.POINTER(C.c_int64)), code.ctypes.data_as(C.POINTER(C.c_int32)),
                off.ctypes.data_as(C.POINTER(C.c_int32)), _lib.dptr(consts), consts.size), 'set_tape_programs')
        elif model.kind == MODEL_TAPE:
            code = np.ascontiguousarray(model.tape, np.int32)
            consts = np.ascontiguousarray(model.consts, np.float64)
            _check(lib, h, lib.lsqamd_set_tape(h, code.ctypes.data_as(C.POINTER(C.c_int32)), code.size,
                                               _lib.dptr(consts), consts.size), 'set_tape')
        ym = np.ascontiguousarray(whitening.ymean[a:b])
        if model.kind == MODEL_IDENTITY:
            # identity rows address p by GLOBAL row index: only unsharded use is supported
            if (a, b) != (0, whitening.n_data):
                raise ValueError('identity model cannot be row-sharded')
        wd = np.ascontiguousarray(whitening.wdiag[a:b])
        _check(lib, h, lib.lsqamd_set_data(
            h, _lib.dptr(ym), _lib.dptr(wd), len(size),
            row0.ctypes.data_as(C.POINTER(C.c_int64)), size.ctypes.data_as(C.POINTER(C.c_int64)),
            modes.ctypes.data_as(C.POINTER(C.c_int64)), tri.ctypes.data_as(C.POINTER(C.c_int32)),
            _lib.anyptr(wt)), 'set_data')
        del wt
        if whitening.has_prior:
            dev = getattr(whitening, 'prior_prec_dev', None)      # made on the device: stays there
            self.set_prior(whitening.prior_mean, whitening.prior_prec if dev is None else dev)
        if joint:
            rp = np.ascontiguousarray(whitening.row_param[a:b], np.int32)
            _check(lib, h, lib.lsqamd_set_param_rows(h, rp.ctypes.data_as(C.POINTER(C.c_int32))), 'set_param_rows')
        self._reduce_cb = None
        if reduce_hook is not None:
            self.set_reduce(reduce_hook)
        lib.lsqamd_set_adds_prior(h, int(bool(adds_prior)))
        self.t_setup = time.perf_counter() - t0

    # -- knobs ------------------------------------------------------------------------
    def _after_callers_stream(self):
        import torch
        cur = torch.cuda.current_stream(self.device)
        if cur.cuda_stream != 0:
            self.stream.wait_stream(cur)

    def set_prior(self, mean, prec):
        """Prior mean (P) and precision: P entries (diagonal) or P x P (dense).  A diagonal given
        to a handle created with a dense prior is expanded; the reverse is refused (the library
        copies exactly what the handle's config promises)."""
        P = self.P
        mean = np.ascontiguousarray(mean, np.float64).reshape(-1)
        if hasattr(prec, 'data_ptr'):                # a CUDA tensor (lsqamd_whiten_blocks' inv(C))
            if not (self.cfg.prior_dense and tuple(prec.shape) == (P, P) and prec.is_contiguous()):
                raise ValueError('a device-resident prior precision must be a contiguous %d x %d tensor' % (P, P))
        else:
            prec = np.ascontiguousarray(prec, np.float64)
        if mean.size != P:
            raise ValueError('prior mean has %d entries, the model has %d parameters' % (mean.size, P))
        if hasattr(prec, 'data_ptr'):
            self._after_callers_stream()             # produced on the caller's stream (if it has one), copied on ours
        elif self.cfg.prior_dense:
            if prec.size == P:
                prec = np.ascontiguousarray(np.diag(prec.reshape(-1)))
            elif prec.shape != (P, P):
                raise ValueError('prior precision must have %d or %d x %d entries' % (P, P, P))
        elif prec.size != P:
            raise ValueError('this problem was created with a diagonal prior: the precision must have '
                             '%d entries, got %s' % (P, prec.shape))
        _check(self.lib, self.h, self.lib.lsqamd_set_prior(self.h, _lib.dptr(mean), _lib.anyptr(prec)), 'set_prior')

    def set_ymean(self, ymean):
        """New data means for this problem's rows (same covariance)."""
        if self.perm is not None and np.size(ymean) == self.wh.n_data:
            ymean = np.asarray(ymean, np.float64).reshape(-1)[self.perm]
        ymean = np.ascontiguousarray(np.asarray(ymean, np.float64).reshape(-1)[self.rows[0]:self.rows[1]]
                                     if np.size(ymean) == self.wh.n_data else ymean, np.float64)
        if ymean.size != self.N:
            raise ValueError('ymean has %d entries, expected %d' % (ymean.size, self.N))
        _check(self.lib, self.h, self.lib.lsqamd_set_ymean(self.h, _lib.dptr(ymean)), 'set_ymean')

    def fcn(self, p):
        """Unwhitened model values f(x; p) for this problem's rows."""
        p = np.ascontiguousarray(p, np.float64)
        out = np.empty(self.N)
        _check(self.lib, self.h, self.lib.lsqamd_eval_fcn(self.h, _lib.dptr(p), _lib.dptr(out), out.size), 'eval_fcn')
        if self.perm is not None and self.N == self.wh.n_data:
            back = np.empty_like(out)
            back[self.perm] = out
            return back
        return out          # (a shard of reordered rows: in the whitening's order, rows self.rows of wh.perm)

    def set_reduce(self, hook):
        """hook(dev_ptr:int, count:int) -> None sums count doubles over ranks in place."""
        def cb(user, ptr, count):
            try:
                hook(int(ptr), int(count))
                return 0
            except Exception as e:      # never unwind through the C frames
                self._reduce_error = e
                return 1
        self._reduce_error = None
        self._reduce_cb = _lib.REDUCE_FN(cb)
        _check(self.lib, self.h, self.lib.lsqamd_set_reduce(self.h, self._reduce_cb, None), 'set_reduce')

    # -- in-library RCCL communicator (include/lsqfit_amd.h, lsqamd_comm_*) ------------------
    def comm_unique_id(self):
        """bytes: the id rank 0 creates and ships to the other ranks."""
        buf = C.create_string_buffer(_lib.COMM_ID_BYTES)
        rc = self.lib.lsqamd_comm_unique_id(buf, _lib.COMM_ID_BYTES)
        if rc != 0:
            raise RuntimeError('lsqfit_amd: comm_unique_id failed (%s)' % _lib.ERRORS.get(rc, rc))
        return buf.raw

    def comm_init(self, uid, rank, nranks):
        """Join the communicator (collective over all ranks); afterwards the sums of the sharded
        fit run as RCCL reduce-scatter + all-gather on this handle's stream."""
        uid = bytes(uid)
        _check(self.lib, self.h, self.lib.lsqamd_comm_init(self.h, uid, len(uid), int(rank), int(nranks)),
               'comm_init')

    def comm_stats(self):
        """(milliseconds the handle's communicator took to create, handles of this process sharing it)"""
        ms, n = C.c_double(0.0), C.c_int32(0)
        self.lib.lsqamd_comm_stats(self.h, C.byref(ms), C.byref(n))
        return ms.value, n.value

    def comm_info(self):
        r, n = C.c_int32(), C.c_int32()
        self.lib.lsqamd_comm_info(self.h, C.byref(r), C.byref(n))
        return r.value, n.value

    def comm_destroy(self):
        self.lib.lsqamd_comm_destroy(self.h)

    def view(self, dev_ptr, count):
        """float64 torch view of a region of the workspace (what the reduce hook sums)."""
        from .dist import WorkspaceView
        # the region is inside the handle's workspace, or inside a scratch tensor a running call
        # (chi2_points) handed to the library
        cache = self.__dict__.setdefault('_view_cache', {})
        hit = cache.get((int(dev_ptr), int(count)))      # the same two regions recur every LM step
        if hit is not None:
            return hit
        for t in [self.workspace] + list(getattr(self, '_aux_tensors', [])) + \
                [w for w in [getattr(self, '_qr_work', None)] if w is not None]:
            base = t.data_ptr()
            if base <= int(dev_ptr) and int(dev_ptr) + 8 * int(count) <= base + t.numel():
                v = WorkspaceView(t)(dev_ptr, count)
                if t is self.workspace and len(cache) < 8:
                    cache[(int(dev_ptr), int(count))] = v
                return v
        raise ValueError('pointer outside the workspace')

    def set_options(self, tol, maxit, scaler='more', factor_up=3.0, factor_down=2.0, alg='lm', avmax=0.75,
                    solver='cholesky'):
        xtol, gtol, ftol = normalize_tol(tol)
        if solver not in _SOLVERS:
            raise ValueError('unkown solver ' + str(solver))           # _gsl.pyx:652-653
        if _SOLVERS[solver] == 1 and getattr(self, '_qr_work', None) is None:
            import torch                                                # scratch of the QR-grade covariance
            nbytes = self.lib.lsqamd_qr_work_bytes(self.h)
            self._qr_work = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            _check(self.lib, self.h, self.lib.lsqamd_set_qr_work(self.h, C.c_void_p(self._qr_work.data_ptr()), nbytes),
                   'set_qr_work')
        if scaler not in _SCALERS:
            raise ValueError('unkown scaler ' + str(scaler))
        if alg not in _BOUNDED and alg not in _ALGS:
            raise ValueError('unkown algorithm ' + str(alg))          # _gsl.pyx:634-635
        opt = _lib.Options(xtol=xtol, gtol=gtol, ftol=ftol, maxit=int(maxit), scaler=_SCALERS[scaler],
                           solver=_SOLVERS[solver], trs=_BOUNDED[alg] if alg in _BOUNDED else _ALGS[alg], factor_up=factor_up,
                           factor_down=factor_down, avmax=avmax)
        _check(self.lib, self.h, self.lib.lsqamd_set_options(self.h, C.byref(opt)), 'set_options')

    def set_bounds(self, bounds=None):
        """(lower, upper) for the 'trf' method, each a scalar or length-P array (+-inf = open);
        None clears them (src/lsqfit/__init__.py:641-655)."""
        if bounds is None:
            rc = self.lib.lsqamd_set_bounds(self.h, None, None)
        else:
            if len(bounds) != 2:
                raise ValueError('`bounds` must contain 2 elements.')
            lo = np.ascontiguousarray(np.broadcast_to(np.asarray(bounds[0], np.float64).reshape(-1), (self.P,)))
            hi = np.ascontiguousarray(np.broadcast_to(np.asarray(bounds[1], np.float64).reshape(-1), (self.P,)))
            rc = self.lib.lsqamd_set_bounds(self.h, _lib.dptr(lo), _lib.dptr(hi))
        if rc == -1:
            raise ValueError(self.lib.lsqamd_last_error(self.h).decode())
        _check(self.lib, self.h, rc, 'set_bounds')

    def set_linear(self, index=None):
        """Parameters the fit function is linear in (nonlinear_fit's ``linear=``,
        src/lsqfit/__init__.py:738-787); None or empty clears."""
        idx = np.ascontiguousarray([] if index is None else index, np.int32).reshape(-1)
        rc = self.lib.lsqamd_set_linear(self.h, idx.ctypes.data_as(C.POINTER(C.c_int32)), idx.size)
        if rc == -1:
            raise ValueError(self.lib.lsqamd_last_error(self.h).decode() or 'set_linear: bad index')
        _check(self.lib, self.h, rc, 'set_linear')

    def qr_info(self):
        """(orthogonalisation passes, max |Q^T Q - I| before the last factor) of the last
        QR-grade covariance."""
        n, d = C.c_int32(), C.c_double()
        self.lib.lsqamd_qr_info(self.h, C.byref(n), C.byref(d))
        return n.value, d.value

    def timing(self, on=True):
        self.lib.lsqamd_timing_enable(self.h, int(on))

    def timings(self):
        out = {}
        for i, name in enumerate(_lib.T_NAMES):
            ms, cnt = C.c_double(), C.c_int64()
            self.lib.lsqamd_timing_get(self.h, i, C.byref(ms), C.byref(cnt))
            out[name] = (ms.value, cnt.value)
        return out

    def timing_reset(self):
        self.lib.lsqamd_timing_reset(self.h)

    # -- kernel-level calls -------------------------------------------------------------
    def _raise_reduce(self):
        e, self._reduce_error = getattr(self, '_reduce_error', None), None
        if e is not None:
            raise e

    def chi2(self, p):
        p = np.ascontiguousarray(p, np.float64)
        out = C.c_double()
        rc = self.lib.lsqamd_eval_residual(self.h, _lib.dptr(p), C.byref(out))
        self._raise_reduce()
        _check(self.lib, self.h, rc, 'eval_residual')
        return out.value

    def normal(self, p):
        p = np.ascontiguousarray(p, np.float64)
        out = C.c_double()
        rc = self.lib.lsqamd_eval_normal(self.h, _lib.dptr(p), C.byref(out))
        self._raise_reduce()
        _check(self.lib, self.h, rc, 'eval_normal')
        return out.value

    def solve_damped(self, mu, diag):
        diag = np.ascontiguousarray(diag, np.float64)
        v = np.empty(self.P)
        _check(self.lib, self.h, self.lib.lsqamd_solve_damped(self.h, float(mu), _lib.dptr(diag), _lib.dptr(v)),
               'solve_damped')
        return v

    def _get(self, fn, n, shape=None):
        if int(n) >= (1 << 17):     # >= 1 MiB: page-locked destination (cov is 134 MB, J 2.3 GB at C4)
            import torch
            out = torch.empty(int(n), dtype=torch.float64, pin_memory=True).numpy()
        else:
            out = np.empty(int(n))
        _check(self.lib, self.h, fn(self.h, _lib.dptr(out), out.size), fn.__name__)
        return out if shape is None else out.reshape(shape)

    def get_x(self):
        return self._get(self.lib.lsqamd_get_x, self.P)

    def get_grad(self):
        return self._get(self.lib.lsqamd_get_grad, self.P)

    def get_jtj(self):
        return self._get(self.lib.lsqamd_get_jtj, self.P * self.P, (self.P, self.P))

    def get_cov(self):
        return self._get(self.lib.lsqamd_get_cov, self.P * self.P, (self.P, self.P))

    def nf_data(self):
        nf = self.lib.lsqamd_nf(self.h)
        return nf - (self.P if self.wh.has_prior else 0)

    def get_f_data(self):
        return self._get(self.lib.lsqamd_get_f, self.nf_data())

    def get_J_data(self):
        n = self.nf_data()
        return self._get(self.lib.lsqamd_get_J, n * self.P, (n, self.P))

    def chi2_points(self, ps, max_scratch_bytes=1 << 30):
        """chi**2 at every row of ``ps`` (m x P) in one device pass (chunked to the scratch)."""
        import torch
        ps = np.ascontiguousarray(np.atleast_2d(np.asarray(ps, np.float64)))
        if ps.shape[1] != self.P:
            raise ValueError('points must have %d columns' % self.P)
        m = ps.shape[0]
        out = np.empty(m)
        if m == 0:
            return out
        nbytes = min(self.lib.lsqamd_chi2_points_work_bytes(self.h, m),
                     max(int(max_scratch_bytes), self.lib.lsqamd_chi2_points_work_bytes(self.h, 1)))
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        self._aux_tensors = [scratch]         # the per-point sums are all-reduced in place in there
        try:
            rc = self.lib.lsqamd_chi2_points(self.h, _lib.dptr(ps), m, C.c_void_p(scratch.data_ptr()), nbytes,
                                             _lib.dptr(out))
        finally:
            self._aux_tensors = []
        self._raise_reduce()
        _check(self.lib, self.h, rc, 'chi2_points')
        return out

    def dpdy(self, G=None):
        """Sensitivity of the best-fit parameters to the inputs at the current point:
        ``D = cov [J_f ; I]^T inv(C_reg)`` (``_getp``, src/lsqfit/__init__.py:897-911).

        ``G`` None -> ``D`` itself, shape (P, N + P) (columns: this problem's data rows in the
        caller's order, then the prior entries; (P, N) without a prior).  ``G`` of shape
        (m, P) -> ``G @ D`` without forming ``D`` (gradients of m derived outputs)."""
        import torch
        P = self.P
        ncol = self.N + (P if self.wh.has_prior else 0)
        if G is None:
            m, gt = P, None
        else:
            G = np.atleast_2d(np.asarray(G, np.float64))
            if G.shape[1] != P:
                raise ValueError('G must have %d columns' % P)
            m = G.shape[0]
            gt = np.ascontiguousarray(G.T)
        nbytes = self.lib.lsqamd_dpdy_work_bytes(self.h, m)
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        # pinned destination (the full D at the north-star shape is 2.3 GB) and a transposed VIEW of
        # it as the result: no pageable staging, no host-side transpose
        out_t = torch.empty((ncol, m), dtype=torch.float64, pin_memory=True)
        out = out_t.numpy()
        rc = self.lib.lsqamd_dpdy(self.h, None if gt is None else _lib.dptr(gt), m,
                                  C.c_void_p(scratch.data_ptr()), nbytes, _lib.dptr(out), out.size)
        _check(self.lib, self.h, rc, 'dpdy')
        del scratch
        return out.T

    def close(self):
        if getattr(self, 'h', None) is not None and self.h:
            self.lib.lsqamd_destroy(self.h)               # (waits for the stream)
            self.h = None
            idle = _STREAMS.setdefault(self.device.index, [])
            if len(idle) < 8 and getattr(self, 'stream', None) is not None:
                idle.append(self.stream)
                self.stream = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class mi355x_lm(object):
    r"""MI355X fitter for nonlinear least-squares multidimensional fits.

    Args mirror :class:`lsqfit.gsl_multifit` (src/lsqfit/_gsl.pyx:563-575):
    ``x0, n, f, tol, maxit, alg, solver, scaler, factor_up, factor_down, avmax``
    plus ``problem`` (a :class:`DeviceProblem`).  ``solver``: ``'cholesky'`` (default here: damped
    normal equations throughout) or ``'qr'`` (the reference's default; ``'svd'`` is taken as
    ``'qr'``): same LM steps, the covariance and ``logdet_jtj`` from a CholeskyQR factorisation of
    the whitened Jacobian -- accurate to cond(J) eps instead of cond(J)^2 eps (qr.hip).  Attributes: ``x cov f J nit
    tol stopping_criterion error description results`` (+ ``chi2``,
    ``logdet_jtj``, ``summary``).  ``f`` and ``J`` are fetched from the device on
    first access (J is n x P float64 -- 2.3 GB at the north-star shape).
    """

    def __init__(self, x0, n, f=None, tol=(1e-5, 0.0, 0.0), maxit=1000, alg='lm', solver='cholesky',
                 scaler='more', factor_up=3.0, factor_down=2.0, avmax=0.75, problem=None):
        if problem is None:
            problem = _problem_from_residual(x0, n, f)
        if alg not in _ALGS:
            raise ValueError('unkown algorithm ' + str(alg))            # _gsl.pyx:634-635
        if solver not in _SOLVERS:
            raise ValueError('unkown solver ' + str(solver))            # _gsl.pyx:652-653
        self.tol = normalize_tol(tol)
        self.maxit = maxit
        self.alg, self.solver, self.scaler = alg, solver, scaler
        self.factor_up, self.factor_down, self.avmax = factor_up, factor_down, avmax
        self.x0 = np.ascontiguousarray(x0, np.float64)
        self.n = n
        self.error = None
        self.description = describe(alg, scaler, solver, avmax)
        pr = self.problem = problem
        if self.x0.size != pr.P:
            raise ValueError('len(x0) = %d but the model has %d parameters' % (self.x0.size, pr.P))
        nf_total = pr.wh.nchiv
        if n is not None and int(n) != nf_total:
            raise ValueError('n = %d but the whitened residual has %d entries' % (n, nf_total))
        pr.set_options(self.tol, maxit, scaler, factor_up, factor_down, alg=alg, avmax=avmax, solver=solver)
        lib = pr.lib
        s = _lib.Summary()
        rc = lib.lsqamd_run(pr.h, _lib.dptr(self.x0), C.byref(s))
        pr._raise_reduce()
        _check(lib, pr.h, rc, 'run')
        self.summary = s
        self.nit = s.nit
        self.chi2 = s.chi2
        self.logdet_jtj = s.logdet_jtj
        self.stopping_criterion = s.stopping_criterion
        # _gsl.pyx:686-687,:714-717
        if s.status:
            self.error = (s.status, {11: 'exceeded max number of iterations'}.get(s.status, 'error'))
        if s.status == 11 and self.nit < maxit:
            self.error = "gsl_multifit can't improve on starting value; may have converged already."
        if s.info == 0 and self.error is None and maxit > 0:
            self.error = "gsl_multifit didn't converge in {} iterations".format(maxit)
        _cov_failed(self, s)
        self.x = pr.get_x()
        self.cov = pr.get_cov()
        self.results = None
        self._f = self._J = None

    def _prior_rows(self):
        return self.problem.wh.prior_rows(self.x)

    @property
    def f(self):
        if self._f is None:
            wh = self.problem.wh
            fd = self.problem.get_f_data()
            if not wh.has_prior:
                self._f = fd
            else:
                nd1 = wh.n_data - sum(b['size'] for b in wh.blocks)
                rows = wh.prior_rows(self.x)
                if wh.prior_W[0] == 'diag':
                    self._f = np.concatenate([fd[:nd1], rows[0], fd[nd1:]])
                else:
                    self._f = np.concatenate([fd[:nd1], rows[0], fd[nd1:]] + list(rows[2]))
        return self._f

    @property
    def J(self):
        if self._J is None:
            wh = self.problem.wh
            Jd = self.problem.get_J_data()
            if not wh.has_prior:
                self._J = Jd
            else:
                nd1 = wh.n_data - sum(b['size'] for b in wh.blocks)
                rows = wh.prior_rows(self.x)
                if wh.prior_W[0] == 'diag':
                    self._J = np.vstack([Jd[:nd1], rows[1], Jd[nd1:]])
                else:
                    self._J = np.vstack([Jd[:nd1], rows[1], Jd[nd1:]] + list(rows[3]))
        return self._J


class mi355x_trf(mi355x_lm):
    r"""MI355X counterpart of :class:`lsqfit.scipy_least_squares` (src/lsqfit/_scipy.py:20-181)
    with its three methods: the Trust Region Reflective algorithm (default) and the dogleg method
    in a rectangular trust region, which honour box bounds, and MINPACK's Levenberg-Marquardt.

    ``x0, n, f, tol, maxit`` as there (``tol`` default ``(1e-8, 1e-8, 1e-8)``, ``maxit`` = cap on
    function evaluations); ``method`` None / ``'trf'``, ``'dogbox'`` or ``'lm'`` (no bounds, all
    tolerances above machine epsilon); ``bounds=(lower, upper)``;
    ``x_scale`` a positive number, an array of P of them, or ``'jac'``; ``loss`` one of scipy's ``'linear'``,
    ``'soft_l1'``, ``'huber'``, ``'cauchy'``, ``'arctan'`` with ``f_scale`` (methods trf and dogbox; a callable loss is not
    something a device can run); ``tr_solver`` None or ``'exact'`` (``'lsmr'`` -- scipy's iterative solver for sparse
    Jacobians -- is declined), ``tr_options`` must be empty, ``jac_sparsity`` None: the pass-through options
    :76-79 names and :147-153 forwards.  ``nit`` counts function evaluations (:161),
    ``stopping_criterion`` follows :176-181, ``cov`` is ``inv(J^T J)`` of the (loss-scaled) Jacobian scipy returns (what
    :165-169 gives for a full-rank Jacobian; a rank-deficient one gives the truncated inverse).
    """
    LOSSES = dict(linear=0, soft_l1=1, huber=2, cauchy=3, arctan=4)

    def __init__(self, x0, n, f=None, tol=(1e-8, 1e-8, 1e-8), maxit=1000, method=None, bounds=None,
                 x_scale=1.0, loss='linear', f_scale=1.0, tr_solver=None, tr_options=None, jac_sparsity=None, problem=None):
        if problem is None:
            problem = _problem_from_residual(x0, n, f)
        if method is None:
            method = 'trf'                                                 # _scipy.py:135-139
        if method not in ('trf', 'dogbox', 'lm'):
            raise ValueError("`method` must be 'trf', 'dogbox' or 'lm'.")
        xs = None
        if isinstance(x_scale, str):
            if x_scale != 'jac':
                raise ValueError("`x_scale` must be 'jac' or array_like with positive numbers.")     # scipy's wording
            scaler = 'more'
        else:
            scaler = 'levenberg'
            xs = np.asarray(x_scale, float)
            if xs.ndim > 1 or not np.all(np.isfinite(xs)) or np.any(xs <= 0):
                raise ValueError("`x_scale` must be 'jac' or array_like with positive numbers.")
            if xs.ndim == 1 and xs.size != np.size(x0):
                raise ValueError('Inconsistent shapes between `x_scale` and `x0`.')
            xs = None if np.all(xs == 1.0) else np.ascontiguousarray(np.broadcast_to(xs, (np.size(x0),)), np.float64)
        if callable(loss):
            raise NotImplementedError('a callable loss cannot run on the device; use one of %s' % sorted(self.LOSSES))
        if loss not in self.LOSSES:
            raise ValueError('`loss` must be one of %s or a callable.' % list(self.LOSSES))           # scipy's wording
        if method == 'lm' and loss != 'linear':
            raise ValueError("method='lm' supports only 'linear' loss function.")
        f_scale = float(f_scale)
        if loss != 'linear' and not f_scale > 0:
            raise ValueError('`f_scale` must be positive')
        if tr_solver not in (None, 'exact'):
            if tr_solver == 'lsmr':
                raise NotImplementedError("tr_solver='lsmr' (iterative, for sparse Jacobians) is not built; the device solves the "
                                          "trust-region sub-problems exactly")
            raise ValueError("`tr_solver` must be None, 'exact' or 'lsmr'.")
        if tr_options:
            raise NotImplementedError("tr_options are lsmr's; the exact solver takes none")
        if jac_sparsity is not None:
            raise NotImplementedError('jac_sparsity: the device Jacobian is dense')
        if maxit is not None and maxit <= 0:
            raise ValueError('`max_nfev` must be None or positive integer.')
        self.tol = normalize_tol(tol)
        self.maxit = maxit
        self.method, self.x_scale, self.loss, self.f_scale = method, x_scale, loss, f_scale
        self.x0 = np.ascontiguousarray(x0, np.float64)
        self.n = n
        self.error = None
        self.description = 'method = {}'.format(method)                    # _scipy.py:134-139
        pr = self.problem = problem
        if self.x0.size != pr.P:
            raise ValueError('len(x0) = %d but the model has %d parameters' % (self.x0.size, pr.P))
        if n is not None and int(n) != pr.wh.nchiv:
            raise ValueError('n = %d but the whitened residual has %d entries' % (n, pr.wh.nchiv))
        pr.set_options(self.tol, 100 * pr.P if maxit is None else maxit, scaler,
                       alg='minpack_lm' if method == 'lm' else method)
        pr.set_bounds(bounds)
        lib = pr.lib
        s = _lib.Summary()
        try:
            _check(lib, pr.h, lib.lsqamd_set_loss(pr.h, self.LOSSES[loss], f_scale), 'set_loss')
            _check(lib, pr.h, lib.lsqamd_set_x_scale(pr.h, None if xs is None else _lib.dptr(xs)), 'set_x_scale')
            rc = lib.lsqamd_run(pr.h, _lib.dptr(self.x0), C.byref(s))
            pr._raise_reduce()
            if rc == -1:                     # infeasible x0 / tolerances: scipy raises ValueError
                raise ValueError(lib.lsqamd_last_error(pr.h).decode())
            _check(lib, pr.h, rc, 'run')
        finally:
            pr.set_bounds(None)
            lib.lsqamd_set_loss(pr.h, 0, 1.0)
            lib.lsqamd_set_x_scale(pr.h, None)
        self.summary = s
        self.nit = s.nit
        self.chi2 = s.chi2
        self.logdet_jtj = s.logdet_jtj
        self.stopping_criterion = s.stopping_criterion
        self.status = s.info - 100           # scipy's OptimizeResult.status
        _cov_failed(self, s)
        self.x = pr.get_x()
        self.cov = pr.get_cov()
        self.results = None
        self._f = self._J = None


def register(lsqfit_module):
    """``lsqfit.nonlinear_fit.FITTERS['mi355x_lm'] = mi355x_lm`` (and ``'mi355x_trf'``) when
    lsqfit is importable (src/lsqfit/__init__.py:110-126,:453)."""
    lsqfit_module.nonlinear_fit.FITTERS['mi355x_lm'] = mi355x_lm
    lsqfit_module.nonlinear_fit.FITTERS['mi355x_trf'] = mi355x_trf
    return 'mi355x_lm'
