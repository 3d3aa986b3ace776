"""ctypes binding of liblsqfit_amd.so (the C ABI in include/lsqfit_amd.h).

There is deliberately no fallback: if the HIP library is missing, ``load()``
raises, and every compute entry point of the package goes through ``load()``.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIBPATH = os.environ.get('LSQAMD_LIBPATH') or os.path.join(HERE, 'liblsqfit_amd.so')   # (the variable: developer builds, tools/build_variant.sh)
ABI_VERSION = 8

COMM_ID_BYTES = 128

T_NAMES = ['residual', 'jacobian', 'whiten', 'syrk', 'grad', 'reduce', 'cholesky', 'solve', 'covar', 'exch_coll', 'exch_wait']

ERRORS = {-1: 'EINVAL', -2: 'EHIP', -3: 'ENOMEM', -4: 'ENOTPD', -5: 'ENONFINITE',
          -6: 'EUNSUPPORTED', -7: 'EREDUCE', -8: 'ECAPACITY', -9: 'EINACCURATE', -10: 'EINTERNAL'}


class Config(C.Structure):
    _fields_ = [('abi_version', C.c_int32), ('model', C.c_int32), ('n_data', C.c_int64),
                ('n_param', C.c_int64), ('n_x', C.c_int32), ('has_prior', C.c_int32),
                ('prior_dense', C.c_int32), ('n_blocks', C.c_int32), ('max_block', C.c_int64),
                ('sum_block_sq', C.c_int64), ('want_jacobian_out', C.c_int32), ('n_batch', C.c_int32),
                ('tape_len', C.c_int32), ('reserved1', C.c_int32)]


class Options(C.Structure):
    _fields_ = [('xtol', C.c_double), ('gtol', C.c_double), ('ftol', C.c_double), ('maxit', C.c_int32),
                ('scaler', C.c_int32), ('solver', C.c_int32), ('trs', C.c_int32),
                ('factor_up', C.c_double), ('factor_down', C.c_double), ('avmax', C.c_double)]


class Summary(C.Structure):
    _fields_ = [('status', C.c_int32), ('info', C.c_int32), ('stopping_criterion', C.c_int32),
                ('nit', C.c_int32), ('nfev', C.c_int32), ('njev', C.c_int32), ('ntrial', C.c_int32),
                ('chol_fail', C.c_int32), ('cov_status', C.c_int32), ('qr_trials', C.c_int32),
                ('chi2', C.c_double), ('mu', C.c_double),
                ('logdet_jtj', C.c_double), ('t_setup_ms', C.c_double), ('t_run_ms', C.c_double)]


REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64)

_dp = C.POINTER(C.c_double)
_vp = C.c_void_p

# name -> (restype, argtypes): every symbol include/lsqfit_amd.h declares
PROTOTYPES = {
    'lsqamd_abi_version': (C.c_int, []),
    'lsqamd_workspace_bytes': (C.c_size_t, [C.POINTER(Config)]),
    'lsqamd_create': (C.c_int, [C.POINTER(Config), _vp, C.c_size_t, _vp, C.POINTER(_vp)]),
    'lsqamd_destroy': (C.c_int, [_vp]),
    'lsqamd_last_error': (C.c_char_p, [_vp]),
    'lsqamd_whiten_work_bytes': (C.c_size_t, [C.c_int64, C.c_int32]),
    'lsqamd_whiten_blocks': (C.c_int, [_vp, C.c_int64, C.c_int32, _vp, C.c_double, _vp, _vp, _vp, C.c_size_t,
                                       _dp, _dp, _dp, C.POINTER(C.c_int32)]),
    'lsqamd_set_x': (C.c_int, [_vp, _dp, C.c_int64, C.c_int32]),
    'lsqamd_set_tape': (C.c_int, [_vp, C.POINTER(C.c_int32), C.c_int32, _dp, C.c_int32]),
    'lsqamd_set_data': (C.c_int, [_vp, _dp, _dp, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                  C.POINTER(C.c_int64), C.POINTER(C.c_int32), _vp]),
    'lsqamd_set_prior': (C.c_int, [_vp, _dp, _vp]),
    'lsqamd_set_ymean': (C.c_int, [_vp, _dp]),
    'lsqamd_set_options': (C.c_int, [_vp, C.POINTER(Options)]),
    'lsqamd_qr_work_bytes': (C.c_size_t, [_vp]),
    'lsqamd_set_qr_work': (C.c_int, [_vp, _vp, C.c_size_t]),
    'lsqamd_qr_info': (C.c_int, [_vp, C.POINTER(C.c_int32), _dp]),
    'lsqamd_query_devices': (C.c_int, [C.POINTER(C.c_int32), C.c_int32, C.c_char_p, C.c_size_t,
                                       C.POINTER(C.c_int64)]),
    'lsqamd_set_bounds': (C.c_int, [_vp, _dp, _dp]),
    'lsqamd_set_linear': (C.c_int, [_vp, C.POINTER(C.c_int32), C.c_int32]),
    'lsqamd_set_param_rows': (C.c_int, [_vp, C.POINTER(C.c_int32)]),
    'lsqamd_set_loss': (C.c_int, [_vp, C.c_int32, C.c_double]),
    'lsqamd_set_x_scale': (C.c_int, [_vp, _vp]),
    'lsqamd_set_reduce': (C.c_int, [_vp, REDUCE_FN, _vp]),
    'lsqamd_set_adds_prior': (C.c_int, [_vp, C.c_int32]),
    'lsqamd_comm_unique_id': (C.c_int, [_vp, C.c_size_t]),
    'lsqamd_comm_init': (C.c_int, [_vp, _vp, C.c_size_t, C.c_int32, C.c_int32]),
    'lsqamd_comm_destroy': (C.c_int, [_vp]),
    'lsqamd_comm_info': (C.c_int, [_vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    'lsqamd_comm_stats': (C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
    'lsqamd_comm_shutdown': (C.c_int, []),
    'lsqamd_set_tape_programs': (C.c_int, [_vp, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int32), _dp,
                                           C.c_int32]),
    'lsqamd_tape_codegen': (C.c_int, [C.POINTER(C.c_int32), C.c_int32, _dp, C.c_int32, C.c_int32, C.c_int32, C.c_char_p,
                                      C.c_size_t, C.POINTER(C.c_int32), C.c_int32]),
    'lsqamd_run': (C.c_int, [_vp, _dp, C.POINTER(Summary)]),
    'lsqamd_init': (C.c_int, [_vp, _dp]),
    'lsqamd_step': (C.c_int, [_vp, C.POINTER(C.c_int32)]),
    'lsqamd_finish': (C.c_int, [_vp, C.POINTER(Summary)]),
    'lsqamd_eval_residual': (C.c_int, [_vp, _dp, _dp]),
    'lsqamd_eval_normal': (C.c_int, [_vp, _dp, _dp]),
    'lsqamd_eval_fcn': (C.c_int, [_vp, _dp, _dp, C.c_size_t]),
    'lsqamd_solve_damped': (C.c_int, [_vp, C.c_double, _dp, _dp]),
    'lsqamd_op_gemm_tn': (C.c_int, [_vp, C.c_int64, C.c_int64, C.c_int64, C.c_double, _vp, C.c_int64,
                                    _vp, C.c_int64, C.c_double, _vp, C.c_int64, C.c_int32, C.c_int32]),
    'lsqamd_op_potrf_upper': (C.c_int, [_vp, _vp, C.c_int64, C.c_int64, C.c_int64, _vp, C.c_size_t, _vp]),
    'lsqamd_op_potrf_work_bytes': (C.c_size_t, [C.c_int64]),
    'lsqamd_op_truncated_inverse': (C.c_int, [_dp, C.c_int64, C.c_int64, C.c_int32, _dp, C.POINTER(C.c_int32)]),
    'lsqamd_get_x': (C.c_int, [_vp, _dp, C.c_size_t]),
    'lsqamd_get_f': (C.c_int, [_vp, _dp, C.c_size_t]),
    'lsqamd_get_J': (C.c_int, [_vp, _dp, C.c_size_t]),
    'lsqamd_get_jtj': (C.c_int, [_vp, _dp, C.c_size_t]),
    'lsqamd_get_grad': (C.c_int, [_vp, _dp, C.c_size_t]),
    'lsqamd_get_cov': (C.c_int, [_vp, _dp, C.c_size_t]),
    'lsqamd_nf': (C.c_int64, [_vp]),
    'lsqamd_chi2_points_work_bytes': (C.c_size_t, [_vp, C.c_int64]),
    'lsqamd_chi2_points': (C.c_int, [_vp, _dp, C.c_int64, _vp, C.c_size_t, _dp]),
    'lsqamd_dpdy_work_bytes': (C.c_size_t, [_vp, C.c_int64]),
    'lsqamd_dpdy': (C.c_int, [_vp, _dp, C.c_int64, _vp, C.c_size_t, _dp, C.c_size_t]),
    'lsqamdb_workspace_bytes': (C.c_size_t, [C.POINTER(Config), C.c_int32]),
    'lsqamdb_create': (C.c_int, [C.POINTER(Config), C.c_int32, _vp, C.c_size_t, _vp, C.POINTER(_vp)]),
    'lsqamdb_destroy': (C.c_int, [_vp]),
    'lsqamdb_last_error': (C.c_char_p, [_vp]),
    'lsqamdb_set_x': (C.c_int, [_vp, _dp, C.c_int64, C.c_int32]),
    'lsqamdb_set_tape': (C.c_int, [_vp, C.POINTER(C.c_int32), C.c_int32, _dp, C.c_int32]),
    'lsqamdb_set_data': (C.c_int, [_vp, _dp, _dp]),
    'lsqamdb_set_blocks': (C.c_int, [_vp, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                     C.POINTER(C.c_int64), C.POINTER(C.c_int32), _vp]),
    'lsqamdb_set_data_means': (C.c_int, [_vp, _dp]),
    'lsqamdb_set_priors': (C.c_int, [_vp, _dp, _vp]),
    'lsqamdb_set_options': (C.c_int, [_vp, C.POINTER(Options)]),
    'lsqamdb_run': (C.c_int, [_vp, _dp, C.POINTER(Summary), C.c_int32]),
    'lsqamdb_get_x': (C.c_int, [_vp, _dp, C.c_size_t]),
    'lsqamdb_covariance': (C.c_int, [_vp, _dp, C.c_size_t]),
    'lsqamdb_get_cov': (C.c_int, [_vp, C.c_int32, _dp, C.c_size_t]),
    'lsqamdb_get_cov_all': (C.c_int, [_vp, _dp, C.c_size_t]),
    'lsqamdb_rounds': (C.c_int32, [_vp]),
    'lsqamdb_timing_enable': (C.c_int, [_vp, C.c_int32]),
    'lsqamdb_timing_get': (C.c_int, [_vp, C.c_int32, _dp, C.POINTER(C.c_int64)]),
    'lsqamd_timing_enable': (C.c_int, [_vp, C.c_int32]),
    'lsqamd_timing_get': (C.c_int, [_vp, C.c_int32, _dp, C.POINTER(C.c_int64)]),
    'lsqamd_timing_reset': (C.c_int, [_vp]),
    'lsqamd_debug_flags': (C.c_int64, [_vp]),
    'lsqamd_handoff_stats': (C.c_int, [C.POINTER(C.c_int64)]),
    'lsqamd_debug_throw': (C.c_int, [_vp, C.c_int32]),
    'lsqamd_debug_capture_selftest': (C.c_int, [_vp, C.POINTER(C.c_int32)]),
    'lsqamd_debug_syrk_work': (C.c_int64, [C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    'lsqamd_debug_per_device_once': (C.c_int, [C.c_int32, C.c_int32]),
    'lsqamd_jit_cache_stats': (C.c_int, [C.POINTER(C.c_int64)]),
    'lsqamd_debug_set_potf2_stamps': (None, [_vp]),
}

_lib = None


def load():
    """Return the bound library; raise if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIBPATH):
        raise ImportError(
            'lsqfit_amd: %s is missing -- the HIP backend has not been built '
            '(run `python -m lsqfit_amd.build`); there is no CPU fallback' % LIBPATH)
    import torch  # noqa: F401  (loads the HIP runtime the process shares with torch)
    lib = C.CDLL(LIBPATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)           # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.lsqamd_abi_version() != ABI_VERSION:
        raise ImportError('lsqfit_amd: ABI version mismatch')
    _lib = lib
    return lib


_side = None


def side_stream(device=None):
    """A non-blocking torch stream per host thread for the package's OWN device work outside a handle (block factorisations
    of the whitening set-up, concatenating device-made weights, fetching them to the host).  Never the legacy default
    stream: a legacy-stream operation synchronises with every blocking stream of the process and, on this runtime, fails
    with hipErrorStreamCaptureImplicit -- and invalidates the capture -- while ANY other thread is capturing a graph
    (handles capture their LM step; tests/test_gpu_threads.py).  Work queued on it is waited for (``.synchronize()``) before
    its results are handed to another stream."""
    global _side
    import threading
    import torch
    if _side is None:
        _side = threading.local()
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    key = 's%d' % dev
    s = getattr(_side, key, None)
    if s is None:
        s = torch.cuda.Stream(device=dev)
        setattr(_side, key, s)
    return s


def dptr(a):
    return a.ctypes.data_as(_dp)


def anyptr(a):
    """void* of a numpy array (host) or a torch tensor (device): the arguments of the C ABI that
    accept either kind of memory."""
    if hasattr(a, 'data_ptr'):
        return C.c_void_p(a.data_ptr())
    return C.c_void_p(a.ctypes.data)
