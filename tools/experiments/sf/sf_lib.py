"""Loader of the VARIANT library that carries the streamed-factorisation experiment (tools/experiments/sf/build.sh ->
lsqfit_amd/build/libsf.so): the product library without it is lsqfit_amd/liblsqfit_amd.so."""
import ctypes as C
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..', '..'))
SF_LIB = os.path.join(ROOT, 'lsqfit_amd', 'build', 'libsf.so')


def load():
    if not os.path.exists(SF_LIB):
        raise SystemExit('run tools/experiments/sf/build.sh first (%s is missing)' % SF_LIB)
    os.environ['LSQAMD_LIBPATH'] = SF_LIB
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from lsqfit_amd import _lib
    assert _lib.LIBPATH == SF_LIB, 'lsqfit_amd._lib was imported before sf_lib.load()'
    lib = _lib.load()
    vp = C.c_void_p
    lib.lsqamd_op_sf_work_bytes.restype = C.c_size_t
    lib.lsqamd_op_sf_work_bytes.argtypes = [C.c_int64, C.c_int64, C.c_int32]
    lib.lsqamd_op_sf_factor.restype = C.c_int
    lib.lsqamd_op_sf_factor.argtypes = [vp, vp, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp,
                                        C.c_int32, vp, C.c_double, C.c_int32, vp, vp, vp, vp, C.c_size_t, vp, vp, C.c_int32]
    lib.lsqamd_debug_where.restype = C.c_int
    lib.lsqamd_debug_where.argtypes = [vp, C.c_int32, vp, C.c_int32]
    return lib
