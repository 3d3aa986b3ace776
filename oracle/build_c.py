"""Build recipe for the oracle's C restatement (test / baseline infrastructure): gcc -O2 on oracle/csrc/*.c ->
oracle/_build/liboracle_c.so (git-ignored; travels to the GPU box like the product's built .so).

    python -m oracle.build_c
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = [os.path.join(HERE, 'csrc', 'qrpt_unblocked.c')]
OUT = os.path.join(HERE, '_build', 'liboracle_c.so')


def build(force=False):
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    if not force and os.path.exists(OUT) and all(os.path.getmtime(s) <= os.path.getmtime(OUT) for s in SRC):
        return OUT
    cmd = [os.environ.get('CC', 'gcc'), '-O2', '-fPIC', '-shared', '-o', OUT] + SRC + ['-lm']
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError('gcc failed:\n%s\n%s' % (' '.join(cmd), r.stdout))
    return OUT


def load():
    """ctypes handle with oracle_qrpt_unblocked bound (builds on first use when gcc is there)."""
    import ctypes as C
    lib = C.CDLL(build())
    lib.oracle_qrpt_unblocked.restype = C.c_int
    lib.oracle_qrpt_unblocked.argtypes = [C.POINTER(C.c_double), C.c_long, C.c_long, C.POINTER(C.c_double),
                                          C.POINTER(C.c_long), C.POINTER(C.c_double)]
    return lib


def qrpt(A):
    """-> (R[n x n upper], perm) of the unblocked pivoted Householder QR of A (m x n, m >= n)."""
    import ctypes as C
    import numpy as np
    lib = load()
    A = np.array(A, dtype=np.float64, order='C')
    m, n = A.shape
    tau, work = np.empty(n), np.empty(2 * n)
    perm = np.empty(n, dtype=np.int64)
    dp = C.POINTER(C.c_double)
    lib.oracle_qrpt_unblocked(A.ctypes.data_as(dp), m, n, tau.ctypes.data_as(dp), perm.ctypes.data_as(C.POINTER(C.c_long)),
                              work.ctypes.data_as(dp))
    return np.triu(A[:n]), perm


if __name__ == '__main__':
    print(build(force=True))
