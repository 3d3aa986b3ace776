"""Host port of the cosmix workload's normal equations (ORACLE side: test infrastructure and the bench's CPU baseline).

What ``chiv`` + ``_c_df`` + the 'cholesky' solver's ``J^T J`` amount to for the bench model (src/lsqfit/_utilities.pyx:65-94,
src/lsqfit/_gsl.pyx:742-760,:646-653) on every host core, with blocked kernels -- SURVEY.md 8d's "strong" CPU mode:

  trig       cos / sin of the N x K phase matrix in row chunks on a thread pool (numpy releases the GIL inside its loops),
             written straight into the preallocated Jacobian (no hstack, no 2 GB temporaries);
  whiten     J_b = W_b J_b per covariance block (blocks up to 1024 rows inside the same pool task, one BLAS thread each;
             larger ones as one multi-threaded GEMM), W_b = inv(chol(C_b));
  syrk       J^T J with BLAS dsyrk (the triangle only: half the flops of J.T @ J), mirrored once.

Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may import this (nothing under lsqfit_amd/ does).
"""
import os
import time

import numpy as np


class _Factor:
    """a large block's Cholesky factor standing in for W = inv(L): ``W @ v`` is a triangular solve"""

    def __init__(self, L):
        self.L = L
        self.shape = L.shape

    def __matmul__(self, v):
        import scipy.linalg as sla
        return sla.solve_triangular(self.L, v, lower=True)


class CosmixPort:
    def __init__(self, d):
        import scipy.linalg as sla
        self.x, self.ymean = np.asarray(d['x'], float), np.asarray(d['ymean'], float)
        self.pm, perr = d['prior']
        self.P = self.pm.size
        self.K = self.P // 2
        self.N = self.ymean.size
        yerr = d['yerr']
        sd = np.asarray(yerr['sdev'] if isinstance(yerr, dict) else yerr, float)
        blocks = yerr['blocks'] if isinstance(yerr, dict) else []
        # whitening set-up (untimed, as on the GPU side): W_b = inv(chol(C_b))
        self.Ws, self._ld = [], []
        for r0, cov in blocks:
            L = sla.cholesky(cov, lower=True)
            self._ld.append(2.0 * float(np.sum(np.log(np.diag(L)))))
            # W_b = inv(L) explicitly for the blocks whitened inside a pool task (small GEMMs); a LARGE block keeps its factor
            # and is whitened by a triangular solve (an 8192^3 explicit inverse would cost more than the fit's iterations)
            self.Ws.append((int(r0), sla.solve_triangular(L, np.eye(cov.shape[0]), lower=True) if cov.shape[0] <= 1024 else _Factor(L)))
        inblk = np.zeros(self.N, bool)
        for r0, W in self.Ws:
            inblk[r0:r0 + W.shape[0]] = True
        self.wdiag = np.where(inblk, 1.0, 1.0 / sd)
        if np.ndim(perr) == 2:
            Lp = sla.cholesky(np.asarray(perr, float), lower=True)
            self.prec = sla.cho_solve((Lp, True), np.eye(self.P))
            logdet_prior = 2.0 * float(np.sum(np.log(np.diag(Lp))))
        else:
            self.prec = np.diag(1.0 / np.asarray(perr, float) ** 2)
            logdet_prior = 2.0 * float(np.sum(np.log(np.asarray(perr, float))))
        # log det of the covariance of concat(y, prior): blocks from their Cholesky factors, the rest from the sdevs
        self.logdet = logdet_prior + 2.0 * float(np.sum(np.log(sd[~inblk]))) + sum(self._ld)
        self.cores = os.cpu_count() or 1
        self.workers = max(1, min(self.cores, 64))
        # row chunks: whole small blocks, else 512 rows; big blocks are whitened afterwards
        edges, small, r = [], {}, 0
        big = []
        for r0, W in sorted(self.Ws, key=lambda t: t[0]):
            B = W.shape[0]
            while r < r0:
                edges.append((r, min(r + 512, r0), None)); r = edges[-1][1]
            if B <= 1024:
                edges.append((r0, r0 + B, W))
            else:
                big.append((r0, W))
                a = r0
                while a < r0 + B:
                    edges.append((a, min(a + 512, r0 + B), None)); a = edges[-1][1]
            r = r0 + B
        while r < self.N:
            edges.append((r, min(r + 512, self.N), None)); r = edges[-1][1]
        self.chunks, self.big = edges, big
        self.J = np.empty((self.N, self.P))
        self.rw = np.empty(self.N)
        self.phases = dict(trig=0.0, whiten=0.0, syrk=0.0, cholesky=0.0)
        from concurrent.futures import ThreadPoolExecutor
        self.pool = ThreadPoolExecutor(self.workers)

    def _limits(self, n):
        try:
            import threadpoolctl
            return threadpoolctl.threadpool_limits(limits=n)
        except Exception:
            import contextlib
            return contextlib.nullcontext()

    def _rows(self, a, b, W, p, jac, acc):
        K, x = self.K, self.x[a:b]
        t0 = time.perf_counter()
        wx = np.multiply.outer(x, p[K:])
        c = np.cos(wx)
        raw = c @ p[:K] - self.ymean[a:b]
        if jac:
            np.sin(wx, out=wx)
            wx *= -p[:K]
            wx *= x[:, None]
        t1 = time.perf_counter()
        if W is not None:
            self.rw[a:b] = W @ raw
            if jac:
                self.J[a:b, :K] = W @ c
                self.J[a:b, K:] = W @ wx
        else:
            wd = self.wdiag[a:b]
            self.rw[a:b] = wd * raw
            if jac:
                np.multiply(c, wd[:, None], out=self.J[a:b, :K])
                np.multiply(wx, wd[:, None], out=self.J[a:b, K:])
        t2 = time.perf_counter()
        acc.append((t1 - t0, t2 - t1))

    def _assemble(self, p, jac):
        """whitened residual (and Jacobian) at p -> self.rw (self.J); thread-time of the two sub-phases apportions the wall"""
        acc = []
        t0 = time.perf_counter()
        with self._limits(1):
            list(self.pool.map(lambda ch: self._rows(ch[0], ch[1], ch[2], p, jac, acc), self.chunks))
        wall = time.perf_counter() - t0
        tt, tw = sum(a for a, _ in acc), sum(b for _, b in acc)
        self.phases['trig'] += wall * tt / max(tt + tw, 1e-30)
        self.phases['whiten'] += wall * tw / max(tt + tw, 1e-30)
        t0 = time.perf_counter()
        for r0, W in self.big:                    # one large dense block (c3): a multi-threaded GEMM
            B = W.shape[0]
            self.rw[r0:r0 + B] = W @ self.rw[r0:r0 + B]
            if jac:
                self.J[r0:r0 + B] = W @ self.J[r0:r0 + B]
        self.phases['whiten'] += time.perf_counter() - t0

    def chi2_fn(self, p):
        self._assemble(p, False)
        dp = p - self.pm
        return float(self.rw @ self.rw + dp @ self.prec @ dp)

    def normal_eq(self, p):
        import scipy.linalg.blas as blas
        self._assemble(p, True)
        t0 = time.perf_counter()
        U = blas.dsyrk(1.0, self.J.T, trans=0, lower=0)      # J.T is Fortran-contiguous: no copy; upper triangle of J^T J
        A = np.ascontiguousarray(U)
        A += np.triu(U, 1).T
        g = self.J.T @ self.rw
        self.phases['syrk'] += time.perf_counter() - t0
        dp = p - self.pm
        return A + self.prec, g + self.prec @ dp, float(self.rw @ self.rw + dp @ self.prec @ dp)

    def close(self):
        self.pool.shutdown()
