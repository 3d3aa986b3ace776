"""Whitening of the joint [y ; prior] distribution (ORACLE ONLY).

Restates what ``nonlinear_fit`` asks of ``gvar.PDF`` at
src/lsqfit/__init__.py:1892-1900 and then reads back at :553-561,:574,:723
and src/lsqfit/_utilities.pyx:58-61:

  * the covariance of concat(y, prior) is split into its block-diagonal
    structure; all 1x1 blocks are gathered into one "diagonal" entry whose
    weights are 1/sdev (pinned by tests/test_lsqfit.py:961-962,:1031-1032);
  * every larger block is regulated on its *correlation* matrix: eigenvalues
    below ``svdcut * max eigenvalue`` are raised to that floor
    (doc/source/overview.rst:1546-1556), or dropped when ``svdcut < 0``
    (tests/test_lsqfit.py:837-841); ``nmod`` counts the touched modes;
  * weights ``W`` with ``sum_rows outer(w, w) == inv(C_regulated)``
    (tests/test_lsqfit.py:923-931,:1013-1017);
  * ``logdet == log det C_regulated`` over kept modes (:932-943).

gvar itself is third-party and absent from /root/reference; its block search
and eigen-regulation are restated from its documentation.  The ``eps``
regulation mode (``gvar.regulate(g, eps=...)``: every correlated block's CORRELATION
matrix gets ``eps * ||corr||_inf`` added to its diagonal, then plain Cholesky-style
weights; all of a block's modes count as modified) is restated from gvar's published
documentation as well, but NO literal expected value for it exists anywhere in the
reference (no test, no example output): PARITY UNPINNED for ``eps`` -- the tests
only check the stated identity ``W^T W = inv(C + eps ||corr||_inf D^2)``.
"""
import numpy as np


def find_blocks(cov, tol=0.0):
    """Connected components of the off-diagonal sparsity pattern of ``cov``."""
    cov = np.asarray(cov, float)
    n = cov.shape[0]
    label = -np.ones(n, int)
    comps = []
    nz = np.abs(cov) > tol
    for s in range(n):
        if label[s] >= 0:
            continue
        stack = [s]
        label[s] = len(comps)
        members = []
        while stack:
            i = stack.pop()
            members.append(i)
            for j in np.nonzero(nz[i])[0]:
                if label[j] < 0:
                    label[j] = len(comps)
                    stack.append(j)
        comps.append(np.array(sorted(members), int))
    return comps


class PDF:
    """mean: float[n]; sdev: float[n]; blocks: list of (idx int[B], cov float[B,B])
    overriding ``sdev`` on their indices (B >= 2)."""

    def __init__(self, mean, sdev, blocks=(), svdcut=1e-12, eps=None):
        # src/lsqfit/__init__.py:240-245: eps is ignored when svdcut is given (and not None)
        if svdcut is not None:
            eps = None
        if eps is not None and eps < 0:
            raise ValueError('eps must not be negative')
        self.mean = np.array(mean, float)
        n = self.mean.size
        sdev = np.array(sdev, float)
        self.svdcut = svdcut
        self.eps = eps
        in_block = np.zeros(n, bool)
        for idx, _ in blocks:
            in_block[np.asarray(idx, int)] = True
        d_idx = np.nonzero(~in_block)[0]
        self.i_invwgts = [(d_idx, 1.0 / sdev[d_idx])]
        self.logdet = 2.0 * float(np.sum(np.log(sdev[d_idx])))
        self.nmod = 0
        self.nblocks = {}
        if d_idx.size:
            self.nblocks[1] = int(d_idx.size)
        self.correction_var = np.zeros(n)     # variance added to each entry
        self.cov_blocks = []                  # regulated covariance, per block
        self.sdev = sdev.copy()
        for idx, cov in sorted(blocks, key=lambda b: int(np.min(b[0]))):
            idx = np.asarray(idx, int)
            cov = np.asarray(cov, float)
            B = idx.size
            self.nblocks[B] = self.nblocks.get(B, 0) + 1
            sd = np.sqrt(np.diag(cov))
            corr = cov / np.outer(sd, sd)
            if eps:                                # Tikhonov shift of the whole spectrum (UNPINNED, see header)
                corr = corr + eps * np.linalg.norm(corr, np.inf) * np.eye(B)
                self.nmod += B
            lam, vec = np.linalg.eigh(corr)
            keep = np.ones(B, bool)
            lam_reg = lam.copy()
            if svdcut is not None and svdcut != 0:
                lmin = abs(svdcut) * lam[-1]
                low = lam < lmin
                self.nmod += int(np.sum(low))
                if svdcut > 0:
                    lam_reg[low] = lmin
                else:
                    keep = ~low
            lam_k = lam_reg[keep]
            vec_k = vec[:, keep]
            W = (vec_k / np.sqrt(lam_k)).T / sd[None, :]
            self.i_invwgts.append((idx, W))
            self.logdet += float(np.sum(np.log(lam_k)) + 2.0 * np.sum(np.log(sd)))
            corr_reg = (vec_k * lam_k) @ vec_k.T
            if np.all(keep):
                cov_reg = corr_reg * np.outer(sd, sd)
                self.correction_var[idx] = np.diag(cov_reg) - np.diag(cov)
                self.sdev[idx] = np.sqrt(np.diag(cov_reg))
            else:
                cov_reg = cov
                self.sdev[idx] = sd
            self.cov_blocks.append((idx, cov_reg))
        self.nchiv = sum(len(w) for _, w in self.i_invwgts)

    @classmethod
    def from_dense(cls, mean, cov, svdcut=1e-12, eps=None):
        cov = np.asarray(cov, float)
        sdev = np.sqrt(np.diag(cov))
        blocks = [(c, cov[np.ix_(c, c)]) for c in find_blocks(cov) if c.size > 1]
        return cls(mean, sdev, blocks, svdcut=svdcut, eps=eps)

    def icov(self):
        """sum_w outer(w,w): the inverse regulated covariance (small n only)."""
        n = self.mean.size
        ans = np.zeros((n, n))
        i, w = self.i_invwgts[0]
        ans[i, i] = w ** 2
        for i, W in self.i_invwgts[1:]:
            ans[np.ix_(i, i)] += W.T @ W
        return ans
