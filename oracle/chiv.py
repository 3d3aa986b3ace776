"""Whitened residual vector (ORACLE ONLY).

Restates ``chiv.__call__`` (src/lsqfit/_utilities.pyx:65-94):
``delta = concat(fcn(p), p) - mean`` (:74-77, or without ``p`` when there is
no prior), the 1x1 blocks first as ``wgts * delta[iw]`` (:85-89), then each
correlated block as ``wgt @ delta[iw]`` (:90-93).  With a ``Dual`` parameter
vector (the role gvar's GVars play at src/lsqfit/_gsl.pyx:748) the same code
path yields the Jacobian rows (``dot``, _utilities.pyx:20-36).
"""
import numpy as np

from .dual import Dual, concatenate


class Chiv:
    def __init__(self, pdf, fcn, noprior):
        self.mean = pdf.mean
        self.nw = pdf.nchiv
        self.inv_wgts = pdf.i_invwgts
        self.fcn = fcn
        self.noprior = noprior

    def _delta(self, p):
        f = self.fcn(p)
        if self.noprior:
            parts = [f]
        else:
            parts = [f, p]
        d = concatenate([q if isinstance(q, Dual) else np.asarray(q, float).reshape(-1)
                         for q in parts])
        return d - self.mean

    def __call__(self, p):
        delta = self._delta(p)
        if isinstance(delta, Dual):
            val = np.zeros(self.nw)
            der = np.zeros((self.nw, delta.der.shape[-1]))
        else:
            val = np.zeros(self.nw)
            der = None
        iw, wgts = self.inv_wgts[0]
        i1, i2 = 0, len(iw)
        if i2 > 0:
            if der is None:
                val[i1:i2] = wgts * delta[iw]
            else:
                val[i1:i2] = wgts * delta.val[iw]
                der[i1:i2] = wgts[:, None] * delta.der[iw]
        for iw, wgt in self.inv_wgts[1:]:
            i1 = i2
            i2 += len(wgt)
            if der is None:
                val[i1:i2] = wgt @ delta[iw]
            else:
                val[i1:i2] = wgt @ delta.val[iw]
                der[i1:i2] = wgt @ delta.der[iw]
        return val if der is None else Dual(val, der)

    def residual(self, p):
        return self(np.asarray(p, float))

    def jacobian(self, p):
        return self(Dual.seed(p)).der
