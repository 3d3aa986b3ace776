"""Forward-mode AD carrier (ORACLE ONLY).

Restates what the reference gets from ``gvar.valder`` on the Jacobian path:
``_valder = gvar.valder(p*[0.0])`` (src/lsqfit/_gsl.pyx:671) gives P
variables with identity seeds, ``f(_valder + x)`` (:748) propagates them
through the user's fit function, and ``GVar.d`` / ``GVar.der`` is read out
row by row into the dense Jacobian (:752-756, src/lsqfit/_scipy.py:149-154).

``Dual`` holds an array of values ``val`` (shape S) and the dense derivative
``der`` (shape S + (P,)).  Only what the fixtures' fit functions need is
implemented: + - * / **, unary -, exp log sqrt sin cos tan arctan abs,
indexing, and ``sum`` along axis 0.
"""
import numpy as np


class Dual:
    __array_priority__ = 1000

    def __init__(self, val, der):
        self.val = np.asarray(val, float)
        self.der = np.asarray(der, float)

    # -- construction -----------------------------------------------------
    @staticmethod
    def seed(p):
        """valder + p: P variables with unit derivative vectors."""
        p = np.asarray(p, float)
        return Dual(p.copy(), np.eye(p.size).reshape(p.shape + (p.size,)))

    @property
    def shape(self):
        return self.val.shape

    @property
    def size(self):
        return self.val.size

    def __len__(self):
        return len(self.val)

    def __getitem__(self, k):
        return Dual(self.val[k], self.der[k])

    def __iter__(self):
        for i in range(len(self.val)):
            yield self[i]

    def reshape(self, *shape):
        if len(shape) == 1 and not np.isscalar(shape[0]):
            shape = tuple(shape[0])
        P = self.der.shape[-1]
        v = self.val.reshape(shape)
        return Dual(v, self.der.reshape(v.shape + (P,)))

    @property
    def flat(self):
        return self.reshape(-1)

    # -- helpers ----------------------------------------------------------
    @staticmethod
    def _lift(o, P):
        if isinstance(o, Dual):
            return o
        o = np.asarray(o, float)
        return Dual(o, np.zeros(o.shape + (P,)))

    def _P(self):
        return self.der.shape[-1]

    def _chain(self, val, dval):
        return Dual(val, np.asarray(dval)[..., None] * self.der)

    # -- arithmetic -------------------------------------------------------
    def __neg__(self):
        return Dual(-self.val, -self.der)

    def __pos__(self):
        return self

    def __add__(self, o):
        o = Dual._lift(o, self._P())
        v = self.val + o.val
        return Dual(v, np.broadcast_to(self.der, v.shape + (self._P(),))
                    + np.broadcast_to(o.der, v.shape + (self._P(),)))

    __radd__ = __add__

    def __sub__(self, o):
        return self + (-Dual._lift(o, self._P()))

    def __rsub__(self, o):
        return (-self) + o

    def __mul__(self, o):
        o = Dual._lift(o, self._P())
        v = self.val * o.val
        return Dual(v, self.val[..., None] * o.der + o.val[..., None] * self.der)

    __rmul__ = __mul__

    def __truediv__(self, o):
        o = Dual._lift(o, self._P())
        v = self.val / o.val
        return Dual(v, (self.der - v[..., None] * o.der) / o.val[..., None])

    def __rtruediv__(self, o):
        return Dual._lift(o, self._P()) / self

    def __pow__(self, o):
        if isinstance(o, Dual):
            return exp(o * log(self))
        o = np.asarray(o, float)
        v = self.val ** o
        return Dual(v, (o * self.val ** (o - 1.0))[..., None] * self.der)

    def __rpow__(self, o):
        return exp(self * np.log(o))

    def sum(self, axis=0):
        return Dual(self.val.sum(axis=axis), self.der.sum(axis=axis))

    def __repr__(self):
        return 'Dual(%r)' % (self.val,)


def _unary(fv, fd):
    def fn(x):
        if isinstance(x, Dual):
            return x._chain(fv(x.val), fd(x.val))
        return fv(np.asarray(x, float))
    return fn


exp = _unary(np.exp, np.exp)
log = _unary(np.log, lambda v: 1.0 / v)
sqrt = _unary(np.sqrt, lambda v: 0.5 / np.sqrt(v))
sin = _unary(np.sin, np.cos)
cos = _unary(np.cos, lambda v: -np.sin(v))
tan = _unary(np.tan, lambda v: 1.0 / np.cos(v) ** 2)
arctan = _unary(np.arctan, lambda v: 1.0 / (1.0 + v * v))
# gvar (pinned dependency, not under /root/reference): GVar.__abs__ / fabs return self when the mean is >= 0 and -self
# otherwise -- the derivative at 0 is +1, not sign(0) = 0
fabs = _unary(np.fabs, lambda v: np.where(v >= 0.0, 1.0, -1.0))
sinh = _unary(np.sinh, np.cosh)
cosh = _unary(np.cosh, np.sinh)
tanh = _unary(np.tanh, lambda v: 1.0 / np.cosh(np.clip(v, -700.0, 700.0)) ** 2)
arcsin = _unary(np.arcsin, lambda v: 1.0 / np.sqrt(1.0 - v * v))
arccos = _unary(np.arccos, lambda v: -1.0 / np.sqrt(1.0 - v * v))


def concatenate(parts):
    """numpy.concatenate for a mix of Dual / float 1-d pieces."""
    P = None
    for q in parts:
        if isinstance(q, Dual):
            P = q._P()
    if P is None:
        return np.concatenate([np.asarray(q, float).reshape(-1) for q in parts])
    parts = [Dual._lift(q, P).reshape(-1) for q in parts]
    return Dual(np.concatenate([q.val for q in parts]),
                np.concatenate([q.der for q in parts], axis=0))


def stack_sum(terms):
    """Python ``sum(...)`` over Duals/floats (examples/y-vs-x.py:61)."""
    tot = 0.0
    for t in terms:
        tot = t + tot
    return tot


NAMESPACE = dict(exp=exp, log=log, sqrt=sqrt, sin=sin, cos=cos, tan=tan,
                 arctan=arctan, atan=arctan, fabs=fabs, abs=fabs, sinh=sinh, cosh=cosh, tanh=tanh,
                 arcsin=arcsin, asin=arcsin, arccos=arccos, acos=arccos, pi=np.pi)
