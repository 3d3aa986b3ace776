"""Trace a Python fit function into the device tape.

The reference hands its plugins ``f``, built from the user's ``fcn(x, p)`` / ``fcn(p)`` with array- or dict-valued
parameters and outputs (``_unpack_fcn``, ``flatfcn_aa/ad/da/dd``, src/lsqfit/__init__.py:1997-2042) and differentiates
it by CALLING it on overloaded numbers (``gvar.valder``, src/lsqfit/_gsl.pyx:742-760).  A GPU cannot call Python, but
the same trick records what the function does: :func:`trace` calls ``fcn`` ONCE on tracer arrays (:class:`TArr`:
operator overloads, ``__array_ufunc__`` / ``__array_function__`` for numpy's functions, indexing, broadcasting against
``x`` and any constants the function closes over), turns the recorded expression graph into one RPN program per
contiguous range of output rows (the opcodes of :mod:`lsqfit_amd.models`; arrays of constants the function uses
row by row become predictor columns) and hands that to the formula compiler the string front end
(:func:`lsqfit_amd.expr`, :func:`lsqfit_amd.piecewise`) already feeds.

A function whose control flow depends on parameter values (``if p[0] > 0``, ``np.where(p > 0, ...)``, ``math.exp(p[0])``)
cannot be recorded by one call and is refused with :class:`TraceError`; a selection on the DATA (``np.where(x < 1, ...)``,
boolean masks of ``x``) is a piecewise model and is recorded as one formula per range of rows.
"""
import numpy as np

from .models import MODEL_TAPE, OP, TAPE_MAX_PARAM, TAPE_MAX_STACK, Model


class TraceError(TypeError):
    pass


_UNARY = dict(exp='EXP', log='LOG', sin='SIN', cos='COS', tan='TAN', arctan='ATAN', sqrt='SQRT', sinh='SINH', cosh='COSH',
              tanh='TANH', arcsin='ASIN', arccos='ACOS', absolute='ABS', fabs='ABS', negative='NEG')
_BINARY = dict(add='ADD', subtract='SUB', multiply='MUL', true_divide='DIV', divide='DIV', power='POW', float_power='POW')
_NP_UN = dict(EXP=np.exp, LOG=np.log, SIN=np.sin, COS=np.cos, TAN=np.tan, ATAN=np.arctan, SQRT=np.sqrt, SINH=np.sinh,
              COSH=np.cosh, TANH=np.tanh, ASIN=np.arcsin, ACOS=np.arccos, ABS=np.abs, NEG=np.negative)
_NP_BIN = dict(ADD=np.add, SUB=np.subtract, MUL=np.multiply, DIV=np.true_divide, POW=np.power)
_COMPARE = {'greater', 'greater_equal', 'less', 'less_equal', 'equal', 'not_equal', 'logical_and', 'logical_or', 'logical_not',
            'maximum', 'minimum', 'fmax', 'fmin', 'sign', 'heaviside', 'isnan', 'isfinite', 'isinf', 'floor', 'ceil', 'rint'}
MAX_PROGRAMS = 512
TAPE_MAX_CODE = 16384     # LSQAMD_TAPE_MAX_CODE (include/lsqfit_amd.h)
MANY_PROGRAMS = 48       # beyond this many row ranges a parameter selected row by row is written with indicator columns
_FOLD = [True]          # arithmetic between ARRAYS of constants is done by numpy during the recording (see trace(fold=))


def _foldable(*nodes):
    return _FOLD[0] or all(n.shape == () for n in nodes)


def _control_flow(what):
    return TraceError('cannot trace %s of a parameter-dependent value: the fit function\'s control flow or a non-smooth '
                      'selection depends on the parameters, which one recording call cannot capture (write the model '
                      'with arithmetic and the functions exp log sqrt sin cos tan arctan sinh cosh tanh arcsin arccos abs)'
                      % what)


class TArr(object):
    """A traced array: shape + how its elements follow from parameters and constants."""
    __array_priority__ = 1.0e6
    __slots__ = ('op', 'args', 'shape', 'aux')

    def __init__(self, op, args, shape, aux=None):
        self.op, self.args, self.shape, self.aux = op, args, tuple(int(s) for s in shape), aux

    # -- shape protocol ---------------------------------------------------------------------
    @property
    def size(self):
        return int(np.prod(self.shape, dtype=np.int64))

    @property
    def ndim(self):
        return len(self.shape)

    dtype = np.dtype(object)

    def __len__(self):
        if not self.shape:
            raise TypeError('len() of a 0-d traced value')
        return self.shape[0]

    def __iter__(self):
        if not self.shape:
            raise TypeError('iteration over a 0-d traced value')
        for i in range(self.shape[0]):
            yield self[i]

    def __repr__(self):
        return 'TArr(%s, shape=%s)' % (self.op, self.shape)

    # -- refusals ----------------------------------------------------------------------------
    def __bool__(self):
        raise _control_flow('the truth value')

    def __float__(self):
        raise _control_flow('float()')

    __int__ = __index__ = __float__

    def _cmp(self, other):
        raise _control_flow('a comparison')

    __lt__ = __le__ = __gt__ = __ge__ = _cmp
    __hash__ = object.__hash__

    def __eq__(self, other):
        raise _control_flow('a comparison')

    def __ne__(self, other):
        raise _control_flow('a comparison')

    def __setitem__(self, key, value):
        raise TraceError('item assignment into a traced array is not supported: build the output with numpy.concatenate / '
                         'numpy.stack, a list, or a dictionary')

    # -- arithmetic ---------------------------------------------------------------------------
    def __add__(self, o):
        return _binary('ADD', self, o)

    def __radd__(self, o):
        return _binary('ADD', o, self)

    def __sub__(self, o):
        return _binary('SUB', self, o)

    def __rsub__(self, o):
        return _binary('SUB', o, self)

    def __mul__(self, o):
        return _binary('MUL', self, o)

    def __rmul__(self, o):
        return _binary('MUL', o, self)

    def __truediv__(self, o):
        return _binary('DIV', self, o)

    def __rtruediv__(self, o):
        return _binary('DIV', o, self)

    def __pow__(self, o):
        return _binary('POW', self, o)

    def __rpow__(self, o):
        return _binary('POW', o, self)

    def __neg__(self):
        return _unary('NEG', self)

    def __pos__(self):
        return self

    def __abs__(self):
        return _unary('ABS', self)

    def __matmul__(self, o):
        return _matmul(self, o)

    def __rmatmul__(self, o):
        return _matmul(o, self)

    # the methods numpy's object loops and gvar-style code call
    def exp(self):
        return _unary('EXP', self)

    def log(self):
        return _unary('LOG', self)

    def sqrt(self):
        return _unary('SQRT', self)

    def sin(self):
        return _unary('SIN', self)

    def cos(self):
        return _unary('COS', self)

    def tan(self):
        return _unary('TAN', self)

    def arctan(self):
        return _unary('ATAN', self)

    def sinh(self):
        return _unary('SINH', self)

    def cosh(self):
        return _unary('COSH', self)

    def tanh(self):
        return _unary('TANH', self)

    def arcsin(self):
        return _unary('ASIN', self)

    def arccos(self):
        return _unary('ACOS', self)

    # -- indexing and shape changes: all of them are gathers through an index map ---------------
    def _index_map(self):
        return np.arange(self.size, dtype=np.int64).reshape(self.shape)

    def __getitem__(self, key):
        for k in (key if isinstance(key, tuple) else (key,)):
            if isinstance(k, TArr):
                raise _control_flow('an index')
        return _gather(self, self._index_map()[key])

    def reshape(self, *shape, **kw):
        if len(shape) == 1 and not isinstance(shape[0], (int, np.integer)):
            shape = tuple(shape[0])
        return _gather(self, self._index_map().reshape(shape))

    def ravel(self, order='C'):
        return _gather(self, self._index_map().ravel())

    flatten = ravel

    @property
    def flat(self):
        return self.ravel()

    def transpose(self, *axes):
        if len(axes) == 1 and not isinstance(axes[0], (int, np.integer)):
            axes = axes[0]
        return _gather(self, self._index_map().transpose(*axes) if axes and axes[0] is not None else self._index_map().T)

    @property
    def T(self):
        return _gather(self, self._index_map().T)

    def squeeze(self, axis=None):
        return _gather(self, self._index_map().squeeze(axis))

    def sum(self, axis=None, dtype=None, out=None, keepdims=False, **kw):
        return _sum(self, axis, keepdims)

    def mean(self, axis=None, dtype=None, out=None, keepdims=False, **kw):
        s = _sum(self, axis, keepdims)
        return s / (self.size / max(s.size, 1))

    def dot(self, o):
        return _dot(self, o)

    def cumsum(self, axis=None):
        return _cumsum(self, axis)

    def prod(self, axis=None, dtype=None, out=None, keepdims=False, **kw):
        return _prod(self, axis, keepdims)

    def cumprod(self, axis=None):
        return _cumprod(self, axis)

    def var(self, axis=None, dtype=None, out=None, ddof=0, keepdims=False, **kw):
        return _var(self, axis, ddof, keepdims)

    def std(self, axis=None, dtype=None, out=None, ddof=0, keepdims=False, **kw):
        return _unary('SQRT', _var(self, axis, ddof, keepdims))

    def trace(self, offset=0, axis1=0, axis2=1, **kw):
        return _sum(self.diagonal(offset, axis1, axis2), -1)

    def diagonal(self, offset=0, axis1=0, axis2=1):
        return _gather(self, self._index_map().diagonal(offset, axis1, axis2))

    def swapaxes(self, a, b):
        return _gather(self, self._index_map().swapaxes(a, b))

    def repeat(self, repeats, axis=None):
        return _gather(self, self._index_map().repeat(repeats, axis))

    def take(self, indices, axis=None, **kw):
        return _gather(self, self._index_map().take(indices, axis))

    def copy(self, *a, **kw):
        return self

    def astype(self, dtype, *a, **kw):
        if np.dtype(dtype).kind not in 'fO':
            raise _control_flow('a conversion to %s' % np.dtype(dtype))
        return self

    def conj(self):
        return self

    conjugate = conj
    real = property(lambda self: self)

    def _selection(self, *a, **kw):
        raise _control_flow('max / min / sort / round / clip')

    max = min = argmax = argmin = sort = argsort = round = clip = ptp = any = all = nonzero = item = tolist = _selection

    # -- numpy protocols ----------------------------------------------------------------------
    def __array_ufunc__(self, ufunc, method, *inputs, **kw):
        if kw.get('out') is not None:
            raise TraceError('out= is not supported while tracing')
        name = ufunc.__name__
        if method == '__call__':
            if name in _UNARY:
                return _unary(_UNARY[name], inputs[0])
            if name in _BINARY:
                return _binary(_BINARY[name], inputs[0], inputs[1])
            if name == 'positive':
                return _lift(inputs[0])
            if name == 'square':
                return _binary('POW', inputs[0], 2.0)
            if name == 'reciprocal':
                return _binary('DIV', 1.0, inputs[0])
            if name == 'exp2':
                return _unary('EXP', _binary('MUL', inputs[0], float(np.log(2.0))))
            if name == 'expm1':       # e^x - 1 = 2 e^(x/2) sinh(x/2): no cancellation for small x (exp(x) - 1 loses the digits numpy's keeps)
                h = _binary('MUL', 0.5, inputs[0])
                return _binary('MUL', 2.0, _binary('MUL', _unary('EXP', h), _unary('SINH', h)))
            if name == 'log1p':
                # the tape has no log1p: log(1 + x) -- for |x| << 1 the sum 1 + x keeps only ~16 + log10|x| digits of x, numpy's
                # log1p keeps them all (documented in INTEGRATION.md; a fit that depends on it should fit log(1 + x)'s argument)
                return _unary('LOG', _binary('ADD', 1.0, inputs[0]))
            if name == 'log10':
                return _binary('DIV', _unary('LOG', inputs[0]), float(np.log(10.0)))
            if name == 'log2':
                return _binary('DIV', _unary('LOG', inputs[0]), float(np.log(2.0)))
            if name == 'matmul':
                return _matmul(inputs[0], inputs[1])
            if name == 'hypot':
                return _unary('SQRT', _binary('ADD', _binary('POW', inputs[0], 2.0), _binary('POW', inputs[1], 2.0)))
            if name == 'arctan2':
                # half-angle forms: 2 atan(y / (r + x)) is exact in the right half plane and NaN / inaccurate on and near the
                # negative real axis; 2 atan((r - x) / y) is the form for x < 0.  Which one a row needs depends on the SIGN OF
                # x: known when x is data (a selection by row, like numpy.where on the data), control flow when x depends on the
                # parameters -- refused rather than silently wrong on the left half plane
                y, x = inputs
                if isinstance(x, TArr) and x.op != 'data':
                    raise TraceError('numpy.arctan2(y, x) with x depending on the parameters: the branch is chosen by the sign of x, which '
                                     'one recording cannot capture (write 2*arctan(y/(sqrt(x*x + y*y) + x)) if x > 0 is guaranteed)')
                xd = np.asarray(x.aux if isinstance(x, TArr) else x, float)
                r = _unary('SQRT', _binary('ADD', _binary('POW', xd, 2.0), _binary('POW', y, 2.0)))
                right = _binary('MUL', 2.0, _unary('ATAN', _binary('DIV', y, _binary('ADD', r, xd))))
                if np.all(xd >= 0.0):
                    return right
                left = _binary('MUL', 2.0, _unary('ATAN', _binary('DIV', _binary('SUB', r, xd), y)))
                if np.all(xd < 0.0):
                    return left
                shape = np.broadcast_shapes(xd.shape, _lift(y).shape)
                return _where(np.broadcast_to(xd >= 0.0, shape), right, left)
            if name == 'cbrt':
                raise TraceError('numpy.cbrt has no counterpart on the device tape (x ** (1 / 3) for positive x)')
            if name in ('conjugate', 'real'):
                return _lift(inputs[0])
            if name in _COMPARE:
                raise _control_flow('numpy.' + name)
        elif method == 'reduce' and name == 'add':
            return _sum(inputs[0], kw.get('axis', 0), kw.get('keepdims', False))
        elif method == 'reduce' and name == 'multiply':
            return _prod(inputs[0], kw.get('axis', 0), kw.get('keepdims', False))
        elif method == 'outer' and name in _BINARY:
            a, b = _lift(inputs[0]), _lift(inputs[1])
            return _binary(_BINARY[name], a.reshape(a.shape + (1,) * b.ndim), b)
        raise TraceError('numpy.%s%s has no counterpart on the device tape' % (name, '' if method == '__call__' else '.' + method))

    def __array_function__(self, func, types, args, kwargs):
        h = _FUNCTIONS.get(func)
        if h is None:
            raise TraceError('numpy.%s is not supported while tracing a fit function' % getattr(func, '__name__', str(func)))
        return h(*args, **kwargs)


# ---- construction --------------------------------------------------------------------------------------------------
def _data(a):
    a = np.asarray(a, np.float64)
    return TArr('data', (), a.shape, a)


def _from_objects(o):
    """object ndarray (or nested list) whose elements are numbers and 0-d traced values -> one traced array.
    Elements that are entries of ONE traced array (what numpy leaves behind when a traced vector is assigned into an object
    array or iterated: ``ans[i1:i2] = vector`` in lsqfit's chiv, src/lsqfit/_utilities.pyx:88-93) are put back together as one
    gather of that array, and scalar parameters as one parameter vector: the recording stays a handful of vector nodes instead
    of one node per element (a 177-row residual resolved 100x faster)."""
    o = np.asarray(o, dtype=object) if not isinstance(o, np.ndarray) else o
    flat = o.ravel()
    n = flat.size
    nums = np.zeros(n)
    which = np.zeros(n, np.int64)
    pos = np.arange(n, dtype=np.int64)
    kids = [None]
    groups = {}          # id(parent array) -> [kid slot, parent, [element positions], [indices into the parent]]
    params = None        # [kid slot, [element positions], [parameter indices]]
    for i in range(n):
        e = flat[i]
        if isinstance(e, TArr):
            if e.shape != ():
                if e.size != 1:
                    raise TraceError('ragged traced output: element %d has shape %s' % (i, e.shape))
                e = e.reshape(())
            if e.op == 'gather' and len(e.args) == 1:
                g = groups.get(id(e.args[0]))
                if g is None:
                    g = groups[id(e.args[0])] = [len(kids), e.args[0], [], []]
                    kids.append(None)
                g[2].append(i)
                g[3].append(int(np.asarray(e.aux).reshape(-1)[0]))
                continue
            if e.op == 'param':
                if params is None:
                    params = [len(kids), [], []]
                    kids.append(None)
                params[1].append(i)
                params[2].append(int(np.asarray(e.aux).reshape(-1)[0]))
                continue
            which[i] = len(kids)
            pos[i] = 0
            kids.append(e)
        else:
            nums[i] = float(e)
    for slot, parent, where, idx in groups.values():
        kids[slot] = _gather(parent, np.array(idx, np.int64))
        which[where] = slot
        pos[where] = np.arange(len(where))
    if params is not None:
        slot, where, idx = params
        kids[slot] = TArr('param', (), (len(idx),), np.array(idx, np.int64))
        which[where] = slot
        pos[where] = np.arange(len(where))
    kids[0] = _data(nums)
    if len(kids) == 1:
        return TArr('data', (), o.shape, nums.reshape(o.shape))
    return TArr('cat', tuple(kids), o.shape, (which.reshape(o.shape), pos.reshape(o.shape)))


def _lift(v):
    if isinstance(v, TArr):
        return v
    if isinstance(v, memoryview):
        v = np.asarray(v)
    if isinstance(v, (list, tuple)):
        v = np.array(v, dtype=object) if _has_tracer(v) else np.asarray(v, np.float64)
    if isinstance(v, np.ndarray) and v.dtype == object:
        return _from_objects(v)
    try:
        return _data(v)
    except (TypeError, ValueError):
        raise TraceError('cannot use a %s inside a traced fit function' % type(v).__name__)


def _has_tracer(v):
    if isinstance(v, TArr):
        return True
    if isinstance(v, (list, tuple)):
        return any(_has_tracer(e) for e in v)
    if isinstance(v, np.ndarray) and v.dtype == object:
        return any(isinstance(e, TArr) for e in v.ravel())
    return False


def _gather(a, idx):
    idx = np.asarray(idx, np.int64)
    if a.op == 'data':
        return TArr('data', (), idx.shape, a.aux.ravel()[idx])
    if a.op == 'param':
        return TArr('param', (), idx.shape, a.aux.ravel()[idx])
    if a.op == 'gather':            # compose
        return TArr('gather', a.args, idx.shape, a.aux.ravel()[idx])
    return TArr('gather', (a,), idx.shape, idx)


def _unary(op, a):
    a = _lift(a)
    if a.op == 'data' and _foldable(a):
        with np.errstate(all='ignore'):
            return _data(_NP_UN[op](a.aux))
    return TArr('un', (a,), a.shape, op)


def _binary(op, a, b):
    a, b = _lift(a), _lift(b)
    if a.op == 'data' and b.op == 'data' and _foldable(a, b):
        with np.errstate(all='ignore'):
            return _data(_NP_BIN[op](a.aux, b.aux))
    shape = np.broadcast_shapes(a.shape, b.shape)
    # ``sum(generator)`` starts from the integer 0
    if op == 'ADD' and a.op == 'data' and a.shape == () and a.aux == 0.0 and b.shape == shape:
        return b
    return TArr('bin', (a, b), shape, op)


def _sum(a, axis=None, keepdims=False):
    a = _lift(a)
    if isinstance(axis, tuple):
        for ax in sorted((x % a.ndim for x in axis), reverse=True):
            a = _sum(a, ax, keepdims)
        return a
    if a.op == 'data':
        return _data(a.aux.sum(axis=axis, keepdims=keepdims))
    idx = a._index_map()
    if axis is None:
        mat = idx.reshape(-1)
        shape = (1,) * a.ndim if keepdims else ()
        mat = mat.reshape((mat.size,) + shape)
    else:
        mat = np.moveaxis(idx, axis, 0)
        if keepdims:
            mat = np.expand_dims(mat, (axis % a.ndim) + 1)
        shape = mat.shape[1:]
    if mat.shape[0] == 0:
        return _data(np.zeros(shape))
    if mat.shape[0] == 1:
        return _gather(a, mat[0])
    return TArr('nsum', (a,), shape, mat)


def _cumsum(a, axis=None):
    a = _lift(a)
    if axis is None:
        a, axis = a.ravel(), 0
    parts = [_sum(a[(slice(None),) * (axis % a.ndim) + (slice(0, k + 1),)], axis, True) for k in range(a.shape[axis])]
    return _concatenate(parts, axis)


def _prod(a, axis=None, keepdims=False):
    """product along an axis: a chain of multiplications (the tape has sums of terms, not products of factors)"""
    a = _lift(a)
    if isinstance(axis, tuple):
        for ax in sorted((x % a.ndim for x in axis), reverse=True):
            a = _prod(a, ax, keepdims)
        return a
    if a.op == 'data':
        return _data(a.aux.prod(axis=axis, keepdims=keepdims))
    if axis is None:
        shape = (1,) * a.ndim if keepdims else ()
        a, axis = a.ravel(), 0
        parts = [a[k] for k in range(a.shape[0])]
    else:
        axis = axis % a.ndim
        sl = (slice(None),) * axis
        parts = [a[sl + ((slice(k, k + 1) if keepdims else k),)] for k in range(a.shape[axis])]
        shape = parts[0].shape if parts else tuple(1 if i == axis else n for i, n in enumerate(a.shape) if keepdims or i != axis)
    if not parts:
        return _data(np.ones(shape))
    out = parts[0]
    for q in parts[1:]:
        out = _binary('MUL', out, q)
    return out.reshape(shape)


def _cumprod(a, axis=None):
    a = _lift(a)
    if axis is None:
        a, axis = a.ravel(), 0
    axis = axis % a.ndim
    sl = (slice(None),) * axis
    parts = [a[sl + (slice(0, 1),)]] if a.shape[axis] else []
    for k in range(1, a.shape[axis]):
        parts.append(_binary('MUL', parts[-1], a[sl + (slice(k, k + 1),)]))
    return _concatenate(parts, axis) if parts else a


def _var(a, axis=None, ddof=0, keepdims=False):
    a = _lift(a)
    n = a.size if axis is None else (int(np.prod([a.shape[x] for x in axis])) if isinstance(axis, tuple) else a.shape[axis])
    d = _binary('SUB', a, _binary('DIV', _sum(a, axis, True), float(n)))
    return _binary('DIV', _sum(_binary('MUL', d, d), axis, keepdims), float(n - ddof))


def _norm(a, ord=None, axis=None, keepdims=False):
    if ord not in (None, 2, 'fro') or (ord == 2 and axis is None and _lift(a).ndim > 1):
        raise TraceError('numpy.linalg.norm: only the 2-norm of vectors and the Frobenius norm are traced')
    a = _lift(a)
    return _unary('SQRT', _sum(_binary('MUL', a, a), axis, keepdims))


def _tensordot(a, b, axes=2):
    a, b = _lift(a), _lift(b)
    if isinstance(axes, (int, np.integer)):
        axa, axb = list(range(a.ndim - axes, a.ndim)), list(range(axes))
    else:
        axa, axb = axes
        axa = [axa] if isinstance(axa, (int, np.integer)) else list(axa)
        axb = [axb] if isinstance(axb, (int, np.integer)) else list(axb)
    axa, axb = [x % a.ndim for x in axa], [x % b.ndim for x in axb]
    keep_a, keep_b = [i for i in range(a.ndim) if i not in axa], [i for i in range(b.ndim) if i not in axb]
    k = int(np.prod([a.shape[i] for i in axa], dtype=np.int64))
    if k != int(np.prod([b.shape[i] for i in axb], dtype=np.int64)):
        raise ValueError('shape-mismatch for sum')
    oa, ob = tuple(a.shape[i] for i in keep_a), tuple(b.shape[i] for i in keep_b)
    a2 = a.transpose(keep_a + axa).reshape((int(np.prod(oa, dtype=np.int64)), k))
    b2 = b.transpose(axb + keep_b).reshape((k, int(np.prod(ob, dtype=np.int64))))
    return _matmul(a2, b2).reshape(oa + ob)


def _einsum(subscripts, *ops, **kw):
    """explicit or implicit subscripts without ellipses or repeated labels inside one operand: every operand is laid out over
    the union of the labels, multiplied, and the labels that do not appear in the output are summed"""
    if not isinstance(subscripts, str) or '.' in subscripts:
        raise TraceError('numpy.einsum: only subscript strings without ellipses are traced')
    sub = subscripts.replace(' ', '')
    ins, out = sub.split('->') if '->' in sub else (sub, None)
    ins = ins.split(',')
    if len(ins) != len(ops):
        raise ValueError('einsum: %d operands for %d subscript groups' % (len(ops), len(ins)))
    labels = sorted(set(''.join(ins)))
    if out is None:
        out = ''.join(c for c in labels if ''.join(ins).count(c) == 1)
    prod = None
    for lab, o in zip(ins, ops):
        o = _lift(o)
        if len(set(lab)) != len(lab) or len(lab) != o.ndim:
            raise TraceError('numpy.einsum: repeated labels inside one operand are not traced')
        order = [lab.index(c) for c in labels if c in lab]
        o = o.transpose(order) if o.ndim > 1 else o
        o = o.reshape(tuple(o.shape[[c for c in labels if c in lab].index(c)] if c in lab else 1 for c in labels))
        prod = o if prod is None else _binary('MUL', prod, o)
    gone = tuple(i for i, c in enumerate(labels) if c not in out)
    if gone:
        prod = _sum(prod, gone)
    left = [c for c in labels if c in out]
    return prod.transpose([left.index(c) for c in out]) if len(out) > 1 else prod


def _via_index(func):
    """a numpy function that only rearranges / repeats elements: applied to the index map"""
    def h(a, *args, **kw):
        a = _lift(a)
        r = func(a._index_map(), *args, **kw)
        return [_gather(a, q) for q in r] if isinstance(r, (list, tuple)) else _gather(a, r)
    return h


def _like(value):
    def h(a, *args, **kw):
        shape = kw.get('shape') or _lift(a).shape
        return _data(np.full(shape, value if value is not None else (args[0] if args else kw.get('fill_value'))))
    return h


def _select_triangle(func):
    def h(a, k=0):
        a = _lift(a)
        keep = func(np.ones(a.shape, bool), k)
        return _where(keep, a, 0.0)
    return h


def _diag(a, k=0):
    a = _lift(a)
    if a.ndim == 2:
        return a.diagonal(k)
    n = a.size + abs(k)
    idx = np.diag(np.arange(1, a.size + 1), k)       # 0 = no element
    return _where(idx > 0, _gather(a, np.maximum(idx - 1, 0)), 0.0)


def _concatenate(arrays, axis=0, **kw):
    kids = [_lift(a) for a in arrays]
    if axis is None:
        kids, axis = [k.ravel() for k in kids], 0
    if all(k.op == 'data' for k in kids):
        return _data(np.concatenate([k.aux for k in kids], axis))
    which = np.concatenate([np.full(k.shape, i, np.int64) for i, k in enumerate(kids)], axis)
    pos = np.concatenate([k._index_map() for k in kids], axis)
    return TArr('cat', tuple(kids), which.shape, (which, pos))


def _stack(arrays, axis=0, **kw):
    kids = [_lift(a) for a in arrays]
    shape = np.broadcast_shapes(*[k.shape for k in kids])
    kids = [_broadcast_to(k, shape) for k in kids]
    ax = axis % (len(shape) + 1)
    return _concatenate([_gather(k, np.expand_dims(k._index_map(), ax)) for k in kids], ax)


def _broadcast_to(a, shape, **kw):
    a = _lift(a)
    return _gather(a, np.broadcast_to(a._index_map(), shape))


def _matmul(a, b):
    a, b = _lift(a), _lift(b)
    if a.ndim == 0 or b.ndim == 0:
        raise TraceError('matmul of a 0-d value')
    a2 = a if a.ndim > 1 else a.reshape((1,) + a.shape)
    b2 = b if b.ndim > 1 else b.reshape(b.shape + (1,))
    ia = np.expand_dims(a2._index_map(), -1)            # (..., m, k, 1)
    ib = np.expand_dims(b2._index_map(), -3)            # (..., 1, k, n)
    out = _sum(_binary('MUL', _gather(a2, ia), _gather(b2, ib)), -2)
    if a.ndim == 1:
        out = out.squeeze(-2)
    if b.ndim == 1:
        out = out.squeeze(-1)
    return out


def _dot(a, b, **kw):
    a, b = _lift(a), _lift(b)
    if a.ndim == 0 or b.ndim == 0:
        return _binary('MUL', a, b)
    if b.ndim <= 2:
        return _matmul(a, b)
    raise TraceError('numpy.dot with more than two dimensions on the right')


def _outer(a, b, **kw):
    a, b = _lift(a).ravel(), _lift(b).ravel()
    return _binary('MUL', a.reshape((a.size, 1)), b.reshape((1, b.size)))


def _where(cond, a=None, b=None, **kw):
    """``numpy.where(cond, a, b)`` with a condition that does not depend on the parameters (``x < x0``: numpy has evaluated it to a
    boolean array before the tracer sees it) is a selection by ROW -- a piecewise model, one formula per range of rows; a
    condition on parameter values is control flow one recording cannot capture."""
    if isinstance(cond, TArr) or _has_tracer(cond) or a is None or b is None:
        raise _control_flow('numpy.where')
    cond = np.asarray(cond, bool)
    a, b = _lift(a), _lift(b)
    shape = np.broadcast_shapes(cond.shape, a.shape, b.shape)
    a, b = _broadcast_to(a, shape), _broadcast_to(b, shape)
    which = np.where(np.broadcast_to(cond, shape), 0, 1).astype(np.int64)
    pos = np.arange(int(np.prod(shape, dtype=np.int64)), dtype=np.int64).reshape(shape)
    if a.op == 'data' and b.op == 'data':
        return _data(np.where(np.broadcast_to(cond, shape), a.aux, b.aux))
    return TArr('cat', (a, b), shape, (which, pos))


_FUNCTIONS = {
    np.sum: lambda a, axis=None, dtype=None, out=None, keepdims=False, **kw: _sum(a, axis, keepdims),
    np.mean: lambda a, axis=None, dtype=None, out=None, keepdims=False, **kw: _lift(a).mean(axis, keepdims=keepdims),
    np.cumsum: lambda a, axis=None, **kw: _cumsum(a, axis),
    np.concatenate: _concatenate,
    np.stack: _stack,
    np.hstack: lambda t, **kw: _concatenate([np.atleast_1d(x) if not isinstance(x, TArr) else (x if x.ndim else x.reshape(1)) for x in t],
                                            0 if _lift(t[0]).ndim <= 1 else 1),
    np.vstack: lambda t, **kw: _concatenate([(lambda k: k if k.ndim >= 2 else k.reshape((1, k.size)))(_lift(x)) for x in t], 0),
    np.reshape: lambda a, *s, **kw: _lift(a).reshape(*(s or (kw.get('newshape', kw.get('shape')),))),
    np.transpose: lambda a, axes=None: _lift(a).transpose(axes),
    np.ravel: lambda a, order='C': _lift(a).ravel(),
    np.squeeze: lambda a, axis=None: _lift(a).squeeze(axis),
    np.expand_dims: lambda a, axis: _gather(_lift(a), np.expand_dims(_lift(a)._index_map(), axis)),
    np.broadcast_to: _broadcast_to,
    np.atleast_1d: lambda a: _lift(a) if _lift(a).ndim else _lift(a).reshape(1),
    np.outer: _outer,
    np.dot: _dot,
    np.matmul: lambda a, b, **kw: _matmul(a, b),
    np.shape: lambda a: _lift(a).shape,
    np.size: lambda a, axis=None: _lift(a).size if axis is None else _lift(a).shape[axis],
    np.ndim: lambda a: _lift(a).ndim,
    np.where: _where,
    np.absolute: lambda a: _unary('ABS', a),
    np.take: lambda a, indices, axis=None, **kw: _gather(_lift(a), np.take(_lift(a)._index_map(), indices, axis)),
    np.flip: lambda a, axis=None: _gather(_lift(a), np.flip(_lift(a)._index_map(), axis)),
    np.moveaxis: lambda a, s, d: _gather(_lift(a), np.moveaxis(_lift(a)._index_map(), s, d)),
    np.swapaxes: lambda a, s, d: _gather(_lift(a), np.swapaxes(_lift(a)._index_map(), s, d)),
    np.diff: lambda a, n=1, axis=-1, **kw: _diff(a, axis),
    np.prod: lambda a, axis=None, dtype=None, out=None, keepdims=False, **kw: _prod(a, axis, keepdims),
    np.cumprod: lambda a, axis=None, **kw: _cumprod(a, axis),
    np.var: lambda a, axis=None, dtype=None, out=None, ddof=0, keepdims=False, **kw: _var(a, axis, ddof, keepdims),
    np.std: lambda a, axis=None, dtype=None, out=None, ddof=0, keepdims=False, **kw: _unary('SQRT', _var(a, axis, ddof, keepdims)),
    np.linalg.norm: _norm,
    np.tensordot: _tensordot,
    np.einsum: _einsum,
    np.inner: lambda a, b: _tensordot(a, b, ((-1,), (-1,))) if _lift(a).ndim and _lift(b).ndim else _binary('MUL', a, b),
    np.vdot: lambda a, b: _sum(_binary('MUL', _lift(a).ravel(), _lift(b).ravel())),
    np.trace: lambda a, offset=0, axis1=0, axis2=1, **kw: _lift(a).trace(offset, axis1, axis2),
    np.diagonal: lambda a, offset=0, axis1=0, axis2=1: _lift(a).diagonal(offset, axis1, axis2),
    np.diag: _diag,
    np.triu: _select_triangle(np.triu),
    np.tril: _select_triangle(np.tril),
    np.tile: _via_index(np.tile),
    np.repeat: _via_index(np.repeat),
    np.roll: _via_index(np.roll),
    np.rot90: _via_index(np.rot90),
    np.fliplr: _via_index(np.fliplr),
    np.flipud: _via_index(np.flipud),
    np.atleast_2d: _via_index(np.atleast_2d),
    np.split: _via_index(np.split),
    np.array_split: _via_index(np.array_split),
    np.delete: _via_index(np.delete),
    np.column_stack: lambda t, **kw: _concatenate([(lambda k: k.reshape((k.size, 1)) if k.ndim < 2 else k)(_lift(x)) for x in t], 1),
    np.append: lambda a, v, axis=None: _concatenate([a, v], axis),
    np.zeros_like: _like(0.0),
    np.ones_like: _like(1.0),
    np.empty_like: _like(0.0),
    np.full_like: _like(None),
    np.copy: lambda a, **kw: _lift(a),
    np.real: lambda a: _lift(a),
    np.conj: lambda a: _lift(a),
    np.square: lambda a: _binary('POW', a, 2.0),
    np.average: lambda a, axis=None, weights=None, **kw: _average(a, axis, weights, **kw),
}
for _name in ('amax', 'amin', 'max', 'min', 'argmax', 'argmin', 'sort', 'argsort', 'clip', 'round', 'around', 'median', 'percentile',
              'quantile', 'nanmax', 'nanmin', 'maximum', 'minimum', 'sign', 'floor', 'ceil', 'trunc', 'isclose', 'allclose', 'array_equal',
              'searchsorted', 'digitize', 'unique', 'nonzero', 'argwhere', 'any', 'all', 'interp', 'select', 'piecewise', 'heaviside'):
    if hasattr(np, _name):
        _FUNCTIONS[getattr(np, _name)] = (lambda n: lambda *a, **kw: (_ for _ in ()).throw(_control_flow('numpy.' + n)))(_name)


def _average(a, axis=None, weights=None, returned=False, keepdims=False, **kw):
    """numpy.average: 1-D weights lie along `axis` (numpy's rule), the normalisation is the sum of the weights along that axis"""
    a = _lift(a)
    if returned:
        raise TraceError('numpy.average(..., returned=True) is not recorded')
    if weights is None:
        return a.mean(axis, keepdims=keepdims)
    if isinstance(weights, TArr) or _has_tracer(weights):
        raise TraceError('numpy.average with weights that depend on the parameters is not recorded (write sum(w * a) / sum(w))')
    w = np.asarray(weights, float)
    if w.shape != a.shape:
        if axis is None:
            raise TypeError('Axis must be specified when shapes of a and weights differ.')
        if w.ndim != 1:
            raise TypeError('1D weights expected when shapes of a and weights differ.')
        if w.shape[0] != a.shape[axis]:
            raise ValueError('Length of weights not compatible with specified axis.')
        shp = [1] * a.ndim
        shp[axis % a.ndim] = w.size
        w = np.broadcast_to(w.reshape(shp), a.shape)
    den = w.sum(axis=axis, keepdims=keepdims)
    if np.any(den == 0.0):
        raise ZeroDivisionError("Weights sum to zero, can't be normalized")
    return _binary('DIV', _sum(_binary('MUL', a, w), axis, keepdims), den)


def _diff(a, axis=-1):
    a = _lift(a)
    idx = a._index_map()
    hi = np.take(idx, np.arange(1, a.shape[axis]), axis)
    lo = np.take(idx, np.arange(0, a.shape[axis] - 1), axis)
    return _binary('SUB', _gather(a, hi), _gather(a, lo))


# ---- resolution: the expression of every output row --------------------------------------------------------------------
# Trees: ('P', int[n]) | ('D', float[n]) | ('U', OP, t) | ('B', OP, l, r) | ('S', [t...]) | ('C', [rows...], [t...])
# leaves carry one entry per row of the set they were resolved for.
def _child_index(idx, shape, cshape):
    """flat indices into an array of ``shape`` -> flat indices into a child of ``cshape`` that broadcasts to it"""
    if cshape == shape:
        return idx
    if int(np.prod(cshape, dtype=np.int64)) == 1:
        return np.zeros_like(idx)
    multi = np.unravel_index(idx, shape)
    multi = multi[len(shape) - len(cshape):]
    multi = tuple(m if s != 1 else np.zeros_like(m) for m, s in zip(multi, cshape))
    return np.ravel_multi_index(multi, cshape)


def _resolve(node, idx, memo):
    key = (id(node), idx.ctypes.data, idx.size)
    hit = memo.get(key)
    if hit is not None and hit[0] is idx:
        return hit[1]
    op = node.op
    if op == 'param':
        t = ('P', node.aux.ravel()[idx])
    elif op == 'data':
        t = ('D', node.aux.ravel()[idx])
    elif op == 'gather':
        t = _resolve(node.args[0], node.aux.ravel()[idx], memo)
    elif op == 'un':
        k = _resolve(node.args[0], idx, memo)
        if k[0] == 'D' and _FOLD[0]:
            with np.errstate(all='ignore'):
                t = ('D', _NP_UN[node.aux](k[1]))
        else:
            t = ('U', node.aux, k)
    elif op == 'bin':
        a, b = node.args
        l = _resolve(a, _child_index(idx, node.shape, a.shape), memo)
        r = _resolve(b, _child_index(idx, node.shape, b.shape), memo)
        if l[0] == 'D' and r[0] == 'D' and _FOLD[0]:
            with np.errstate(all='ignore'):
                t = ('D', _NP_BIN[node.aux](l[1], r[1]))
        else:
            t = ('B', node.aux, l, r)
    elif op == 'nsum':
        mat = node.aux.reshape(node.aux.shape[0], -1)
        t = ('S', [_resolve(node.args[0], np.ascontiguousarray(mat[k][idx]), memo) for k in range(mat.shape[0])])
    elif op == 'cat':
        which, pos = node.aux[0].ravel()[idx], node.aux[1].ravel()[idx]
        kids = np.unique(which)
        if kids.size == 1:
            t = _resolve(node.args[int(kids[0])], np.ascontiguousarray(pos), memo)
        else:
            rows, trees = [], []
            for c in kids:
                sel = np.nonzero(which == c)[0]
                rows.append(sel)
                trees.append(_resolve(node.args[int(c)], np.ascontiguousarray(pos[sel]), memo))
            t = ('C', rows, trees)
    else:
        raise TraceError('internal: unknown node ' + op)
    memo[key] = (idx, t)
    return t


def _restrict(t, sub):
    k = t[0]
    if k == 'P' or k == 'D':
        return (k, t[1][sub])
    if k == 'U':
        return ('U', t[1], _restrict(t[2], sub))
    if k == 'B':
        return ('B', t[1], _restrict(t[2], sub), _restrict(t[3], sub))
    if k == 'S':
        return ('S', [_restrict(x, sub) for x in t[1]])
    raise TraceError('internal: restrict on ' + k)


def _combine(parts, n, build):
    """parts: per operand a list of (rows | None, tree); -> list of (rows | None, build(trees))"""
    if all(len(p) == 1 and p[0][0] is None for p in parts):
        return [(None, build([p[0][1] for p in parts]))]
    labs, full = [], []
    for p in parts:
        lab = np.zeros(n, np.int64)
        norm = []
        for i, (rows, tree) in enumerate(p):
            rows = np.arange(n) if rows is None else rows
            lab[rows] = i
            norm.append((rows, tree))
        full.append(norm)
        labs.append(lab)
    # rows with the same combination of pieces share a formula.  The combinations are found on the STACKED labels (a mixed-radix
    # number  label * len(p) + lab  in int64 overflows silently from 64 two-piece operands on -- a sum over 64 columns of a
    # numpy.where on the data -- and merged rows of different formulas)
    combos, inverse = np.unique(np.stack(labs, axis=1), axis=0, return_inverse=True)
    inverse = np.asarray(inverse).reshape(-1)
    out = []
    for v, picks in enumerate(combos):
        rows = np.nonzero(inverse == v)[0]
        trees = []
        for p, i in zip(full, picks):
            prow, tree = p[int(i)]
            trees.append(_restrict(tree, np.searchsorted(prow, rows)))
        out.append((rows, build(trees)))
    return out


def _pieces(t, n):
    """split a resolved tree at its 'C' nodes: -> [(rows | None (all n), tree without 'C')]; rows ascending"""
    k = t[0]
    if k == 'P' or k == 'D':
        return [(None, t)]
    if k == 'U':
        return [(rows, ('U', t[1], s)) for rows, s in _pieces(t[2], n)]
    if k == 'B':
        return _combine([_pieces(t[2], n), _pieces(t[3], n)], n, lambda ts: ('B', t[1], ts[0], ts[1]))
    if k == 'S':
        return _combine([_pieces(x, n) for x in t[1]], n, lambda ts: ('S', list(ts)))
    out = []
    for sel, sub in zip(t[1], t[2]):
        for rows, s in _pieces(sub, sel.size):
            out.append((sel if rows is None else sel[rows], s))
    return out


def _leaves(t, out):
    k = t[0]
    if k == 'P' or k == 'D':
        out.append(t)
    elif k == 'U':
        _leaves(t[2], out)
    elif k == 'B':
        _leaves(t[2], out)
        _leaves(t[3], out)
    else:
        for x in t[1]:
            _leaves(x, out)
    return out


def _signature(t):
    k = t[0]
    if k == 'P':
        return ('P', int(t[1][0]))
    if k == 'D':
        return 'D'
    if k == 'U':
        return ('U', t[1], _signature(t[2]))
    if k == 'B':
        return ('B', t[1], _signature(t[2]), _signature(t[3]))
    return ('S',) + tuple(_signature(x) for x in t[1])


def _signature_any_parameter(t):
    """the formula's shape with the parameter INDICES left out: rows that differ only in which parameter they read"""
    k = t[0]
    if k == 'P':
        return 'P'
    if k == 'D':
        return 'D'
    if k == 'U':
        return ('U', t[1], _signature_any_parameter(t[2]))
    if k == 'B':
        return ('B', t[1], _signature_any_parameter(t[2]), _signature_any_parameter(t[3]))
    return ('S',) + tuple(_signature_any_parameter(x) for x in t[1])


def _rebuild(t, it):
    k = t[0]
    if k == 'P' or k == 'D':
        return (k, next(it))
    if k == 'U':
        return ('U', t[1], _rebuild(t[2], it))
    if k == 'B':
        l = _rebuild(t[2], it)
        return ('B', t[1], l, _rebuild(t[3], it))
    return ('S', [_rebuild(x, it) for x in t[1]])


class _Emitter(object):
    def __init__(self, consts):
        self.consts = consts
        self.cidx = {v: i for i, v in enumerate(consts)}
        self.code = []
        self.cols, self.colkey = [], {}
        self.depth = self.maxdepth = 0

    def put(self, op, arg=0, push=0):
        self.code.append((OP[op] & 0xff) | (int(arg) << 8))
        self.depth += push
        self.maxdepth = max(self.maxdepth, self.depth)

    def const(self, v):
        v = float(v)
        key = v if v == v else 'nan'
        if key not in self.cidx:
            self.cidx[key] = len(self.consts)
            self.consts.append(v)
        self.put('CONST', self.cidx[key], +1)

    def column(self, a):
        key = a.tobytes()
        if key not in self.colkey:
            self.colkey[key] = len(self.cols)
            self.cols.append(a)
        self.put('X', self.colkey[key], +1)

    def emit(self, t, reorder):
        k = t[0]
        if k == 'P':
            self.put('P', int(t[1][0]), +1)
        elif k == 'D':
            a = t[1]
            if a.size == 0 or np.all(a == a[0]) or (a[0] != a[0] and np.all(a != a)):
                self.const(a[0] if a.size else 0.0)
            else:
                self.column(np.ascontiguousarray(a, np.float64))
        elif k == 'U':
            self.emit(t[2], reorder)
            self.put(t[1])
        elif k == 'B':
            op, l, r = t[1], t[2], t[3]
            if op == 'POW' and r[0] == 'D' and r[1].size and np.all(r[1] == r[1][0]):
                e = float(r[1][0])
                if e == int(e) and abs(e) < 2 ** 20:      # the string front end's rule (models._Compiler.visit_BinOp)
                    self.emit(l, reorder)
                    return self.put('POWI', int(e))
            if reorder and op in ('ADD', 'MUL') and _need(r) > _need(l):
                l, r = r, l
            self.emit(l, reorder)
            self.emit(r, reorder)
            self.put(op, 0, -1)
        else:
            for i, x in enumerate(t[1]):
                self.emit(x, reorder)
                if i:
                    self.put('ADD', 0, -1)


def _need(t):
    k = t[0]
    if k == 'P' or k == 'D':
        return 1
    if k == 'U':
        return _need(t[2])
    if k == 'B':
        a, b = _need(t[2]), _need(t[3])
        return max(a, b + 1)
    return 1 + max(_need(x) for x in t[1])


class Traced(object):
    """What :func:`trace` returns: ``model`` (a tape :class:`Model`, piecewise when rows differ in formula), ``x`` (the
    predictor matrix the tape's X instructions index: N x n_x, built from the arrays the function used row by row),
    ``n_rows``, ``n_param`` and the layouts (``pkeys`` / ``pshapes``, ``ykeys`` / ``yshapes``; keys are None for arrays)."""

    def __init__(self, model, x, pkeys, pshapes, ykeys, yshapes):
        self.model, self.x = model, x
        self.n_rows, self.n_param = x.shape[0], model.n_param
        self.pkeys, self.pshapes, self.ykeys, self.yshapes = pkeys, pshapes, ykeys, yshapes

    def pack_params(self, p):
        """array / dict of parameters -> flat vector in the traced order"""
        if self.pkeys is None or not hasattr(p, 'keys'):      # (a flat vector in the traced order passes through: warm starts)
            flat = np.asarray(p, float).reshape(-1)
            if flat.size != self.n_param:
                raise ValueError('%d parameter values for a fit function of %d' % (flat.size, self.n_param))
            return flat
        return np.concatenate([np.asarray(p[k], float).reshape(-1) for k in self.pkeys])

    def unpack_params(self, flat):
        flat = np.asarray(flat)
        if self.pkeys is None:
            return flat.reshape(self.pshapes[0])
        out, i = ParamDict(), 0
        for k, s in zip(self.pkeys, self.pshapes):
            n = int(np.prod(s, dtype=np.int64))
            out[k] = flat[i:i + n].reshape(s)
            i += n
        return out


class ParamDict(dict):
    """Dictionary of parameters with gvar.BufferDict's distribution keys: a prior entered under ``'log(a)'`` (``'sqrt(a)'``)
    makes ``p['a']`` available to the fit function as ``exp(p['log(a)'])`` (``p['sqrt(a)'] ** 2``) -- log-normal / sqrt-normal
    priors, tests/test_lsqfit.py:1594-1640."""
    _INVERSE = (('log(', lambda v: np.exp(v)), ('sqrt(', lambda v: v * v))

    def __missing__(self, key):
        if isinstance(key, str):
            for prefix, inverse in self._INVERSE:
                full = prefix + key + ')'
                if dict.__contains__(self, full):
                    return inverse(dict.__getitem__(self, full))
        raise KeyError(key)

    def __contains__(self, key):
        if dict.__contains__(self, key):
            return True
        return isinstance(key, str) and any(dict.__contains__(self, pre + key + ')') for pre, _ in self._INVERSE)


def param_tracers(p0):
    """-> (tracer(s) shaped like ``p0``, P, keys | None, shapes): the flattened order is the reference's -- arrays in C
    order, dictionaries key by key in insertion order (``gvar.BufferDict``, src/lsqfit/__init__.py:1935-1993)"""
    if hasattr(p0, 'keys'):
        keys, shapes, out, i = list(p0.keys()), [], ParamDict(), 0
        for k in keys:
            s = np.shape(p0[k])
            n = int(np.prod(s, dtype=np.int64))
            out[k] = TArr('param', (), s, np.arange(i, i + n, dtype=np.int64).reshape(s))
            shapes.append(s)
            i += n
        return out, i, keys, shapes
    s = np.shape(p0)
    n = int(np.prod(s, dtype=np.int64))
    return TArr('param', (), s, np.arange(n, dtype=np.int64).reshape(s)), n, None, [s]


def flatten_output(out, y=None):
    """the function's return value -> (one 1-d traced array, ykeys | None, yshapes), flattened as ``flatfcn_*`` do
    (src/lsqfit/__init__.py:2013-2042): arrays through ``.flat``, dictionaries key by key in the order of the DATA's keys"""
    if hasattr(out, 'keys'):
        keys = list(y.keys()) if (y is not None and hasattr(y, 'keys')) else list(out.keys())
        parts, shapes = [], []
        for k in keys:
            if k not in out:
                raise ValueError('the fit function returned no entry for data key %r' % (k,))
            v = _lift(out[k])
            if y is not None and hasattr(y, 'keys') and np.shape(y[k]) != v.shape:
                raise ValueError('shape mismatch between data and fit function for key %r: %s vs %s' % (k, np.shape(y[k]), v.shape))
            shapes.append(v.shape)
            parts.append(v.ravel())
        return _concatenate(parts, 0) if len(parts) > 1 else parts[0], keys, shapes
    if y is not None and hasattr(y, 'keys'):
        raise ValueError('the data are a dictionary but the fit function returned an array')
    v = _lift(out)
    if y is not None and np.shape(y) != v.shape and np.size(y) != v.size:
        raise ValueError('shape mismatch between data and fit function: %s vs %s' % (np.shape(y), v.shape))
    return v.ravel(), None, [v.shape]


def _select_by_indicator(t):
    """a parameter leaf whose index changes from row to row, written as sum_j [row reads p_j] * p_j over the indices it takes:
    the rows share ONE formula again, the selection travels in 0 / 1 predictor columns"""
    k = t[0]
    if k == 'P':
        a = t[1]
        if a.size == 0 or not np.any(a != a[0]):
            return t
        return ('S', [('B', 'MUL', ('D', (a == j).astype(np.float64)), ('P', np.full(a.size, j, a.dtype))) for j in np.unique(a)])
    if k == 'D':
        return t
    if k == 'U':
        return ('U', t[1], _select_by_indicator(t[2]))
    if k == 'B':
        return ('B', t[1], _select_by_indicator(t[2]), _select_by_indicator(t[3]))
    return ('S', [_select_by_indicator(x) for x in t[1]])


def _count_nodes(t):
    k = t[0]
    if k == 'P' or k == 'D':
        return 1
    if k == 'U':
        return 1 + _count_nodes(t[2])
    if k == 'B':
        return 1 + _count_nodes(t[2]) + _count_nodes(t[3])
    return len(t[1]) + sum(_count_nodes(x) for x in t[1])


def programs_of(flat, many=None):
    """one 1-d traced array -> ([(row0, row1, tree)], N): the resolved expression of every contiguous range of rows that
    shares one formula (same operations, same parameters; constants may differ row by row).  Rows that read DIFFERENT
    parameters through one expression (``p['x'][i]`` in row i -- errors in variables, examples/x-err.py; ``p['norm'][group]``)
    are different formulas, one program per contiguous run -- unless that makes more than MANY_PROGRAMS of them: then the
    selection is written with indicator columns (_select_by_indicator) and the rows are one formula.  ``many``: that threshold
    (default MANY_PROGRAMS; the recorded residual of the plugin path passes 2: its P prior rows ``w_j (p_j - mean_j)`` each read
    another parameter, and P + 1 formulas cost P + 1 run-time compilations -- 8 s for a 9-parameter fit, 27 s for
    examples/x-err.py -- where two do)."""
    many = MANY_PROGRAMS if many is None else many
    try:
        out, N = _programs_of(flat, False)
    except TraceError:
        return _programs_of(flat, True)
    # ... or when the runs are SHORT (a parameter per row, examples/x-err.py: 15 rows, 15 formulas, 15 run-time compilations of
    # ~0.7 s each on a cold cache against one): few rows per formula make the indicator columns cheap and the formulas dear
    if len(out) > many or (len(out) > 2 and N < 16 * len(out)):
        try:
            alt, _ = _programs_of(flat, True)
        except TraceError:
            return out, N
        if len(alt) < len(out) and sum(_count_nodes(t) for _, _, t in alt) <= TAPE_MAX_CODE:
            return alt, N
    return out, N


def _programs_of(flat, by_indicator):
    N = flat.size
    tree = _resolve(flat, np.arange(N, dtype=np.int64), {})
    pieces = []
    for rows, t in _pieces(tree, N):
        rows = np.arange(N) if rows is None else rows
        # rows that read different parameters through the same expression (p[index_array]) are different formulas
        pl = [l[1] for l in _leaves(t, []) if l[0] == 'P']
        if by_indicator and pl and any(a.size and np.any(a != a[0]) for a in pl):
            pieces.append((rows, _select_by_indicator(t)))
        elif pl and any(a.size and np.any(a != a[0]) for a in pl):
            mat = np.stack(pl, 1)
            uniq, inv = np.unique(mat, axis=0, return_inverse=True)
            inv = inv.reshape(-1)
            for g in range(uniq.shape[0]):
                sub = np.nonzero(inv == g)[0]
                pieces.append((rows[sub], _restrict(t, sub)))
        else:
            pieces.append((rows, t))
    # the same formula reached along different paths (an output assembled element by element) is one formula
    merged, order = {}, []
    for rows, t in pieces:
        if rows.size == 0:
            continue
        # by_indicator: pieces that differ only in WHICH parameter a leaf reads (the P prior rows of a recorded residual, the
        # rows of an errors-in-variables model assembled element by element) are one formula too -- the selection becomes
        # 0 / 1 predictor columns below
        sig = _signature_any_parameter(t) if by_indicator else _signature(t)
        if sig not in merged:
            merged[sig] = []
            order.append(sig)
        merged[sig].append((rows, t))
    groups = []
    for sig in order:
        lst = merged[sig]
        if len(lst) == 1:
            rows, t = lst[0]
        else:
            rows = np.concatenate([r for r, _ in lst])
            leaf_lists = [_leaves(t, []) for _, t in lst]
            arrays = [np.concatenate([ll[i][1] for ll in leaf_lists]) for i in range(len(leaf_lists[0]))]
            t = _rebuild(lst[0][1], iter(arrays))
            if by_indicator and any(l[0] == 'P' and l[1].size and np.any(l[1] != l[1][0]) for l in _leaves(t, [])):
                o = np.argsort(rows, kind='stable')          # (the indicator columns are built per row: sort first)
                rows, t = rows[o], _restrict(t, o)
                t = _select_by_indicator(t)
        o = np.argsort(rows, kind='stable')
        if np.any(o != np.arange(o.size)):
            rows, t = rows[o], _restrict(t, o)
        groups.append((rows, t))
    gid = np.full(N, -1, np.int64)
    for g, (rows, _) in enumerate(groups):
        gid[rows] = g
    if N and gid.min() < 0:
        raise TraceError('internal: output rows without a formula')
    out = []
    r0 = 0
    while r0 < N:
        g = gid[r0]
        r1 = r0 + 1
        while r1 < N and gid[r1] == g:
            r1 += 1
        rows, t = groups[g]
        a = int(np.searchsorted(rows, r0))
        out.append((r0, r1, _restrict(t, np.arange(a, a + (r1 - r0)))))
        if len(out) > MAX_PROGRAMS:
            raise TraceError('the fit function uses more than %d different formulas over contiguous ranges of its output; '
                             'order the output so that rows with the same formula are adjacent' % MAX_PROGRAMS)
        r0 = r1
    return out, N


MERGE_MAX_ROWS = 1024      # small fits: several formulas are written as ONE (merge_small_programs)
MERGE_MAX_FORMULAS = 8


def _extend_rows(t, r0, r1, N):
    """a program's tree over rows [r0, r1) -> the same tree over all N rows: outside its own rows every leaf repeats the value
    (parameter index) of the program's FIRST row, so the formula evaluates there exactly as it does in a row of its own"""
    k = t[0]
    if k == 'P' or k == 'D':
        a = np.asarray(t[1])
        full = np.empty(N, a.dtype)
        full[:] = a[0]
        full[r0:r1] = a
        return (k, full)
    if k == 'U':
        return ('U', t[1], _extend_rows(t[2], r0, r1, N))
    if k == 'B':
        return ('B', t[1], _extend_rows(t[2], r0, r1, N), _extend_rows(t[3], r0, r1, N))
    return ('S', [_extend_rows(x, r0, r1, N) for x in t[1]])


def merge_small_programs(progs, N, P):
    """A SMALL fit whose rows follow several formulas (dictionary-valued fit functions -- one formula per key, examples/simple.py;
    the data rows and the prior rows of a recorded residual) as ONE formula  sum_k [row belongs to k] * f_k : every row evaluates
    every formula (on a copy of the first row of that formula's own rows where it does not belong: a value the fit meets anyway,
    never an overflow of its own making) and keeps one.  0 * finite is an exact zero, so values and derivatives are those of
    the separate formulas bit for bit.  What it buys: one run-time compilation instead of k, and the whole fit in ONE launch
    (the one-launch kernel runs one compiled tape: 3-10x on fits of this size, profiles/r06_small_fits.txt).  Not applied to
    larger fits, where the k-fold arithmetic and the extra predictor columns would cost more than they save."""
    k = len(progs)
    if k < 2 or k > MERGE_MAX_FORMULAS or N > MERGE_MAX_ROWS or P > 32:
        return progs
    if sum(_count_nodes(t) for _, _, t in progs) + 3 * k > TAPE_MAX_CODE:
        return progs
    terms = []
    for r0, r1, t in progs:
        ind = np.zeros(N)
        ind[r0:r1] = 1.0
        terms.append(('B', 'MUL', ('D', ind), _extend_rows(t, r0, r1, N)))
    return [(0, N, ('S', terms))]


def emit_programs(progs, N, P, text='traced'):
    """[(row0, row1, tree)] -> (Model, X[N, n_x])"""
    if P > TAPE_MAX_PARAM:
        raise ValueError('tape models support at most %d parameters' % TAPE_MAX_PARAM)
    consts, codes, cols = [], [], []
    for r0, r1, t in progs:
        em = _Emitter(consts)
        em.emit(t, False)
        if em.maxdepth > TAPE_MAX_STACK:
            em2 = _Emitter(consts)
            em2.emit(t, True)
            em = em2
        if em.maxdepth > TAPE_MAX_STACK:
            raise ValueError('expression needs a stack deeper than %d' % TAPE_MAX_STACK)
        codes.append(np.asarray(em.code, np.int32))
        cols.append(em.cols)
    n_x = max([1] + [len(c) for c in cols])
    X = np.zeros((N, n_x))
    for (r0, r1, _), c in zip(progs, cols):
        for j, a in enumerate(c):
            X[r0:r1, j] = a
    if len(consts) > 1024:
        raise ValueError('a traced model may hold at most 1024 distinct constants (arrays of row data do not count); got %d' % len(consts))
    if len(progs) == 1:
        return Model(MODEL_TAPE, P, n_x, tape=codes[0], consts=consts, text=text), X
    programs = [(r1 - r0, c) for (r0, r1, _), c in zip(progs, codes)]
    return Model(MODEL_TAPE, P, n_x, tape=np.concatenate(codes), consts=consts,
                 text='%s{%d formulas}' % (text, len(programs)), programs=programs), X


def trace(fcn, x=False, p0=None, y=None, fold=True, merge_small=True):
    """Record ``fcn(x, p)`` (``fcn(p)`` when ``x is False``, the reference's convention, src/lsqfit/__init__.py:2013-2016)
    for parameters shaped like ``p0`` (an array or a dictionary of arrays) -> :class:`Traced`.  ``y`` (optional: the data's
    mean, array or dictionary) fixes the key order of a dictionary-valued function and is checked for shape.
    ``fold``: arithmetic that involves no parameter (``x**2``, ``cos(2 * pi * x / 12)``) is done by numpy while the function is
    recorded and reaches the device as one more predictor column; ``fold=False`` keeps it on the tape, operation by
    operation, exactly as the formula string front end (:func:`lsqfit_amd.expr`) would write it.
    ``merge_small``: a small fit with several formulas is recorded as one (:func:`merge_small_programs`)."""
    if p0 is None:
        raise ValueError('trace needs p0 (or the prior mean): the shape of the parameters')
    p, P, pkeys, pshapes = param_tracers(p0)
    saved = _FOLD[0]
    _FOLD[0] = bool(fold)
    try:
        if not fold and x is not False:      # (numpy would otherwise do x's arithmetic before the tracer sees any of it)
            x = {k: _data(v) for k, v in x.items()} if hasattr(x, 'keys') else _data(x)
        out = fcn(p) if x is False else fcn(x, p)
        flat, ykeys, yshapes = flatten_output(out, y)
        progs, N = programs_of(flat)
    finally:
        _FOLD[0] = saved
    name = getattr(fcn, '__name__', 'fcn')
    if merge_small:
        progs = merge_small_programs(progs, N, P)
    model, X = emit_programs(progs, N, P, text='traced:' + name)
    return Traced(model, X, pkeys, pshapes, ykeys, yshapes)


def trace_residual(f, P):
    """Record a flat residual function ``f(p) -> r`` -- what lsqfit hands its plugins (``chiv``, src/lsqfit/_utilities.pyx:50-94,
    called on an object array the way ``_c_df`` calls it on GVars, src/lsqfit/_gsl.pyx:748-750) -- with the parameters as an
    object array of 0-d tracers (so that ``p.reshape(shape)``, ``numpy.concatenate((fcn(p), p))`` and assignment into object
    arrays work as they do for GVars).  ``mixed=True`` asks ``chiv`` for its object-array branch (:71-73,:78-80)."""
    tr = TArr('param', (), (P,), np.arange(P, dtype=np.int64))
    p = np.empty(P, object)
    for j in range(P):
        p[j] = tr[j]
    try:
        out = f(p, mixed=True)
    except TypeError as e:
        if isinstance(e, TraceError) or 'mixed' not in str(e):
            raise
        out = f(p)
    flat, _, _ = flatten_output(out, None)
    progs, N = programs_of(flat, many=2)
    progs = merge_small_programs(progs, N, P)
    model, X = emit_programs(progs, N, P, text='traced:residual')
    return Traced(model, X, None, [(P,)], None, [(N,)])


def flatten_mean_err(mean, err):
    """(mean, err) as arrays or dictionaries of arrays -> (flat mean, error spec for :class:`lsqfit_amd.Whitening`).
    Dictionary entries are flattened key by key in insertion order (``gvar.BufferDict``); an entry's error is a standard
    deviation array shaped like its mean, or the covariance matrix (size x size) of its flattened values."""
    if not hasattr(mean, 'keys'):
        m = np.asarray(mean, float)
        e = np.asarray(err, float) if err is not None and not isinstance(err, dict) else err
        if isinstance(e, np.ndarray) and e.shape == m.shape and m.ndim > 1:
            e = e.reshape(-1)
        return m.reshape(-1), e
    from .whiten import _components
    flat, sdev, blocks, r0 = [], [], [], 0
    for k in mean.keys():
        m = np.asarray(mean[k], float)
        e = np.asarray(err[k], float)
        n = m.size
        if e.ndim == 2 and e.shape == (n, n) and (m.ndim != 2 or m.shape != e.shape):
            sdev.append(np.sqrt(np.diag(e)))
            # one block per connected component of the entry's covariance, as gvar.evalcov_blocks finds them
            # (tests/test_lsqfit.py:1011-1012,1047-1050)
            for c in _components(e):
                if c.size > 1:
                    blocks.append((r0, c, e[np.ix_(c, c)]))
        elif e.shape == m.shape or e.size == 1:
            sdev.append(np.broadcast_to(e, m.shape).reshape(-1).astype(float))
        else:
            raise ValueError('error of entry %r has shape %s, its mean %s' % (k, e.shape, m.shape))
        flat.append(m.reshape(-1))
        r0 += n
    flat, sdev = np.concatenate(flat), np.concatenate(sdev)
    if not blocks:
        return flat, sdev
    if all(int(c[-1]) - int(c[0]) + 1 == c.size for _, c, _ in blocks):
        return flat, dict(sdev=sdev, blocks=[(b0 + int(c[0]), cov) for b0, c, cov in blocks])
    # components that interleave inside an entry: hand the whole covariance over, the whitening reorders the rows
    full = np.diag(sdev ** 2)
    for b0, c, cov in blocks:
        full[np.ix_(b0 + c, b0 + c)] = cov
    return flat, full
