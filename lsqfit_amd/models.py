"""Row models f(x_i; p) with device-side forward-mode AD.

The reference's fit function is arbitrary Python evaluated on gvar.GVar
objects to obtain derivatives (src/lsqfit/_gsl.pyx:671,742-760); a GPU cannot
call it.  This module is the replacement surface: a small closed set of wide
"sum" models plus an expression language compiled to an RPN tape
(``LSQAMD_MODEL_TAPE``) that covers the formulas of the reference's examples
(examples/nist.py, examples/p-corr.py:60-61, examples/empbayes.py:27-28).
"""
import ast

import numpy as np

MODEL_COSMIX, MODEL_MULTIEXP, MODEL_TAPE, MODEL_IDENTITY = 1, 2, 3, 4
TAPE_MAX_PARAM, TAPE_MAX_STACK = 4096, 16

OP = dict(CONST=0, X=1, P=2, ADD=3, SUB=4, MUL=5, DIV=6, POW=7, NEG=8, EXP=9, LOG=10,
          SIN=11, COS=12, ATAN=13, SQRT=14, POWI=15, TAN=16, SINH=17, COSH=18, TANH=19, ASIN=20, ACOS=21, ABS=22)
# the functions gvar overloads for GVars (what a reference fit function may call on its parameters); numpy spellings and
# the math-module ones
_FUNCS = dict(exp='EXP', log='LOG', sin='SIN', cos='COS', arctan='ATAN', atan='ATAN', sqrt='SQRT', tan='TAN', sinh='SINH',
              cosh='COSH', tanh='TANH', arcsin='ASIN', asin='ASIN', arccos='ACOS', acos='ACOS', abs='ABS', fabs='ABS')
_BIN = {ast.Add: 'ADD', ast.Sub: 'SUB', ast.Mult: 'MUL', ast.Div: 'DIV', ast.Pow: 'POW'}


class Model:
    """kind + shapes (+ tape).  ``n_param`` = P, ``n_x`` = predictors per row."""

    def __init__(self, kind, n_param, n_x=1, tape=None, consts=None, text=None, programs=None):
        self.kind = int(kind)
        self.n_param = int(n_param)
        self.n_x = int(n_x)
        self.tape = None if tape is None else np.asarray(tape, np.int32)
        self.consts = np.asarray(consts if consts is not None else [], np.float64)
        self.text = text
        # [(n_rows, int32 code)]: one formula per contiguous range of data rows (``piecewise``); ``tape`` is
        # then their concatenation
        self.programs = programs

    def __repr__(self):
        return 'Model(%s, P=%d)' % (self.text or self.kind, self.n_param)


def cosmix(K):
    """f = sum_k a_k cos(w_k x); p = [a_0..a_{K-1}, w_0..w_{K-1}]."""
    return Model(MODEL_COSMIX, 2 * K, 1, text='cosmix')


def multiexp(K):
    """f = sum_k a_k exp(-E_k x); p = [a.., E..] (examples/y-vs-x.py:58-61)."""
    return Model(MODEL_MULTIEXP, 2 * K, 1, text='multiexp')


def identity(P):
    """f_i = p_i (tests/test_lsqfit.py:1815 ``fcn(p) = p``)."""
    return Model(MODEL_IDENTITY, P, 1, text='identity')


class _Compiler(ast.NodeVisitor):
    def __init__(self, params, xnames):
        self.params = {n: i for i, n in enumerate(params)}
        self.xnames = {n: i for i, n in enumerate(xnames)}
        self.code, self.consts = [], []
        self.depth = self.maxdepth = 0

    def emit(self, op, arg=0, push=0):
        self.code.append((OP[op] & 0xff) | (int(arg) << 8))
        self.depth += push
        self.maxdepth = max(self.maxdepth, self.depth)

    def const(self, v):
        v = float(v)
        if v not in self.consts:
            self.consts.append(v)
        self.emit('CONST', self.consts.index(v), +1)

    @staticmethod
    def _constant_value(node):
        """Fold a parameter-free numeric sub-expression (e.g. ``-.5``, ``2*pi``)."""
        try:
            src = ast.Expression(body=node)
            ast.fix_missing_locations(src)
            names = {n.id for n in ast.walk(node) if isinstance(n, ast.Name)}
            if names - {'pi'}:
                return None
            if any(isinstance(n, ast.Call) for n in ast.walk(node)):
                return None
            return float(eval(compile(src, '<const>', 'eval'), {'__builtins__': {}}, {'pi': np.pi}))
        except Exception:
            return None

    def visit(self, node):
        v = self._constant_value(node)
        if v is not None:
            return self.const(v)
        return super().visit(node)

    def visit_Name(self, node):
        if node.id in self.params:
            self.emit('P', self.params[node.id], +1)
        elif node.id in self.xnames:
            self.emit('X', self.xnames[node.id], +1)
        else:
            raise ValueError('unknown name %r in model expression' % node.id)

    def visit_UnaryOp(self, node):
        self.visit(node.operand)
        if isinstance(node.op, ast.USub):
            self.emit('NEG')
        elif not isinstance(node.op, ast.UAdd):
            raise ValueError('unsupported unary operator')

    def visit_BinOp(self, node):
        if type(node.op) not in _BIN:
            raise ValueError('unsupported operator %s' % type(node.op).__name__)
        if isinstance(node.op, ast.Pow):
            e = self._constant_value(node.right)
            if e is not None and e == int(e) and abs(e) < 2 ** 20:
                self.visit(node.left)
                return self.emit('POWI', int(e))
        self.visit(node.left)
        self.visit(node.right)
        self.emit(_BIN[type(node.op)], 0, -1)

    def visit_Call(self, node):
        if not isinstance(node.func, ast.Name) or node.func.id not in _FUNCS or len(node.args) != 1:
            raise ValueError('unsupported function call in model expression')
        self.visit(node.args[0])
        self.emit(_FUNCS[node.func.id])

    def generic_visit(self, node):
        raise ValueError('unsupported syntax in model expression: %s' % type(node).__name__)


def expr(text, params, xnames=('x',)):
    """Compile e.g. ``expr('b1*(1-exp(-b2*x))', ['b1','b2'])`` to a device tape."""
    params = list(params)
    if len(params) > TAPE_MAX_PARAM:
        raise ValueError('tape models support at most %d parameters' % TAPE_MAX_PARAM)
    tree = ast.parse(text.strip(), mode='eval')
    c = _Compiler(params, list(xnames))
    c.visit(tree.body)
    if c.maxdepth > TAPE_MAX_STACK:
        raise ValueError('expression needs a stack deeper than %d' % TAPE_MAX_STACK)
    return Model(MODEL_TAPE, len(params), len(xnames), tape=c.code, consts=c.consts, text=text)


def tape_sum(term, K, params_per_term=('a', 'w'), xnames=('x',)):
    """Sum of K copies of ``term`` (an expression in ``params_per_term`` and x), parameters laid
    out family by family: p = [a_0..a_{K-1}, w_0..w_{K-1}].  Built instruction by instruction (a
    K-term formula string would nest K levels deep in Python's parser):
    ``tape_sum('a*cos(w*x)', 512)`` is cosmix(512) as a general tape with P = 1024."""
    nfam = len(params_per_term)
    code, consts = [], []
    for k in range(K):
        c = _Compiler(list(params_per_term), list(xnames))
        c.consts = consts
        c.visit(ast.parse(term.strip(), mode='eval').body)
        if c.maxdepth + 1 > TAPE_MAX_STACK:
            raise ValueError('term needs a stack deeper than %d' % (TAPE_MAX_STACK - 1))
        for ins in c.code:
            op, arg = ins & 0xff, ins >> 8
            if op == OP['P']:
                arg = arg * K + k
            code.append((op & 0xff) | (int(arg) << 8))
        if k:
            code.append(OP['ADD'])
    return Model(MODEL_TAPE, nfam * K, len(xnames), tape=code, consts=consts, text='sum_%d(%s)' % (K, term))


def piecewise(parts, params, xnames=('x',)):
    """One formula per contiguous range of data rows -- the flattened form of a fit function that
    returns a dictionary or an array built from different expressions (the reference flattens such
    outputs itself: src/lsqfit/__init__.py:1997-2042; examples/simple.py).  ``parts`` is a list of
    ``(n_rows, 'formula')`` in row order; all formulas share ``params`` and ``xnames``:

        piecewise([(4, 'exp(a + x*b)'), (1, 'b/a')], ['a', 'b'])
    """
    params = list(params)
    if len(params) > TAPE_MAX_PARAM:
        raise ValueError('tape models support at most %d parameters' % TAPE_MAX_PARAM)
    consts, programs, texts = [], [], []
    for n_rows, text in parts:
        if int(n_rows) < 0:
            raise ValueError('piecewise: negative row count')
        c = _Compiler(params, list(xnames))
        c.consts = consts
        c.visit(ast.parse(text.strip(), mode='eval').body)
        if c.maxdepth > TAPE_MAX_STACK:
            raise ValueError('expression needs a stack deeper than %d' % TAPE_MAX_STACK)
        programs.append((int(n_rows), np.asarray(c.code, np.int32)))
        texts.append('%d: %s' % (n_rows, text))
    if not programs:
        raise ValueError('piecewise: no parts')
    tape = np.concatenate([code for _, code in programs])
    return Model(MODEL_TAPE, len(params), len(xnames), tape=tape, consts=consts, text='{' + '; '.join(texts) + '}',
                 programs=programs)
