"""The tracer front end (lsqfit_amd.trace; CPU): a Python fit function called once on tracer arrays becomes the device
tape.  Spec: the four flattening layouts of src/lsqfit/__init__.py:1997-2042 (array / dict parameters x array / dict
outputs), the functions gvar overloads, and loud refusal of what one recording cannot capture."""
import numpy as np
import pytest

import lsqfit_amd as amd
from lsqfit_amd import trace as T
from lsqfit_amd.models import OP
from oracle.dual import Dual
from tests.helpers import load, nist_problem
from tests.nist_lambdas import MODELS

NIST = load('nist.json')
INV = {v: k for k, v in OP.items()}


def run_tape(model, X, p):
    """reference interpreter of the RPN tape (tests only): values of every row"""
    progs = model.programs or [(X.shape[0], model.tape)]
    out, r0 = np.empty(X.shape[0]), 0
    old = np.seterr(all='ignore')
    for n_rows, code in progs:
        x = X[r0:r0 + n_rows]
        st = []
        for ins in code:
            op, arg = INV[int(ins) & 0xff], int(ins) >> 8
            if op == 'CONST':
                st.append(np.full(n_rows, model.consts[arg]))
            elif op == 'X':
                st.append(x[:, arg].copy())
            elif op == 'P':
                st.append(np.full(n_rows, p[arg]))
            elif op in ('ADD', 'SUB', 'MUL', 'DIV', 'POW'):
                b = st.pop()
                a = st.pop()
                st.append({'ADD': a + b, 'SUB': a - b, 'MUL': a * b, 'DIV': a / b, 'POW': a ** b}[op])
            elif op == 'POWI':
                st.append(st.pop() ** arg)
            else:
                f = dict(NEG=np.negative, EXP=np.exp, LOG=np.log, SIN=np.sin, COS=np.cos, ATAN=np.arctan, SQRT=np.sqrt, TAN=np.tan,
                         SINH=np.sinh, COSH=np.cosh, TANH=np.tanh, ASIN=np.arcsin, ACOS=np.arccos, ABS=np.abs)[op]
                st.append(f(st.pop()))
        assert len(st) == 1
        out[r0:r0 + n_rows] = st[0]
        r0 += n_rows
    np.seterr(**old)
    return out


@pytest.mark.parametrize('name', sorted(NIST))
def test_nist_lambda_traces_to_the_formula_tape(name):
    """the user's numpy function and the formula string give THE SAME tape, instruction for instruction: the fits are then
    bit-identical by construction (same compiled code)"""
    pr = nist_problem(name, NIST)
    cols = pr['columns'][1:]
    x = pr['x'] if len(cols) > 1 else pr['x'][cols[0]]
    tr = amd.trace(MODELS[name], x, np.zeros(pr['P']), fold=False)
    ref = amd.expr(pr['expr'], ['b%d' % (i + 1) for i in range(pr['P'])], cols)
    assert tr.model.programs is None
    got = [(INV[int(c) & 0xff], int(c) >> 8) for c in tr.model.tape]
    want = [(INV[int(c) & 0xff], int(c) >> 8) for c in ref.tape]
    # constants are numbered by first use in both front ends; compare them through their values
    def resolve(seq, consts):
        return [(o, float(consts[a])) if o == 'CONST' else (o, a) for o, a in seq]
    assert resolve(got, tr.model.consts) == resolve(want, ref.consts), (name, got, want)
    assert tr.x.shape == (pr['y'].size, len(cols))
    # the default (fold=True): what involves no parameter -- x**2, cos(2 pi x / 12) -- was evaluated by numpy during the
    # recording and is one more predictor column: never a longer tape, the same function
    tf = amd.trace(MODELS[name], x, np.zeros(pr['P']))
    assert tf.model.tape.size <= tr.model.tape.size
    if name in ('enso', 'hahn1', 'kirby2', 'thurber', 'mgh09'):
        assert tf.model.tape.size < tr.model.tape.size and tf.x.shape[1] > len(cols)
    np.testing.assert_allclose(run_tape(tf.model, tf.x, pr['p0']), MODELS[name](x, pr['p0']), rtol=1e-14)
    np.testing.assert_allclose(run_tape(tr.model, tr.x, pr['p0']), MODELS[name](x, pr['p0']), rtol=1e-14)


def test_dictionary_parameters_wide_sum():
    """p['a'], p['E'] (the canonical lsqfit model, examples/y-vs-x.py:58-61): flattened key by key, the sum over states
    unrolled term by term in the layout the formula compiler recognises (a_k at k, E_k at K + k)"""
    x = np.linspace(0.1, 3.0, 13)
    K = 5

    def fcn(x, p):
        return np.sum(p['a'][:, None] * np.exp(-p['E'][:, None] * x[None, :]), axis=0)

    tr = amd.trace(fcn, x, dict(a=np.zeros(K), E=np.zeros(K)))
    assert tr.pkeys == ['a', 'E'] and tr.model.n_param == 2 * K and tr.model.programs is None
    ps = [int(c) >> 8 for c in tr.model.tape if INV[int(c) & 0xff] == 'P']
    assert ps == [v for k in range(K) for v in (k, K + k)]
    rng = np.random.default_rng(1)
    p = dict(a=rng.uniform(0.5, 1.5, K), E=rng.uniform(0.2, 2.0, K))
    np.testing.assert_allclose(run_tape(tr.model, tr.x, tr.pack_params(p)), fcn(x, p), rtol=1e-14)
    back = tr.unpack_params(tr.pack_params(p))
    assert np.array_equal(back['E'], p['E'])


def test_dictionary_output_gives_one_program_per_formula():
    """examples/simple.py: a dictionary-valued function; rows flattened in the order of the DATA's keys"""
    xa = np.array([1., 2., 3., 4.])

    def fcn(x, p):
        return dict(b=p[1] / p[0], a=np.exp(p[0] + x['a'] * p[1]))

    y = dict(a=np.zeros(4), b=0.0)
    tr = amd.trace(fcn, dict(a=xa), np.zeros(2), y=y, merge_small=False)
    assert tr.ykeys == ['a', 'b'] and [n for n, _ in tr.model.programs] == [4, 1]
    ref = amd.piecewise([(4, 'exp(a + x*b)'), (1, 'b/a')], ['a', 'b'])
    for (n1, c1), (n2, c2) in zip(tr.model.programs, ref.programs):
        assert n1 == n2 and [INV[int(c) & 0xff] for c in c1] == [INV[int(c) & 0xff] for c in c2]
    p = np.array([0.3, 0.7])
    np.testing.assert_allclose(run_tape(tr.model, tr.x, p), np.concatenate([np.exp(p[0] + xa * p[1]), [p[1] / p[0]]]), rtol=1e-15)


def test_no_x_closure_constants_indexing_and_every_function():
    t = np.linspace(0.05, 0.9, 11)
    w = np.linspace(1.0, 2.0, 11)

    def fcn(p):            # data = y only: the function closes over its constants (src/lsqfit/__init__.py:2013-2016, x False)
        a, b, c = p
        s = np.sin(a * t) + np.cos(b * t) * np.tan(c * t) + np.arctan(a * t) + np.sqrt(b + t) + np.log(c + t)
        s = s + np.sinh(a * t) - np.cosh(b * t) + np.tanh(c * t) + np.arcsin(a * t / 4) + np.arccos(b * t / 4) + abs(c - t)
        s = s + np.exp(-a * t) * w + (b * t) ** 2 + (c + t) ** 0.5 + 2.0 ** (a * t) + np.square(b) + 1 / c
        return s[::-1]

    tr = amd.trace(fcn, False, np.zeros(3))
    p = np.array([0.7, 1.3, 0.4])
    np.testing.assert_allclose(run_tape(tr.model, tr.x, p), fcn(p), rtol=1e-13)
    assert tr.model.programs is None and tr.x.shape[1] >= 2      # t and w became predictor columns


def test_rows_that_read_different_parameters_are_different_formulas():
    x = np.arange(6.0)
    group = np.array([0, 0, 0, 1, 1, 1])

    def fcn(x, p):
        return p['norm'][group] * np.exp(-p['E'] * x)

    tr = amd.trace(fcn, x, dict(norm=np.zeros(2), E=0.0), merge_small=False)
    assert [n for n, _ in tr.model.programs] == [3, 3]
    p = dict(norm=np.array([2.0, 3.0]), E=0.25)
    np.testing.assert_allclose(run_tape(tr.model, tr.x, tr.pack_params(p)), fcn(x, p), rtol=1e-15)


def test_output_built_element_by_element_is_merged_again():
    x = np.linspace(0, 1, 9)

    def fcn(x, p):
        return [p[0] * np.exp(-p[1] * xi) for xi in x] + [p[0] + p[1]]

    tr = amd.trace(fcn, x, np.zeros(2), merge_small=False)
    assert [n for n, _ in tr.model.programs] == [9, 1]
    p = np.array([1.5, 0.5])
    np.testing.assert_allclose(run_tape(tr.model, tr.x, p), np.array(fcn(x, p), float), rtol=1e-15)


def test_matmul_stack_concatenate():
    A = np.random.default_rng(3).standard_normal((7, 3))

    def fcn(p):
        lin = A @ p[:3]
        return np.concatenate([lin, np.stack([p[3] * p[0], p[3] + 1.0]), np.dot(A[:2], p[:3]) * p[3]])

    tr = amd.trace(fcn, False, np.zeros(4))
    p = np.array([0.3, -1.2, 2.0, 0.7])
    np.testing.assert_allclose(run_tape(tr.model, tr.x, p), fcn(p), rtol=1e-13)


def test_distribution_keys_of_the_prior():
    """a prior under 'log(a)' / 'sqrt(b)' offers p['a'] / p['b'] to the fit function (gvar.BufferDict; tests/test_lsqfit.py:1594-1640)"""
    def fcn(p, N=4):
        return N * [p['a']] + [p['b'] + p['c']]

    tr = amd.trace(fcn, False, {'log(a)': 0.0, 'sqrt(b)': 0.0, 'c': 0.0}, merge_small=False)
    assert [n for n, _ in tr.model.programs] == [4, 1]
    assert [INV[int(c) & 0xff] for c in tr.model.programs[0][1]] == ['P', 'EXP']
    p = np.array([np.log(0.3), 1.5, -0.25])
    np.testing.assert_allclose(run_tape(tr.model, tr.x, p), [0.3] * 4 + [1.5 ** 2 - 0.25], rtol=1e-15)
    vals = tr.unpack_params(p)
    assert vals['a'] == pytest.approx(0.3) and vals['b'] == pytest.approx(2.25) and 'a' in vals and 'zz' not in vals


def test_selection_on_the_data_is_a_piecewise_model():
    """numpy.where / boolean masks on x (not on parameters): rows split into one formula per range"""
    x = np.linspace(0.0, 2.0, 10)

    def fcn(x, p):
        return np.where(x < 1.0, p[0] * x, p[1] + p[2] * np.exp(-x))

    tr = amd.trace(fcn, x, np.zeros(3), merge_small=False)
    assert [n for n, _ in tr.model.programs] == [5, 5]
    p = np.array([0.7, 1.1, 2.0])
    np.testing.assert_allclose(run_tape(tr.model, tr.x, p), fcn(x, p), rtol=1e-15)

    def fcn2(x, p):
        out = p[0] * x
        return np.concatenate([out[x < 0.5], (p[1] * x ** 2)[x >= 0.5]])

    tr2 = amd.trace(fcn2, x, np.zeros(2), merge_small=False)
    np.testing.assert_allclose(run_tape(tr2.model, tr2.x, p[:2]), fcn2(x, p[:2]), rtol=1e-15)


def test_arctan2_with_a_parameter_dependent_abscissa_is_refused():
    """the half-angle form for x > 0 is NaN on / inaccurate near the negative real axis: which form a row needs is the sign of x,
    control flow when x depends on the parameters (round-5 advisor finding: it was recorded as 2 atan(y / (r + x)) silently)"""
    with pytest.raises(amd.TraceError, match='arctan2'):
        amd.trace(lambda x, p: np.arctan2(x, p[0]), np.linspace(0, 1, 5), np.ones(2))
    with pytest.raises(amd.TraceError):
        amd.trace(lambda x, p: np.average(x * p[0], weights=p[1] * x), np.linspace(0, 1, 5), np.ones(2))


@pytest.mark.parametrize('bad', [
    lambda x, p: p[0] * x if p[0] > 0 else p[1] * x,
    lambda x, p: np.where(p[0] * x > 1, p[0], p[1]),
    lambda x, p: np.maximum(p[0] * x, 0.0),
    lambda x, p: __import__('math').exp(p[0]) * x,
    lambda x, p: x[int(p[0])] * p[1],
])
def test_data_dependent_control_flow_is_refused(bad):
    with pytest.raises(amd.TraceError):
        amd.trace(bad, np.linspace(0, 1, 5), np.ones(2))


def test_unsupported_numpy_function_is_refused_by_name():
    with pytest.raises(amd.TraceError, match='cbrt'):
        amd.trace(lambda x, p: np.cbrt(p[0] * x), np.linspace(0, 1, 5), np.ones(2))
    with pytest.raises(amd.TraceError, match='linalg.inv|numpy.inv'):
        amd.trace(lambda x, p: np.linalg.inv(np.outer(p, p))[0, 0] * x, np.linspace(0, 1, 5), np.ones(2))


_XV = np.linspace(0.2, 1.7, 6)
_AV = np.random.default_rng(11).uniform(0.5, 1.5, (6, 4))
VOCABULARY = {
    'prod': lambda x, p: np.prod(p[:3]) * x + p.prod(),
    'prod axis': lambda x, p: np.prod(np.outer(x, p[:2]) + 1.0, axis=1),
    'multiply.reduce': lambda x, p: np.multiply.reduce(p[1:3]) * x,
    'cumprod': lambda x, p: np.cumprod(p)[-1] * x + np.cumprod(p)[1],
    'norm': lambda x, p: np.linalg.norm(p) * x + np.linalg.norm(np.outer(x, p), axis=1),
    'var std': lambda x, p: np.var(p) * x + np.std(p * 2.0, ddof=1) + (p * x[0]).var(),
    'tensordot': lambda x, p: np.tensordot(_AV, p, 1) + np.tensordot(p[:2], np.vstack([x, x ** 2]), axes=([0], [0])),
    'einsum': lambda x, p: np.einsum('ij,j->i', _AV, p) + np.einsum('i,j->j', p[:2], x) + np.einsum('i,i', p, p),
    'einsum matrix': lambda x, p: np.einsum('ik,kj->ji', np.outer(x, p[:2]), np.outer(p[2:], x[:3])).ravel(),
    'inner vdot': lambda x, p: np.inner(_AV, p) + np.vdot(p, p[::-1]),
    'trace diagonal diag': lambda x, p: np.trace(np.outer(p, p)) * x + np.diagonal(np.outer(p, x[:4]))[1] + np.diag(p * 2.0)[2, 2],
    'triu tril': lambda x, p: (np.triu(np.outer(p, p)) - np.tril(np.outer(p, p), -1)).sum(axis=0)[[0, 1, 2, 3, 0, 1]] * x,
    'tile repeat roll': lambda x, p: np.tile(p[:2], 3) * x + np.repeat(p[:3], 2) + np.roll(np.append(p, p[:2] * x[:2]), 2),
    'split delete': lambda x, p: np.split(p * 1.0, 2)[1][0] * x + np.delete(p, 1)[2],
    'column_stack': lambda x, p: np.column_stack([p[0] * x, p[1] * x ** 2, np.ones_like(x) * p[2]]).ravel(),
    'like': lambda x, p: np.zeros_like(p)[0] + np.ones_like(p * x[:4]).sum() * p[0] * x + np.full_like(p, 2.5)[1] * p[3],
    # arctan2(y, x): the branch follows the sign of x -- data here: all positive, all negative, mixed (a selection by row), incl.
    # points ON the negative real axis (y = 0 at x < 0 is the one place no half-angle form reaches: pi there needs y != 0)
    'arctan2 hypot': lambda x, p: np.arctan2(p[0] * x, x + 0.5) + np.hypot(p[2], x) + np.arctan2(p[3] * x, -x) + np.arctan2(p[1] * x, x - 0.9)
                                  + np.arctan2(p[1] + x, np.array(-3.0)),
    'expm1 log1p': lambda x, p: np.expm1(1e-9 * p[0] * x) * 1e9 + np.log1p(p[1] * x) + np.expm1(p[2] * x),
    'average axis 0': lambda x, p: np.average(np.outer(p, x), axis=0, weights=[1.0, 2.0, 3.0, 4.0]) + np.average(np.outer(p, p), weights=np.outer(np.ones(4), [1., 2., 3., 4.])),
    'sum of many selections': lambda x, p: np.sum(np.where(x[:, None] < np.linspace(0.1, 1.8, 70)[None, :], p[0] * x[:, None], p[1] + 0 * x[:, None]), axis=1),
    'subtract.outer': lambda x, p: np.exp(-np.subtract.outer(x, p[:2]) ** 2).sum(axis=1),
    'average': lambda x, p: np.average(np.outer(x, p), axis=1, weights=[1.0, 2.0, 3.0, 4.0]) + np.average(p),
    'methods': lambda x, p: (p * 1.0).copy().astype(float).swapaxes(0, 0).take([1, 2]).repeat(3) * x + np.outer(p, p).trace(),
}


@pytest.mark.parametrize('name', sorted(VOCABULARY))
def test_numpy_vocabulary_of_the_tracer(name):
    """the rest of the numpy functions a fit function is likely to be written with: recorded, then the tape against numpy itself"""
    fcn = VOCABULARY[name]
    tr = amd.trace(fcn, _XV, np.zeros(4))
    p = np.array([0.8, 1.3, 0.6, 1.1])
    ref = np.asarray(fcn(_XV, p), float).ravel()
    np.testing.assert_allclose(run_tape(tr.model, tr.x, p), ref, rtol=2e-13, atol=1e-14)
    q = np.array([1.4, 0.7, 0.9, 0.55])          # (a second point: nothing parameter-dependent was folded into a constant)
    np.testing.assert_allclose(run_tape(tr.model, tr.x, q), np.asarray(fcn(_XV, q), float).ravel(), rtol=2e-13, atol=1e-14)


@pytest.mark.parametrize('bad', [
    lambda x, p: p.max() * x,
    lambda x, p: np.sort(p)[0] * x,
    lambda x, p: np.clip(p[0] * x, 0, 1),
    lambda x, p: np.round(p[0]) * x,
    lambda x, p: np.sign(p[0]) * x,
    lambda x, p: np.interp(p[0], x, x) * x,
    lambda x, p: np.median(p) * x,
    lambda x, p: p.astype(int)[0] * x,
])
def test_selections_and_rounding_of_parameters_are_refused(bad):
    with pytest.raises(amd.TraceError, match='parameter-dependent'):
        amd.trace(bad, np.linspace(0, 1, 5), np.ones(2))


def test_residual_function_as_lsqfit_hands_it_over():
    """``chiv`` called on an object array of overloaded numbers (src/lsqfit/_utilities.pyx:65-94 with mixed=True): the data
    part weighted row by row, one correlated block multiplied by its weight matrix, the prior rows appended"""
    x = np.linspace(0.2, 2.0, 8)
    y = 1.7 * np.exp(-0.6 * x)
    rng = np.random.default_rng(5)
    wd = rng.uniform(5, 9, 5)
    Wb = rng.standard_normal((3, 3)) + 4 * np.eye(3)
    pm, pw = np.array([1.0, 1.0]), np.array([2.0, 3.0])
    mean = np.concatenate([y, pm])
    iw0 = np.concatenate([np.arange(5), [8, 9]])
    w0 = np.concatenate([wd, pw])
    iw1 = np.arange(5, 8)

    def flat_fcn(p):
        return (p[0] * np.exp(-p[1] * x)).flat

    def chiv(p, mixed=False):
        delta = np.concatenate((flat_fcn(p), p)) - mean
        ans = np.zeros(10, object if mixed else float)
        ans[:7] = np.multiply(memoryview(w0), delta[iw0])
        ans[7:] = np.dot(Wb, delta[iw1])
        return ans

    tr = amd.trace_residual(chiv, 2)
    assert tr.n_rows == 10
    for p in (np.array([1.7, 0.6]), np.array([0.9, 1.4])):
        np.testing.assert_allclose(run_tape(tr.model, tr.x, p), chiv(p), rtol=1e-13, atol=1e-14)


@pytest.mark.parametrize('seed', range(24))
def test_random_formulas_trace_to_the_string_compilers_tape(seed):
    """the formula fuzzer of tests/test_gpu_jit_fuzz.py (wide sums with contiguous / scattered / interleaved parameter indices,
    shared parameters, sums inside products and functions, unread parameters) written as a Python function of numpy calls:
    recorded operation by operation (fold=False) it is the string compiler's tape, instruction for instruction -- and with
    parameter-free arithmetic folded (the default) still the same function"""
    from tests.test_gpu_jit_fuzz import random_formula
    rng = np.random.default_rng(1000 + seed)
    text, names = random_formula(rng)
    code = compile(text, '<fuzz>', 'eval')
    funcs = dict(exp=np.exp, cos=np.cos, sin=np.sin, sqrt=np.sqrt, log=np.log, arctan=np.arctan)

    def fcn(x, p):
        ns = dict(funcs)
        ns['x'] = x
        for i, n in enumerate(names):
            ns[n] = p[i]
        return eval(code, {'__builtins__': {}}, ns)

    x = np.sort(rng.uniform(0.05, 2.0, 40))
    P = len(names)
    tr = amd.trace(fcn, x, np.zeros(P), fold=False)
    ref = amd.expr(text, names)

    def resolve(model):
        return [(INV[int(c) & 0xff], float(model.consts[int(c) >> 8])) if INV[int(c) & 0xff] == 'CONST' else (INV[int(c) & 0xff], int(c) >> 8)
                for c in model.tape]
    assert tr.model.programs is None and resolve(tr.model) == resolve(ref)
    p = rng.uniform(0.3, 1.2, P)
    tf = amd.trace(fcn, x, np.zeros(P))
    assert tf.model.tape.size <= tr.model.tape.size
    np.testing.assert_allclose(run_tape(tf.model, tf.x, p), fcn(x, p), rtol=1e-12)


def test_parameter_selected_row_by_row_becomes_indicator_columns():
    """errors in variables (examples/x-err.py:40-43: ``x = p['x']``, row i reads its own x_i) and interleaved groups
    (``p['norm'][group]``): one program per contiguous run of rows while there are few of them, ONE formula with 0 / 1 predictor
    columns doing the selection when there would be dozens"""
    def fcn(p):
        b0, b1, b2, b3 = p['b']
        return b0 / ((1. + np.exp(b1 - b2 * p['x'])) ** (1. / b3))

    for n in (15, 600):      # short runs (one row each): ONE formula whatever their number (a formula is a run-time compilation)
        pp = dict(b=np.array([1.0, 0.5, 2.0, 1.5]), x=np.linspace(1, 2, n))
        tr = amd.trace(fcn, False, pp, merge_small=False)
        assert tr.model.programs is None
        np.testing.assert_allclose(run_tape(tr.model, tr.x, tr.pack_params(pp)), fcn(pp), rtol=1e-15)
        assert tr.x.shape == (n, n) and set(np.unique(tr.x)) == {0.0, 1.0} and len(tr.model.tape) < 4 * n + 40
    # few LONG runs (three groups of 100 contiguous rows): one program per run, no indicator columns
    xs = np.linspace(0, 1, 300)
    g3 = np.repeat(np.arange(3), 100)
    pp = dict(norm=np.array([2.0, 3.0, 4.0]), E=0.25)
    tr = amd.trace(lambda x, p: p['norm'][g3] * np.exp(-p['E'] * x), xs, pp, merge_small=False)
    assert len(tr.model.programs) == 3 and tr.x.shape == (300, 1)
    np.testing.assert_allclose(run_tape(tr.model, tr.x, tr.pack_params(pp)), pp['norm'][g3] * np.exp(-0.25 * xs), rtol=1e-15)
    x = np.linspace(0, 1, 1000)
    group = np.arange(1000) % 2

    def f2(x, p):
        return p['norm'][group] * np.exp(-p['E'] * x)
    pp = dict(norm=np.array([2.0, 3.0]), E=0.25)
    tr = amd.trace(f2, x, pp, merge_small=False)
    assert tr.model.programs is None and tr.x.shape == (1000, 3)
    np.testing.assert_allclose(run_tape(tr.model, tr.x, tr.pack_params(pp)), f2(x, pp), rtol=1e-15)


def test_errors_raised_in_the_fit_function_surface_unchanged():
    """tests/test_lsqfit.py:1684-1698 (test_multifit_exceptions): the reference stashes an exception raised inside the fit
    function and re-raises it after the driver returns; here the function runs once, at recording time, and the exception
    leaves amd.trace as it was raised"""
    def length_mismatch(x, p):
        raise ValueError('operands could not be broadcast together')

    with pytest.raises(ValueError, match='broadcast'):
        amd.trace(length_mismatch, np.arange(3.0), np.ones(2))
    with pytest.raises(ZeroDivisionError):
        amd.trace(lambda x, p: (1 // 0) * p[0] * x, np.arange(3.0), np.ones(2))



def test_small_fits_with_several_formulas_are_recorded_as_one():
    """A small fit whose rows follow several formulas (a dictionary-valued fit function, examples/simple.py: one formula per key)
    is recorded as ONE formula  sum_k [row in k] * f_k  (trace.merge_small_programs): every row evaluates every formula -- on a
    copy of the first row of the formula's own rows where it does not belong -- and keeps one.  Same values BIT FOR BIT as the
    separate formulas (0 * finite is an exact zero); one tape = one run-time compilation and the one-launch fit route.  Larger
    fits keep their formulas apart."""
    xa = np.array([0.1, 1.0, 0.1, 0.5])

    def fcn(x, p):
        return dict(data1=np.exp(p['a'] + x['data1'] * p['b']), data2=np.exp(p['a'] + x['data2'] * p['b']) / (1 + x['data2']),
                    ratio=p['b'] / p['a'], logs=np.log(p['a'] * x['data1'][:1]))
    x = dict(data1=xa[:2], data2=xa[2:])
    p0 = dict(a=0.5, b=0.5)
    sep = amd.trace(fcn, x, p0, merge_small=False)
    one = amd.trace(fcn, x, p0)
    assert len(sep.model.programs) >= 3 and one.model.programs is None
    for p in (np.array([0.5, 0.5]), np.array([0.25, 1.7]), np.array([3.0, -0.4])):
        a, b = run_tape(sep.model, sep.x, p), run_tape(one.model, one.x, p)
        assert np.array_equal(a, b), (a, b)
    # a formula that is singular for the predictor values of ANOTHER formula's rows never sees them: 1 / (x - 1) on rows with
    # x = 2, 3 beside rows whose x is exactly 1 under a different formula
    xs = np.array([2.0, 3.0, 1.0, 1.0])

    def piece(x, p):
        return np.where(np.arange(4) < 2, p[0] / (x - 1.0), p[1] * x)
    sep = amd.trace(piece, xs, np.zeros(2), merge_small=False)
    one = amd.trace(piece, xs, np.zeros(2))
    assert len(sep.model.programs) == 2 and one.model.programs is None
    q = np.array([0.7, 1.3])
    assert np.array_equal(run_tape(one.model, one.x, q), run_tape(sep.model, sep.x, q)) and np.all(np.isfinite(run_tape(one.model, one.x, q)))
    # not for larger fits
    big = np.linspace(0, 1, 3000)
    g = (big > 0.5).astype(int)
    tr = amd.trace(lambda x, p: np.where(x > 0.5, p[0] * x, np.exp(p[1] * x)), big, np.zeros(2))
    assert len(tr.model.programs) == 2
