"""-m gpu: the formula compiler (lsqfit_amd/csrc/jit.hip) on randomly generated formulas -- wide sums with contiguous
and scattered parameter indices, parameters shared by all terms, sums inside products / functions, several sums in
one formula, parameters a formula never reads, formulas without sums -- values and EVERY derivative against the
oracle's forward-mode dual numbers (oracle/dual.py: what gvar.valder gives the reference, src/lsqfit/_gsl.pyx:742-760)."""
import ctypes as C

import numpy as np
import pytest

from oracle import dual

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def amd():
    import lsqfit_amd
    from lsqfit_amd import _lib
    _lib.load()
    return lsqfit_amd


TERMS = ['{a}*cos({b}*x)', '{a}*exp(-{b}*x)', '{a}/(1 + {b}*x**2)', '{a}*sin({b}*x + {s})', '{a}*x*exp(-({b}*x)**2)',
         'sqrt(1 + ({a}*x)**2)*{b}', 'log(1 + {a}**2 + {b}**2*x**2)', '{a}*arctan({b}*x)/{s}',
         '{a}*cos({k}*x) + 0*{b}', '{a}*exp(-{b}*(x - {k})**2)']        # {k}: a literal that differs from term to term
OUTER = ['{S}', '{c} + {S}', 'exp(-{c}*x)*({S})', '({S})/(1 + {c}**2)', '({S})*({T}) + {c}', '{c}*log(1 + ({S})**2)',
         '({S}) - ({T})', '-({S}) + {c}*x', 'sqrt(1 + ({S})**2) + {T}']


def random_formula(rng):
    """-> (text, names): 1-2 sums of 3..70 look-alike terms (16 and more: one loop over a wave's lanes; fewer: unrolled per lane) + an outer expression + possibly unread parameters"""
    names, sums = [], []
    nsum = int(rng.integers(1, 3))
    shared = 'sh%d' % int(rng.integers(0, 100))
    for si in range(nsum):
        K = int(rng.choice([3, 5, 16, 17, 64, 70]))
        tmpl = TERMS[int(rng.integers(0, len(TERMS)))]
        a = ['a%d_%d' % (si, k) for k in range(K)]
        b = ['b%d_%d' % (si, k) for k in range(K)]
        if rng.random() < 0.4:                      # scattered parameter order -> index tables instead of affine indices
            order = rng.permutation(K)
            a = [a[i] for i in order]
        # declaration order decides the parameter indices: families contiguous (affine) or interleaved (stride 2)
        if rng.random() < 0.5:
            names += a + b
        else:
            names += [v for pair in zip(a, b) for v in pair]
        sign = ['+', '-'][int(rng.integers(0, 2))] if rng.random() < 0.3 else '+'
        sums.append((' %s ' % sign).join('(' + tmpl.format(a=a[k], b=b[k], s=shared, k=repr(0.25 * (k + 1))) + ')' for k in range(K)))
    uses_shared = any(shared in s for s in sums)
    outer = OUTER[int(rng.integers(0, len(OUTER)))]
    if '{T}' in outer and nsum == 1:
        outer = outer.replace('{T}', '{c}*x')
    text = outer.format(S=sums[0], T=sums[-1], c='cc')
    if uses_shared:
        names.append(shared)
    if 'cc' in text:
        names.append('cc')
    if rng.random() < 0.5:
        names.insert(int(rng.integers(0, len(names) + 1)), 'unread_p')       # a parameter the formula never reads
    return text, names


def oracle_values(text, names, x, p):
    ns = dict(dual.NAMESPACE)
    ns['x'] = x
    d = dual.Dual.seed(p)
    for i, n in enumerate(names):
        ns[n] = d[i]
    out = eval(compile(text, '<fuzz>', 'eval'), {'__builtins__': {}}, ns)
    return out.val, out.der


@pytest.mark.parametrize('seed', range(32))
def test_compiled_formula_matches_dual_numbers(amd, seed):
    rng = np.random.default_rng(1000 + seed)
    text, names = random_formula(rng)
    P, N = len(names), 300
    model = amd.expr(text, names)
    x = np.sort(rng.uniform(0.05, 2.0, N))
    p = rng.uniform(0.3, 1.5, P)
    ysd = rng.uniform(0.5, 2.0, N)
    f0, J0 = oracle_values(text, names, x, p)
    ym = f0 + 0.1 * rng.standard_normal(N)
    wh = amd.Whitening(ym, ysd)                                    # no prior: rows of J are the model rows alone
    pr = amd.DeviceProblem(model, x, wh)
    compiled = bool(pr.lib.lsqamd_debug_flags(pr.h) & 8)
    chi2 = pr.normal(p)
    J = pr.get_J_data()
    f = pr.get_f_data()
    want_f = (f0 - ym) / ysd
    want_J = J0 / ysd[:, None]
    scale = np.abs(want_J).max(axis=0) + 1e-300
    assert np.max(np.abs(J - want_J) / scale) < 1e-11, (text[:120], compiled)
    assert np.max(np.abs(f - want_f)) < 1e-11 * max(1.0, np.abs(want_f).max())
    assert chi2 == pytest.approx(float(want_f @ want_f), rel=1e-12)
    assert abs(pr.chi2(p) - chi2) < 1e-11 * max(1.0, chi2)         # the residual kernel agrees with the Jacobian kernel's column
    if 'unread_p' in names:
        assert np.all(J[:, names.index('unread_p')] == 0.0)
    # most of these formulas are inside what the generator handles; the ones that are not must still be right (interpreter)
    pr.close()
    test_compiled_formula_matches_dual_numbers.compiled = getattr(test_compiled_formula_matches_dual_numbers, 'compiled', 0) + compiled


def test_most_fuzz_formulas_ran_compiled():
    assert getattr(test_compiled_formula_matches_dual_numbers, 'compiled', 0) >= 24
