"""Developer tool (GPU): small fits whose rows follow several formulas (dictionary-valued fit functions), recorded as one
formula (trace.merge_small_programs: the one-launch route) against one program per formula (the general path)."""
import sys
import time
import numpy as np
sys.path.insert(0, '.')
import lsqfit_amd as amd
from lsqfit_amd import trace as _t
tr = sys.modules['lsqfit_amd.trace']
from tests import lsqfit_protocol as lp

rng = np.random.default_rng(5)


def simple_case():
    x = dict(data1=np.array([0.1, 1.0]), data2=np.array([0.1, 0.5]))
    y = dict(data1=np.array([1.376, 2.010]), data2=np.array([1.329, 1.582]), ba=np.array(2.0))
    cov = dict(data1=np.array([[0.0047, 0.01], [0.01, 0.056]]), data2=np.array([[0.0047, 0.0067], [0.0067, 0.0136]]), ba=np.array(0.25))

    def fcn(x, p):
        return dict(data1=np.exp(p['a'] + x['data1'] * p['b']), data2=np.exp(p['a'] + x['data2'] * p['b']), ba=p['b'] / p['a'])
    return dict(data=(x, y, cov), fcn=fcn, prior=(dict(a=0.5, b=0.5), dict(a=0.5, b=0.5)))


def correlators(nkey=3, nt=24, nexp=2):
    t = np.arange(1.0, nt + 1)
    a = rng.uniform(0.4, 1.0, (nkey, nexp))
    E = np.array([0.5, 1.1, 1.9])[:nexp]
    x, y, sd = {}, {}, {}
    for k in range(nkey):
        key = 'c%d' % k
        x[key] = t
        f = sum(a[k, n] * np.exp(-E[n] * t) for n in range(nexp))
        sd[key] = 0.01 * f
        y[key] = f + sd[key] * rng.standard_normal(nt)

    def fcn(x, p):
        return {('c%d' % k): sum(p['a%d' % k][n] * np.exp(-p['E'][n] * x['c%d' % k]) for n in range(nexp)) for k in range(nkey)}
    pm = dict(E=E * 1.05)
    ps = dict(E=np.full(nexp, 0.5))
    for k in range(nkey):
        pm['a%d' % k] = np.full(nexp, 0.7)
        ps['a%d' % k] = np.full(nexp, 0.5)
    return dict(data=(x, y, sd), fcn=fcn, prior=(pm, ps))


for name, case in (('simple.py (7 rows, 2 parameters, 3 formulas)', simple_case()), ('3 correlators x 24 points, 8 parameters', correlators()),
                   ('6 correlators x 32 points, 14 parameters', correlators(6, 32, 2))):
    for merged in (True, False):
        tr.MERGE_MAX_FORMULAS = 8 if merged else 0
        out = []
        for rep in range(6):
            t0 = time.perf_counter()
            fit = amd.nonlinear_fit(**case)
            wall = 1e3 * (time.perf_counter() - t0)
            s = fit.fitter_results.summary
            flags = fit.problem.lib.lsqamd_debug_flags(fit.problem.h)
            out.append((wall, s.t_run_ms, s.nit, bool(flags & 32), fit.chi2, fit.model.programs is None))
        w = sorted(o[0] for o in out[1:])
        print('%-46s %-28s nit %3d  one launch %-5s  device run %.3f ms  whole call %.2f ms (median of 5, first call %.0f ms)  chi2 %.10g'
              % (name, 'ONE formula' if out[-1][5] else 'one program per formula', out[-1][2], out[-1][3], min(o[1] for o in out[1:]), w[len(w) // 2], out[0][0], out[-1][4]))
