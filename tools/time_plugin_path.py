"""Developer tool (GPU): fits made through the plugin call alone -- FITTERS['mi355x_lm'](p0, nf, chiv) with chiv built as lsqfit
builds it (tests/lsqfit_protocol.py) -- which route do they take, how long?"""
import sys
import time
import numpy as np
sys.path.insert(0, '.')
import lsqfit_amd
from lsqfit_amd import fitter
from tests import lsqfit_protocol as lp
from tests.helpers import load
KAT, NIST = load('kat.json'), load('nist.json')
cases = [('simple.py', lp.simple_example()), ('p-corr.py', lp.p_corr_example(KAT['p_corr'])), ('x-err.py', lp.x_err_example(KAT['x_err'])),
         ('y-vs-x.py nexp=3', lp.y_vs_x_example(KAT['y_vs_x'], 3)), ('nist misra1a', lp.nist_example('misra1a', NIST)[0]),
         ('nist enso', lp.nist_example('enso', NIST)[0]), ('nist hahn1', lp.nist_example('hahn1', NIST)[0])]
for name, ex in cases:
    p0, nf, chiv, pdf = lp.fitter_call(**ex)
    out = []
    for rep in range(5):
        t0 = time.perf_counter()
        fit = fitter.mi355x_lm(p0, nf, chiv, tol=1e-8, maxit=1000)
        wall = 1e3 * (time.perf_counter() - t0)
        s = fit.summary
        pr = fit.problem
        out.append((wall, s.t_run_ms if s else float('nan'), fit.nit, bool(pr.lib.lsqamd_debug_flags(pr.h) & 32), pr.model.programs is None))
    print('%-18s P %2d nf %3d  nit %3d  ONE formula %-5s one launch %-5s device run %.3f ms  whole plugin call %.2f ms (first %.0f ms)'
          % (name, p0.size, nf, out[-1][2], out[-1][4], out[-1][3], min(o[1] for o in out[1:]), sorted(o[0] for o in out[1:])[2], out[0][0]))
