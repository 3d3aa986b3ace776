"""Developer tool (GPU): Jacobian assembly time of the expression-tape model (reverse-mode AD) at
N = 65536 for a P = 64 and a P = 1024 tape, beside the analytic cosmix kernel on the same function.
Run under rocprofv3 --kernel-trace --stats for the per-kernel numbers in profiles/."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
import lsqfit_amd as amd
from lsqfit_amd import models, synth

N = 65536


def damped(K):
    """exp(-g*x) * sum_k a_k cos(w_k x): a root that is NOT a sum (P = 2 K + 1), built instruction by instruction"""
    t = models.tape_sum('a*cos(w*x)', K)
    OP = models.OP
    code = [OP['P'] | ((2 * K) << 8), OP['NEG'], OP['X'], OP['MUL'], OP['EXP']] + [int(c) for c in t.tape] + [OP['MUL']]
    return models.Model(models.MODEL_TAPE, 2 * K + 1, 1, tape=code, consts=t.consts, text='exp(-g*x)*sum_%d(a*cos(w*x))' % K)


for P in (64, 1024, 1025):
    d = synth.make_cosmix(N=N, P=P - P % 2, seed=5, block=0, prior_corr=False)
    tape = models.tape_sum('a*cos(w*x)', P // 2) if P % 2 == 0 else damped(P // 2)
    if P % 2:     # one parameter more than the synthetic problem: g, prior 0.01 +- 0.01, start 0.01
        d['prior'] = (np.append(d['prior'][0], 0.01), np.append(d['prior'][1], 0.01))
        d['p0'] = np.append(d['p0'], 0.01)
    wh = amd.Whitening(d['ymean'], d['yerr'], *d['prior'])
    for name, model in (('tape', tape), ('cosmix', d['model'])):
        if P % 2 and name == 'cosmix':
            continue
        pr = amd.DeviceProblem(model, d['x'], wh)
        if name == 'tape':
            print('          (tape runs %s)' % ('COMPILED (hiprtc)' if pr.lib.lsqamd_debug_flags(pr.h) & 8 else 'through the interpreter kernels'))
        for rep in range(3):       # untimed: code-object load, first-touch of the workspace (0.160 vs 0.037 ms in one round-3 row)
            pr.normal(d['p0'])
            pr.chi2(d['p0'])
        pr.timing(True)
        pr.timing_reset()
        for rep in range(6):
            pr.normal(d['p0'])
        tm = pr.timings()
        jac_ms = tm['jacobian'][0] / tm['jacobian'][1]
        nbytes = 8.0 * N * (P + 1)
        print('P = %4d  %-6s  Jacobian %.3f ms  (J = %.0f MB: %.2f TB/s of J written)  tape length %d'
              % (P, name, jac_ms, nbytes / 1e6, nbytes / jac_ms / 1e9, 0 if model.tape is None else len(model.tape)))
        pr.timing_reset()
        for rep in range(6):
            pr.chi2(d['p0'])
        tm = pr.timings()
        print('          %-6s  residual %.3f ms' % (name, tm['residual'][0] / tm['residual'][1]))
        pr.close()
