"""Developer tool (GPU): single-fit engine on a NaN-poisoned vs a fresh workspace: results must be
bit-identical (flushes out reads of uninitialised workspace memory)."""
import numpy as np
import torch
import lsqfit_amd as amd
from lsqfit_amd import synth

for shape in ((2048, 256, 256, True), (1000, 130, 100, True), (4096, 512, 0, False), (2048, 256, 2048, True)):
    N, P, block, pc = shape
    d = synth.make_cosmix(N=N, P=P, seed=5, block=block, prior_corr=pc)
    out = []
    for poison in (False, True):
        if poison:
            t = torch.full((3 * 1024 ** 3 // 8,), float('nan'), dtype=torch.float64, device='cuda')
            del t
        else:
            torch.cuda.empty_cache()
        fit = amd.nonlinear_fit(data=(d['x'], d['ymean'], d['yerr']), model=d['model'], prior=d['prior'], p0=d['p_true'])
        D = fit.dp_dinputs(np.ones((1, P)))
        out.append((fit.pmean.copy(), fit.cov.copy(), fit.chi2, fit.nit, fit.logGBF, D.copy(), fit.dchi2(fit.pmean[None, :] * 1.0001)))
        fit.problem.close()
        del fit
    a, b = out
    same = all(np.array_equal(np.asarray(x), np.asarray(y)) for x, y in zip(a, b))
    print(shape, 'bit-identical' if same else 'DIFFERENT', 'nit', a[3], b[3], 'finite', np.isfinite(b[0]).all() and np.isfinite(b[1]).all())
