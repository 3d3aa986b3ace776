"""Developer tool (GPU): run the batched engine on a NaN-poisoned workspace to flush out reads of
uninitialised memory."""
import numpy as np
import torch
import lsqfit_amd as amd
from lsqfit_amd import synth

d = synth.make_cosmix(N=4096, P=512, seed=20264, block=0, prior_corr=False)
B = 128
pm = np.tile(d['prior'][0], (B, 1))
ps = np.tile(d['prior'][1], (B, 1))
z = 0.1 * 10 ** (2 * np.arange(B) / (B - 1))
ps[:, :256] = z[:, None]
for poison in (False, True):
    if poison:
        t = torch.full((6 * 1024 ** 3 // 8,), float('nan'), dtype=torch.float64, device='cuda')
        del t
    bf = amd.BatchedFits(d['model'], d['x'], d['ymean'], d['yerr'], pm, ps)
    out = bf.run(p0=d['p0'])
    print('poison', poison, 'status ok', np.all(out['status'] == 0), 'finite logGBF', np.isfinite(out['logGBF']).sum(),
          'chi2 range', out['chi2'].min(), out['chi2'].max(), 'rounds', out['rounds'])
    bf.close()
    del bf
