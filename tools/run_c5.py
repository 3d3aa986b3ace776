"""Developer tool (GPU): BASELINE config 5 (128 batched fits of 4096 x 512) once, for profiling.
usage: run_c5.py [nograph]   (rocprofv3 --kernel-trace is happier without the hipGraph replay)"""
import sys
import numpy as np
import lsqfit_amd as amd
from lsqfit_amd import synth
d = synth.make_cosmix(N=4096, P=512, seed=20264, block=0, prior_corr=False)
B = 128
pm = np.tile(d['prior'][0], (B, 1))
ps = np.tile(d['prior'][1], (B, 1))
ps[:, :256] = (0.1 * 10 ** (2 * np.arange(B) / (B - 1)))[:, None]
bf = amd.BatchedFits(d['model'], d['x'], d['ymean'], d['yerr'], pm, ps)
for rep in range(2):
    out = bf.run(covariance=False, use_graph='nograph' not in sys.argv)
    print('rounds', out['rounds'], 'device ms', round(out['device_ms'], 1), 'nit', out['nit'].min(), out['nit'].max())
