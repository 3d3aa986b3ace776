"""Developer tool (GPU): stress the chained back substitution -- many solves in a row, with and without
another stream hammering the memory system, every result compared bit for bit with the first one."""
import sys
import time
import numpy as np
sys.path.insert(0, '.')
import torch
import lsqfit_amd as amd
from lsqfit_amd import synth

for N, P, reps in ((4096, 4096, 3000), (1024, 512, 6000), (512, 256, 6000)):
    d = synth.make_cosmix(N=N, P=P, seed=11, block=0, prior_corr=False)
    pr = amd.DeviceProblem(d['model'], d['x'], amd.Whitening(d['ymean'], d['yerr'], *d['prior']))
    pr.normal(d['p0'])
    diag = np.sqrt(np.diag(pr.get_jtj()))
    first = pr.solve_damped(1e-2, diag)
    side = torch.cuda.Stream()
    big = torch.empty(1 << 28, dtype=torch.float32, device='cuda')      # 1 GiB
    t0 = time.perf_counter()
    bad = 0
    for i in range(reps):
        if i % 3 == 0:                      # uneven background load on another stream
            with torch.cuda.stream(side):
                big.add_(1.0)
        v = pr.solve_damped(1e-2, diag)
        if not np.array_equal(v, first):
            bad += 1
    torch.cuda.synchronize()
    print('P=%d: %d solves, %d differ, %.3f ms each' % (P, reps, bad, (time.perf_counter() - t0) / reps * 1e3))
    assert bad == 0
    pr.close()
    del big
