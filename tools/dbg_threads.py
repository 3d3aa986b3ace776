"""Developer tool (GPU): which pair of job kinds of tests/test_gpu_threads.py disturbs which when run on two host threads?"""
import sys
import threading
import numpy as np
sys.path.insert(0, '.')
import lsqfit_amd as amd
from tests import test_gpu_threads as T

kinds = {'nist': lambda i: T._nist_job(amd, ('misra1a', 'thurber', 'mgh09')[i % 3]), 'general': lambda i: T._general_job(amd, 700 + i),
         'batched': lambda i: T._batched_job(amd, 800 + i), 'jit': lambda i: T._jit_job(amd, 50 + int(sys.argv[1]) if len(sys.argv) > 1 else 50, i)}


def same(a, b):
    return all(np.array_equal(np.asarray(a[k]), np.asarray(b[k])) for k in a)


ref = {}
for k, mk in kinds.items():
    jobs = [mk(i) for i in range(3)]
    r1 = [j() for j in jobs]
    r2 = [j() for j in jobs]
    print('serial repeat', k, [same(a, b) for a, b in zip(r1, r2)], flush=True)
    ref[k] = (jobs, r1)
names = list(kinds)
for a in range(len(names)):
    for b in range(a, len(names)):
        ka, kb = names[a], names[b]
        out = {}

        def work(tag, k):
            jobs, _ = ref[k]
            out[tag] = [j() for j in jobs for _ in range(2)]
        ts = [threading.Thread(target=work, args=('A', ka)), threading.Thread(target=work, args=('B', kb))]
        [t.start() for t in ts]
        [t.join() for t in ts]
        ra = [same(x, ref[ka][1][i // 2]) for i, x in enumerate(out['A'])]
        rb = [same(x, ref[kb][1][i // 2]) for i, x in enumerate(out['B'])]
        print('%-8s with %-8s: A %s  B %s' % (ka, kb, ra, rb), flush=True)
        if not all(ra):
            i = ra.index(False)
            x, y = out['A'][i], ref[ka][1][i // 2]
            print('    first difference in A:', {k: (np.max(np.abs(np.asarray(x[k], float) - np.asarray(y[k], float))), ) for k in x}, 'nit', x['nit'], y['nit'])
        if not all(rb):
            i = rb.index(False)
            x, y = out['B'][i], ref[kb][1][i // 2]
            print('    first difference in B:', {k: (np.max(np.abs(np.asarray(x[k], float) - np.asarray(y[k], float))), ) for k in x}, 'nit', x['nit'], y['nit'])
