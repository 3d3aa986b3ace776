"""Developer tool (GPU): handle churn from four host threads -- 4 x 300 small fits (one-launch and general path alternating, a batch
every 25th), every handle created and destroyed on its thread; results must repeat bit for bit, pools must not grow."""
import sys
import threading
import time
import numpy as np
sys.path.insert(0, '.')
import torch
import lsqfit_amd as amd
from lsqfit_amd import _lib, synth
from tests import test_gpu_threads as T

lib = _lib.load()
jobs = {0: T._nist_job(amd, 'misra1a'), 1: T._general_job(amd, 701), 2: T._nist_job(amd, 'thurber'), 3: T._general_job(amd, 702)}
bat = T._batched_job(amd, 801)
ref = {k: j() for k, j in jobs.items()}
bref = bat()
bad = []
t0 = time.time()
mem0 = torch.cuda.memory_allocated()


def work(t):
    for i in range(300):
        r = jobs[(t + i) % 4]()
        if not all(np.array_equal(np.asarray(r[k]), np.asarray(ref[(t + i) % 4][k])) for k in r):
            bad.append((t, i))
        if i % 25 == 24:
            b = bat()
            if not all(np.array_equal(np.asarray(b[k]), np.asarray(bref[k])) for k in b):
                bad.append((t, i, 'batch'))


ts = [threading.Thread(target=work, args=(t,)) for t in range(4)]
[t.start() for t in ts]
[t.join() for t in ts]
import ctypes as C
st = (C.c_int64 * 3)()
lib.lsqamd_handoff_stats(st)
jc = (C.c_int64 * 3)()
lib.lsqamd_jit_cache_stats(jc)
print('1200 fits + 48 batches on 4 threads in %.1f s; differing results: %s; hand-off stats %s; jit cache (loaded, held, evicted) %s; torch memory %d -> %d bytes'
      % (time.time() - t0, bad, list(st), list(jc), mem0, torch.cuda.memory_allocated()))
