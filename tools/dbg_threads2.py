"""Developer tool (GPU): general-path fits on one thread, lockstep batches on another -- how often does a batch differ from its
serial result, with / without the captured graphs of either side?  usage: dbg_threads2.py [reps]"""
import os
import sys
import threading
import numpy as np
sys.path.insert(0, '.')
import lsqfit_amd as amd
from lsqfit_amd import synth
from tests import test_gpu_threads as T

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10


def batched_job(seed, use_graph):
    d = synth.make_cosmix(N=512, P=32, seed=seed, block=0, prior_corr=False)
    pm, ps = d['prior']
    B = 6
    psb = np.tile(ps, (B, 1))
    psb[:, :16] = (0.1 * 10 ** (2.0 * np.arange(B) / (B - 1)))[:, None]
    pmb = np.tile(pm, (B, 1))

    def run():
        bf = amd.BatchedFits(d['model'], d['x'], d['ymean'], d['yerr'], pmb, psb)
        out = bf.run(p0=np.tile(d['p0'], (B, 1)), use_graph=use_graph)
        res = dict(p=out['pmean'].copy(), chi2=out['chi2'].copy(), nit=out['nit'].copy())
        bf.close()
        return res
    return run


def same(a, b):
    return all(np.array_equal(np.asarray(a[k]), np.asarray(b[k])) for k in a)


for gen_graph in ('1', '0'):
    for bat_graph in (True, False):
        os.environ['LSQAMD_STEP_GRAPH'] = gen_graph
        gen = [T._general_job(amd, 700 + i) for i in range(3)]
        bat = [batched_job(800 + i, bat_graph) for i in range(3)]
        gref = [j() for j in gen]
        bref = [j() for j in bat]
        bad_g = bad_b = 0
        detail = []
        for rep in range(reps):
            out = {}

            def wg():
                try:
                    out['g'] = [j() for j in gen]
                except Exception as e:
                    print('GENERAL thread failed:', e, flush=True)
                    out['g'] = []

            def wb():
                try:
                    out['b'] = [j() for j in bat]
                except Exception as e:
                    print('BATCHED thread failed:', e, flush=True)
                    out['b'] = []
            ts = [threading.Thread(target=wg), threading.Thread(target=wb)]
            [t.start() for t in ts]
            [t.join() for t in ts]
            bad_g += sum(not same(a, b) for a, b in zip(out['g'], gref))
            for a, b in zip(out['b'], bref):
                if not same(a, b):
                    bad_b += 1
                    detail.append((list(a['nit']), list(b['nit'])))
        print('general graphs %s, batched graphs %s: %d of %d general fits differ, %d of %d batches differ %s'
              % (gen_graph, bat_graph, bad_g, 3 * reps, bad_b, 3 * reps, detail[:3]), flush=True)
