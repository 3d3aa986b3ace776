"""Developer experiment (GPU): how the work-list J^T J launch is actually scheduled.  Needs the variant build
  tools/build_variant.sh syrkstamps gemm_tn_f64.hip -DLSQAMD_SYRK_STAMPS   (per-workgroup start / end stamps, XCC_ID, HW_ID)
usage: LSQAMD_LIBPATH=lsqfit_amd/build/libsyrkstamps.so python3 tools/syrk_schedule.py [N] [P]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
import lsqfit_amd as amd
from lsqfit_amd import _lib, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
P = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
lib = _lib.load()
d = synth.make_cosmix(N=N, P=P, seed=20263, block=256, prior_corr=True)
wh = amd.Whitening(d['ymean'], d['yerr'], *d['prior'])
pr = amd.DeviceProblem(d['model'], d['x'], wh)
nmax = 1 << 16
buf = torch.zeros(5 * nmax, dtype=torch.int64, device='cuda')
dbg = C.CDLL(os.environ['LSQAMD_LIBPATH'])
assert dbg.lsqamd_debug_set_syrk_stamps(C.c_void_p(buf.data_ptr())) == 0
p = d['p0'].copy()
for _ in range(3):
    buf.zero_()
    pr.normal(p)
torch.cuda.synchronize()
s = buf.cpu().numpy().astype(np.uint64).reshape(-1, 5)
s = s[s[:, 1] > 0]
t0, t1 = s[:, 0].astype(np.float64) / 100.0, s[:, 1].astype(np.float64) / 100.0     # us
base = t0.min()
t0 -= base
t1 -= base
xcc = (s[:, 2] & 7).astype(int)
hw = s[:, 3].astype(np.uint32)
cu = (((hw >> 13) & 7) * 16 + ((hw >> 12) & 1) * 16 * 8 + ((hw >> 8) & 15)).astype(int)      # (se, sh, cu) -> one number
w = s[:, 4].astype(np.int64)
n = len(s)
dur = t1 - t0
bid = np.arange(n)
print('%d workgroups, launch spans %.1f us; durations: median %.1f, 5%% %.1f, 95%% %.1f us' % (n, t1.max(), np.median(dur), np.percentile(dur, 5), np.percentile(dur, 95)))
print('blockIdx %% 8 == XCC_ID for %.1f %% of the workgroups' % (100.0 * np.mean((bid % 8) == xcc)))
for x in range(8):
    m = xcc == x
    cus = len(set(cu[m].tolist()))
    order = np.argsort(t0[m])
    st, en = t0[m][order], t1[m][order]
    # busy slots over time
    ev = np.concatenate([np.stack([st, np.ones_like(st)], 1), np.stack([en, -np.ones_like(en)], 1)])
    ev = ev[np.argsort(ev[:, 0], kind='stable')]
    run = np.cumsum(ev[:, 1])
    tt = ev[:, 0]
    area = np.sum(run[:-1] * np.diff(tt))
    full = run.max()
    last = en.max()
    # time at which the number of running workgroups drops below half of the maximum for good
    below = np.nonzero(run >= full / 2)[0]
    t_half = tt[below[-1] + 1] if below[-1] + 1 < len(tt) else last
    print('XCC %d: %4d workgroups on %2d CUs, max %3d at once, first %.1f last end %.1f us, slot-time used %.1f %% of max x span, '
          'below half occupancy from %.1f us (tail %.1f us)' % (x, m.sum(), cus, full, st.min(), last, 100.0 * area / (full * last), t_half, last - t_half))
# how many rounds does a slot see?  order of starts inside one XCC vs list order
m = xcc == 0
print('XCC 0: list indices of the first 8 and last 8 workgroups to start:', w[m][np.argsort(t0[m])][:8].tolist(), w[m][np.argsort(t0[m])][-8:].tolist())
late = np.argsort(t1)[-16:]
print('last 16 to finish: list index, XCC, start, duration:', [(int(w[i]), int(xcc[i]), round(float(t0[i]), 1), round(float(dur[i]), 1)) for i in late])
