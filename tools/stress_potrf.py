"""Developer tool (GPU): stress the factorisation (pivot-wave diagonal kernel: LDS flags between waves) -- many
runs at several sizes, with another stream hammering the memory system, every factor compared bit for bit
with the first one; then a matrix that is not positive definite (must report, not hang)."""
import ctypes as C
import sys
import time
import numpy as np
import torch
sys.path.insert(0, '.')
from lsqfit_amd import _lib

lib = _lib.load()
side = torch.cuda.Stream()
big = torch.empty(1 << 27, dtype=torch.float32, device='cuda')
main = torch.cuda.Stream()
torch.cuda.set_stream(main)
for n, reps in ((4096, 300), (1024, 1500), (200, 3000), (128, 3000)):
    ld = n + 128 if n % 128 == 0 else (n + 16) // 16 * 16
    rng = np.random.default_rng(n)
    G = rng.standard_normal((n + 64, n))
    A = G.T @ G + 0.1 * np.eye(n)
    Ah = np.zeros((n, ld))
    Ah[:, :n] = np.triu(A)
    Ah[:, n] = rng.standard_normal(n)
    src = torch.from_numpy(Ah).cuda()
    wb = lib.lsqamd_op_potrf_work_bytes(n)
    work = torch.zeros(wb // 8 + 8, dtype=torch.float64, device='cuda')
    info = torch.zeros(4, dtype=torch.int32, device='cuda')
    first = None
    bad = 0
    t0 = time.perf_counter()
    for i in range(reps):
        if i % 2 == 0:
            with torch.cuda.stream(side):
                big.add_(1.0)
        dA = src.clone()
        rc = lib.lsqamd_op_potrf_upper(C.c_void_p(main.cuda_stream), C.c_void_p(dA.data_ptr()), n, ld, n + 1,
                                       C.c_void_p(work.data_ptr()), wb, C.c_void_p(info.data_ptr()))
        assert rc == 0
        main.synchronize()
        res = (dA.cpu().numpy(), work.cpu().numpy().copy()) if i % 10 == 0 else None
        if res is not None:
            if first is None:
                first = res
                U = np.triu(res[0][:, :n])
                assert np.abs(U - np.linalg.cholesky(A).T).max() < 1e-10 * np.abs(U).max()
            elif not (np.array_equal(np.triu(res[0][:, :n + 1]), np.triu(first[0][:, :n + 1])) and np.array_equal(res[1], first[1])):
                bad += 1
        assert int(info[0]) == 0
    torch.cuda.synchronize()
    print('n=%d: %d factorisations, %d compared, %d differ, %.3f ms each' % (n, reps, (reps + 9) // 10, bad, (time.perf_counter() - t0) / reps * 1e3))
    assert bad == 0
    # not positive definite: must come back with info set
    Ah2 = Ah.copy()
    k = n // 2
    Ah2[k, k] = -1.0
    dA = torch.from_numpy(Ah2).cuda()
    lib.lsqamd_op_potrf_upper(C.c_void_p(main.cuda_stream), C.c_void_p(dA.data_ptr()), n, ld, n + 1, C.c_void_p(work.data_ptr()), wb,
                              C.c_void_p(info.data_ptr()))
    main.synchronize()
    print('   not positive definite at row %d: info = %d' % (k, int(info[0])))
    assert int(info[0]) != 0
