import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


# Order of the -m gpu files under `-x`: what a red test must never hide comes first.  (1) Parity with the oracle and the
# reference's golden vectors (NIST certified values, printed example outputs, full-size configs C2..C5); (2) the multi-rank
# path; (3) unit checks of single kernels against numpy; (4) route-vs-route equivalence files (one implementation against
# another of this library: a failure there says two routes differ, not that a result is wrong) -- last.
_GPU_ORDER = [
    'test_gpu_parity', 'test_gpu_trace', 'test_gpu_protocol', 'test_gpu_scale', 'test_gpu_whiten', 'test_gpu_trf', 'test_gpu_trs', 'test_gpu_qr', 'test_gpu_points',
    'test_gpu_resample', 'test_gpu_fitp', 'test_gpu_fuzz', 'test_gpu_jit_fuzz', 'test_gpu_midsize', 'test_gpu_batched',
    'test_gpu_tape', 'test_gpu_edge', 'test_gpu_interleaved', 'test_gpu_programs', 'test_gpu_robust', 'test_gpu_cosh', 'test_gpu_one_launch', 'test_gpu_fused_normal',
    'test_gpu_comm', 'test_gpu_comm_multi', 'test_gpu_dist2', 'test_gpu_threads', 'test_gpu_bench_smoke',
    'test_gpu_ops', 'test_gpu_syrk_colsum', 'test_gpu_tri_halves', 'test_gpu_uninit',
    'test_gpu_fused_jacobian', 'test_gpu_stepgraph',
]


def _rank(item):
    name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
    return _GPU_ORDER.index(name) if name in _GPU_ORDER else len(_GPU_ORDER)


def pytest_collection_modifyitems(config, items):
    """Parity evidence first (see _GPU_ORDER); GPU tests never run by accident on a box without a device."""
    items.sort(key=_rank)          # (stable: the order inside a file, and of the CPU files, is kept)
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


def pytest_sessionfinish(session, exitstatus):
    """With the hand-off audit on for the whole run (LSQAMD_VERIFY_HANDOFF=1: after the host has acted on a polled pinned block, the
    library waits for the stream and compares it with the device's own copy), a single differing word fails the session."""
    if not os.environ.get('LSQAMD_VERIFY_HANDOFF'):
        return
    try:
        import ctypes
        from lsqfit_amd import _lib
        if _lib._lib is None:
            return
        st = (ctypes.c_int64 * 3)()
        _lib._lib.lsqamd_handoff_stats(st)
        print('\nhand-offs of this process: %d snapshots polled again, %d served from device memory, %d words differed from the device copy'
              % (st[0], st[1], st[2]))
        if st[2] and session.exitstatus == 0:
            session.exitstatus = 1
    except Exception as e:       # the audit must never hide the suite's own result
        print('hand-off statistics unavailable: %r' % (e,))
