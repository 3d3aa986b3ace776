"""-m "not gpu": the C-ABI library builds for gfx950, loads, and exports every symbol
include/lsqfit_amd.h declares (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def libpath():
    from lsqfit_amd import build
    return build.build()


def header_symbols():
    text = open(os.path.join(ROOT, 'include', 'lsqfit_amd.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(lsqamdb?_[A-Za-z_0-9]+)\s*\(', text)) - {'lsqamd_reduce_fn'})


def test_every_declared_symbol_is_exported(libpath):
    import torch  # noqa: F401  -- first: the library must bind to the HIP runtime torch ships
    so = ctypes.CDLL(libpath)
    names = header_symbols()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(so, n)]
    assert not missing, missing


def test_binding_covers_the_header(libpath):
    from lsqfit_amd import _lib
    assert sorted(_lib.PROTOTYPES) == header_symbols()
    lib = _lib.load()
    assert lib.lsqamd_abi_version() == _lib.ABI_VERSION


def test_struct_layouts_match_header():
    """ctypes mirrors of lsqamd_config / options / summary (sizes as compiled by gcc)."""
    import subprocess
    import tempfile
    from lsqfit_amd import _lib
    src = '#include <stdio.h>\n#include "lsqfit_amd.h"\nint main(){printf("%zu %zu %zu\\n", sizeof(lsqamd_config), sizeof(lsqamd_options), sizeof(lsqamd_summary));return 0;}\n'
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, 't.c')
        open(c, 'w').write(src)
        exe = os.path.join(td, 't')
        subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), c, '-o', exe])
        sizes = [int(v) for v in subprocess.check_output([exe]).split()]
    assert sizes == [ctypes.sizeof(_lib.Config), ctypes.sizeof(_lib.Options), ctypes.sizeof(_lib.Summary)]


def test_workspace_sizing_and_config_validation(libpath):
    from lsqfit_amd import _lib
    lib = _lib.load()
    cfg = _lib.Config(abi_version=_lib.ABI_VERSION, model=1, n_data=65536, n_param=4096, n_x=1, has_prior=1, prior_dense=1,
                      n_blocks=256, max_block=256, sum_block_sq=256 * 256 * 256, want_jacobian_out=1, n_batch=1)
    nbytes = lib.lsqamd_workspace_bytes(ctypes.byref(cfg))
    assert 4e9 < nbytes < 12e9          # two Jacobian-sized buffers dominate (2 x 2.16 GB)
    cfg.n_param = 4095                  # sum models need an even parameter count
    assert lib.lsqamd_workspace_bytes(ctypes.byref(cfg)) == 0
    cfg.n_param = 4096
    cfg.abi_version = 99
    assert lib.lsqamd_workspace_bytes(ctypes.byref(cfg)) == 0
    h = ctypes.c_void_p()
    cfg.abi_version = _lib.ABI_VERSION
    assert lib.lsqamd_create(ctypes.byref(cfg), None, 0, None, ctypes.byref(h)) == -1   # EINVAL, no abort


def test_query_devices_without_compute(libpath):
    import torch
    from lsqfit_amd import _lib
    lib = _lib.load()
    n, mem = ctypes.c_int32(-1), ctypes.c_int64(-1)
    buf = ctypes.create_string_buffer(64)
    assert lib.lsqamd_query_devices(ctypes.byref(n), 0, buf, 64, ctypes.byref(mem)) == 0
    assert n.value == torch.cuda.device_count()
    if n.value:
        assert buf.value.startswith(b'gfx') and mem.value > 1 << 30
    else:
        assert buf.value == b'' and mem.value == 0
    assert lib.lsqamd_query_devices(None, 0, None, 0, None) == -1


def test_exceptions_never_cross_the_abi(libpath):
    """SURVEY.md 8(b) "never throw across the ABI" (the reference's callbacks are noexcept and stash errors,
    src/lsqfit/_gsl.pyx:726-760,:680-685).  (1) a C++ exception raised INSIDE the library comes back as a code -- a
    std::bad_alloc unwinding into ctypes would abort this interpreter; (2) every export the header declares whose body is
    more than one statement is a function-try-block closed by LSQAMD_ABI_CATCH (source check: a new export cannot forget)."""
    from lsqfit_amd import _lib
    lib = _lib.load()
    assert lib.lsqamd_debug_throw(None, 0) == 0
    assert lib.lsqamd_debug_throw(None, 1) == -3      # LSQAMD_ENOMEM
    assert lib.lsqamd_debug_throw(None, 2) == -10     # LSQAMD_EINTERNAL
    assert lib.lsqamd_debug_throw(None, 3) == -10
    assert lib.lsqamd_debug_throw(None, 4) in (-3, -10)   # a real over-sized allocation (length_error or bad_alloc)
    import glob
    names = set(header_symbols())
    seen, unguarded = set(), []
    for path in glob.glob(os.path.join(ROOT, 'lsqfit_amd', 'csrc', '*.hip')):
        lines = open(path).read().split('\n')
        for i, ln in enumerate(lines):
            m = re.match(r'^(?:extern "C" )?[A-Za-z_0-9 \*]+?\b(lsqamdb?_[A-Za-z_0-9]+)\s*\(', ln)
            if not m or m.group(1) not in names:
                continue
            j = i
            while not lines[j].rstrip().endswith(('{', ';', '}')):
                j += 1
            end = lines[j].rstrip()
            if end.endswith(';') and not end.endswith('}'):
                continue                         # a declaration
            seen.add(m.group(1))
            if end.endswith('}'):
                assert end.count(';') == 1, (path, ln)        # a one-statement body: nothing in it can throw ...
                assert 'std::' not in end and 'new ' not in end, (path, ln)
                continue
            if not end.endswith('try {'):
                unguarded.append((os.path.basename(path), m.group(1)))
                continue
            k = j + 1
            while not lines[k].startswith('}'):
                k += 1
            assert lines[k].startswith('} LSQAMD_ABI_CATCH('), (path, m.group(1), lines[k])
    assert not unguarded, unguarded
    assert seen == names, sorted(names - seen)


def test_kernel_attribute_bookkeeping_is_per_device(libpath):
    """hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device only; the ABI lets one process hold
    handles on several GPUs from several threads.  The bookkeeping in front of every such call (csrc/common.h PerDeviceOnce)
    must run the setter once per DEVICE -- not once per process -- and exactly once under concurrent first launches."""
    import threading
    from lsqfit_amd import _lib
    lib = _lib.load()
    lib.lsqamd_debug_per_device_once(0, 1)
    assert lib.lsqamd_debug_per_device_once(0, 0) == 1
    assert lib.lsqamd_debug_per_device_once(0, 0) == 1      # second launch on device 0: not again
    assert lib.lsqamd_debug_per_device_once(1, 0) == 1      # first launch on device 1: its own
    assert lib.lsqamd_debug_per_device_once(7, 0) == 1
    assert lib.lsqamd_debug_per_device_once(1, 0) == 1
    assert lib.lsqamd_debug_per_device_once(64, 0) == 1     # beyond the table: the setter runs every time (correct, slower)
    assert lib.lsqamd_debug_per_device_once(64, 0) == 2
    lib.lsqamd_debug_per_device_once(0, 1)
    out = []
    go = threading.Barrier(8)

    def first_launch(dev):
        go.wait()
        for _ in range(200):
            out.append(lib.lsqamd_debug_per_device_once(dev, 0))
    ts = [threading.Thread(target=first_launch, args=(i % 2,)) for i in range(8)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert set(out) == {1}, sorted(set(out))
    # and no process-global "attribute set" flag is left in the sources
    import glob
    for path in glob.glob(os.path.join(ROOT, 'lsqfit_amd', 'csrc', '*.hip')):
        src = open(path).read()
        for m in re.finditer(r"hipFuncSetAttribute\(", src):
            before = src[max(0, m.start() - 900):m.start()]
            assert re.search(r'\.run\(\[|\.ensure\(', before), (path, src.count('\n', 0, m.start()) + 1)
        assert not re.search(r'static bool [a-z_]*attr', src), path


def test_grouped_work_list_of_the_exchange(libpath):
    """The J^T J launch of the grouped exchange: ONE list, every (tile, K-split) of the upper triangle exactly
    once, each XCD's run of the list in group-major order (so that a group's tiles are complete when that share of the launch
    has run), the runs as long as the kernel's blockIdx -> entry map assumes, diagonal tiles last inside a (run, group)."""
    import numpy as np
    from lsqfit_amd import _lib
    lib = _lib.load()
    ip = ctypes.POINTER(ctypes.c_int32)
    for P, splits, rows in ((4096, 4, [0, 5, 12, 32]), (4096, 16, [0, 9, 32]), (1024, 3, [0, 1, 2, 4, 8]), (384, 2, [0, 1, 3])):
        T = (P + 127) // 128
        G = len(rows) - 1
        nw = T * (T + 1) // 2 * splits
        plain = np.zeros((nw, 4), np.int32)
        assert lib.lsqamd_debug_syrk_work(P, splits, 0, None, plain.ctypes.data_as(ip), None) == nw
        out = np.full((nw, 4), -1, np.int32)
        count = np.zeros(G, np.int32)
        r = np.array(rows, np.int32)
        assert lib.lsqamd_debug_syrk_work(P, splits, G, r.ctypes.data_as(ip), out.ctypes.data_as(ip), count.ctypes.data_as(ip)) == nw
        key = lambda a: sorted(map(tuple, a[:, :3].tolist()))
        assert key(out) == key(plain) and len(set(map(tuple, out[:, :3].tolist()))) == nw          # a permutation of the plain list
        grp = out[:, 3] - 1
        assert np.all((out[:, 0] >= r[grp]) & (out[:, 0] < r[grp + 1]))                             # the group IS the tile row's group
        assert list(count) == [int(np.sum(grp == g)) for g in range(G)] and count.sum() == nw
        q, rem = divmod(nw, 8)
        start = 0
        for x in range(8):
            n = q + (1 if x < rem else 0)
            run = out[start:start + n]
            assert np.all(np.diff(run[:, 3]) >= 0)                                                  # group-major inside the run
            for g in range(G):
                seg = run[run[:, 3] == g + 1]
                diag = (seg[:, 0] == seg[:, 1]).astype(int)
                assert np.all(np.diff(diag) >= 0)                                                   # diagonal tiles last
                assert abs(len(seg) - count[g] / 8) <= G + 1                                        # an even share of every group
            start += n


def test_no_cpu_fallback():
    """Without a GPU the product refuses to fit (it must never route through the oracle)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    import numpy as np
    import lsqfit_amd
    with pytest.raises(RuntimeError, match='no MI355X'):
        lsqfit_amd.nonlinear_fit(data=(np.arange(4.), np.ones(4), np.ones(4)), model=lsqfit_amd.cosmix(1),
                                 p0=np.ones(2))
    import importlib
    for name in ('fitter', 'fit', 'whiten', 'dist', 'models', 'synth', '_lib', 'build'):
        mod = importlib.import_module('lsqfit_amd.' + name)
        src = open(mod.__file__).read()
        assert 'import oracle' not in src and 'from oracle' not in src, name


def test_unloadable_rccl_is_reported_not_crashed(libpath, tmp_path):
    """The header promises LSQAMD_EUNSUPPORTED when no usable librccl.so can be bound.  A named
    library that does not load must come back as that code (round 2 called dlerror() twice on this
    path and built a std::string from NULL)."""
    import subprocess
    import sys
    bogus = str(tmp_path / 'no_such_librccl.so')
    notlib = tmp_path / 'not_a_library.so'
    notlib.write_bytes(b'this is not an ELF file')
    prog = ('import ctypes, sys\n'
            'sys.path.insert(0, %r)\n'
            'from lsqfit_amd import _lib\n'
            'lib = _lib.load()\n'
            'buf = ctypes.create_string_buffer(128)\n'
            'print("rc", lib.lsqamd_comm_unique_id(buf, 128))\n' % ROOT)
    for path in (bogus, str(notlib)):
        env = dict(os.environ, LSQAMD_RCCL_PATH=path)
        r = subprocess.run([sys.executable, '-c', prog], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                           timeout=300)
        assert r.returncode == 0, r.stdout
        assert 'rc -6' in r.stdout, r.stdout


def test_every_module_of_the_package_compiles():
    """A syntax error in a module only the GPU tests import must not wait for the GPU box to be found."""
    import glob
    import py_compile
    files = glob.glob(os.path.join(ROOT, 'lsqfit_amd', '*.py')) + glob.glob(os.path.join(ROOT, 'tests', '*.py')) + \
        glob.glob(os.path.join(ROOT, 'oracle', '*.py')) + [os.path.join(ROOT, 'bench.py'), os.path.join(ROOT, '__graft_entry__.py')]
    assert len(files) > 40
    for f in files:
        compile(open(f).read(), f, "exec")
