"""Build recipe for the gfx950 shared library (hipcc, in-tree).

    python -m lsqfit_amd.build        # -> lsqfit_amd/liblsqfit_amd.so

hipcc cross-compiles for gfx950 without a GPU; the resulting .so is git-ignored
but travels with the tree to the GPU box.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(HERE, 'build')
LIB = os.path.join(HERE, 'liblsqfit_amd.so')
SOURCES = ['gemm_tn_f64.hip', 'chol.hip', 'potf2_mfma.hip', 'model.hip', 'vecops.hip', 'api.hip', 'scipy_methods.hip',
           'batch.hip', 'comm.hip', 'whiten.hip', 'qr.hip', 'jit.hip', 'rankdef.hip', 'robust.hip']
# per-file code-generation switches (reasons in the files' headers)
EXTRA = {'potf2_mfma.hip': ['-mllvm', '-amdgpu-mfma-vgpr-form']}
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function']


def _newer(target, deps):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = os.environ.get('HIPCC', 'hipcc')
    headers = [os.path.join(CSRC, 'common.h'), os.path.join(CSRC, 'fit_state.h'), os.path.join(CSRC, 'devmath.h'),
               os.path.join(CSRC, 'jit.h'), os.path.join(HERE, '..', 'include', 'lsqfit_amd.h')]
    # devmath.h as a string literal for the run-time compiled tapes (jit.hip embeds it in the code it generates, so
    # that compiled formulas and the hand-written kernels share one sincos); generated, git-ignored
    text = open(os.path.join(CSRC, 'devmath.h')).read()
    text = '\n'.join(l for l in text.splitlines() if not l.startswith('#pragma once') and not l.startswith('#include'))
    inc = 'R"LSQJIT(\n' + text + '\n)LSQJIT"\n'
    incpath = os.path.join(CSRC, 'devmath_src.inc')
    if not os.path.exists(incpath) or open(incpath).read() != inc:
        with open(incpath, 'w') as fh:
            fh.write(inc)
    headers.append(incpath)
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace('.hip', '.o'))
        objs.append(o)
        if force or not _newer(o, [s] + headers):
            jobs.append([hipcc] + FLAGS + EXTRA.get(src, []) + ['-c', s, '-o', o])

    def run(cmd):
        if verbose:
            print(' '.join(cmd))
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed:\n%s\n%s' % (' '.join(cmd), r.stdout))
        return r.stdout

    with ThreadPoolExecutor(max_workers=4) as ex:
        for out in ex.map(run, jobs):
            if verbose and out.strip():
                print(out)
    if jobs or not os.path.exists(LIB):
        run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs + ['-ldl'])
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
