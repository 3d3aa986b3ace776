"""-m "not gpu": the tape compiler (lsqfit_amd/csrc/jit.hip) -- plans, generated source and the
hiprtc build for gfx950 (hiprtc needs no GPU).  What the generated kernels COMPUTE is checked on the
device (tests/test_gpu_tape.py, test_gpu_parity.py: every tape fit runs through them)."""
import ctypes as C

import numpy as np
import pytest

from tests.helpers import load


@pytest.fixture(scope='module')
def lib():
    from lsqfit_amd import _lib
    return _lib.load()


def codegen(lib, model, compile=1):
    from lsqfit_amd import _lib
    code = np.ascontiguousarray(model.tape, np.int32)
    consts = np.ascontiguousarray(model.consts, np.float64)
    buf = C.create_string_buffer(1 << 22)
    var = C.c_int32(-1)
    rc = lib.lsqamd_tape_codegen(code.ctypes.data_as(C.POINTER(C.c_int32)), code.size, _lib.dptr(consts), consts.size,
                                 model.n_param, model.n_x, buf, len(buf), C.byref(var), compile)
    return rc, var.value, buf.value.decode()


def body(src, kernel):
    return src[src.index('void %s(' % kernel):].split('extern "C"')[0]


@pytest.mark.parametrize('name', sorted(load('nist.json')))
def test_nist_models_compile_for_gfx950(lib, name):
    """The 27 formulas of examples/nist.py: one lane per data row, the whole expression in registers."""
    import lsqfit_amd as amd
    d = load('nist.json')[name]
    model = amd.expr(d['expr'], ['b%d' % (i + 1) for i in range(d['nparam'])], d['columns'][1:])
    rc, variant, src = codegen(lib, model)
    assert rc == 0, src[:2000]
    assert variant == 0
    for j in range(d['nparam']):          # every parameter gets its column, the residual the last one
        assert 'dst[%d] = w * oacc' % j in body(src, 'lsqamd_jit_jac')
    assert 'dst[%d] = w * (fval - a.ymean[row])' % d['nparam'] in src
    # few parameters: the third kernel that forms J^T J, J^T f and chi2 without writing the Jacobian
    nq = d['nparam'] * (d['nparam'] + 1) // 2 + d['nparam'] + 1
    nrm = body(src, 'lsqamd_jit_nrm').split('constexpr int LP')[0]
    assert '__shared__ double red[4][%d];' % nq in nrm and 'nC += rr * rr;' in nrm
    assert nrm.count(' += dd') == nq - 1
    # ... and the whole-fit kernel over the same sums as functions of one workgroup
    assert 'constexpr int LP = %d, LNA = %d, LNQ = %d,' % (d['nparam'], nq - d['nparam'] - 1, nq) in src
    lm = src[src.index('static __device__ void lm_nrm('):]
    assert lm.split('static __device__ double lm_res(')[0].count(' += dd') == nq - 1
    assert 'void lsqamd_jit_lm(LmArgs a)' in lm and 'lm_solve(sA, sG, sD, sp' in lm


def test_wide_sum_becomes_one_loop_with_contiguous_columns(lib):
    import lsqfit_amd as amd
    rc, variant, src = codegen(lib, amd.models.tape_sum('a*cos(w*x)', 512))
    assert rc == 0 and variant == 1
    jac = body(src, 'lsqamd_jit_jac')
    assert jac.count('for (int k = lane; k < 512; k += 64)') == 1          # adjoint +1 known beforehand: ONE loop
    assert 'dst[(0 + k)] = w * e0;' in jac and 'dst[(512 + k)] = w * e1;' in jac
    assert 'sincos_moderate<true>' in jac and 'cos_moderate<true>' in body(src, 'lsqamd_jit_res')
    assert 'T0_' not in src                                                # affine parameter indices: no table
    assert 'lsqamd_jit_nrm' not in src                                     # wide sums: no register-resident normal equations


def test_sum_inside_a_product_takes_its_adjoint_from_the_outer_sweep(lib):
    import lsqfit_amd as amd
    K = 20
    names = ['a%d' % k for k in range(K)] + ['w%d' % k for k in range(K)] + ['g', 'c', 'phi']
    text = 'c + exp(-g*x)*(' + '+'.join('a%d*cos(w%d*x+phi)' % (k, k) for k in range(K)) + ')'
    rc, variant, src = codegen(lib, amd.expr(text, names))
    assert rc == 0 and variant == 1
    jac = body(src, 'lsqamd_jit_jac')
    assert jac.count('for (int k = lane; k < 20; k += 64)') == 2          # values, then derivatives scaled by the adjoint
    assert 'const double tadj = aS0;' in jac
    assert 'sh2 += e2;' in jac and 'oacc2 += wsum(sh2);' in jac           # phi: shared by all terms, summed over the wave
    assert 'mycol = 40' in jac and 'mycol = 41' in jac and 'mycol = 42' in jac


def test_scattered_parameter_indices_use_a_table(lib):
    import lsqfit_amd as amd
    names = ['p%d' % i for i in range(32)]
    order = [3, 0, 7, 5, 1, 6, 2, 4, 11, 8, 15, 13, 9, 14, 10, 12]
    text = '+'.join('p%d*exp(-p%d*x)' % (order[k], 16 + k) for k in range(16))
    rc, variant, src = codegen(lib, amd.expr(text, names))
    assert rc == 0 and variant == 1
    assert 'static __device__ const int T0_0[16] = {3,0,7,5,1,6,2,4,11,8,15,13,9,14,10,12,};' in src
    assert 'dst[T0_0[k]] = w * e0;' in body(src, 'lsqamd_jit_jac')


def test_parameter_shared_between_two_sums_falls_back_to_the_outer_expression(lib):
    """A parameter that is private in one group but read elsewhere too cannot be a plain store."""
    import lsqfit_amd as amd
    names = ['a%d' % k for k in range(16)] + ['w%d' % k for k in range(16)]
    text = '+'.join('a%d*cos(w%d*x)' % (k, k) for k in range(16)) + ' + a0*x'
    rc, variant, src = codegen(lib, amd.expr(text, names))
    assert rc == 0 and variant == 0         # the group is not formed: everything is outer, one lane per row
    # ... and a short sum (lsqfit's canonical 2-6 exponentials) is unrolled in the one-lane-per-row form on purpose,
    # with the register-resident normal equations
    rc, variant, src = codegen(lib, amd.expr('+'.join('a%d*exp(-w%d*x)' % (k, k) for k in range(4)), names[:4] + names[16:20]))
    assert rc == 0 and variant == 0 and 'lsqamd_jit_nrm' in src


def test_formula_outside_the_generator_is_declined_not_miscompiled(lib):
    import lsqfit_amd as amd
    names = ['p%d' % i for i in range(130)]
    text = '*'.join('(p%d + x)' % i for i in range(130))          # 130 parameters, no sums to stride over
    rc, variant, src = codegen(lib, amd.expr(text, names), compile=0)
    assert rc == -6 and 'too many parameters' in src


def test_unread_parameter_gets_a_zero_column(lib):
    import lsqfit_amd as amd
    rc, variant, src = codegen(lib, amd.expr('py + 0*x', ['py', 'pn']))
    assert rc == 0 and 'static __device__ const int ZC[1] = {1,};' in src
    assert 'dst[ZC[k]] = 0.0;' in body(src, 'lsqamd_jit_jac')


def test_constants_that_differ_from_term_to_term_become_a_table(lib):
    """A Fourier series with literal harmonics: one group, the k's in a table (their values are not part of the
    structure); a constant shared by all terms stays a literal."""
    import lsqfit_amd as amd
    names = ['a%d' % k for k in range(1, 17)]
    text = ' + '.join('a%d*cos(%d*x*0.5)' % (k, k) for k in range(1, 17))
    rc, variant, src = codegen(lib, amd.expr(text, names))
    assert rc == 0 and variant == 1
    assert 'static __device__ const double CT0_0[16] = {0x1p+0,0x1p+1,0x1.8p+1,0x1p+2,0x1.4p+2,0x1.8p+2,0x1.cp+2,0x1p+3,' in src
    jac = body(src, 'lsqamd_jit_jac')
    assert 'CT0_0[k]' in jac and '0x1p-1' in jac and 'CT0_1' not in src


def test_whole_fit_kernel_forms(lib):
    """The whole-fit kernel (lsqamd_jit_lm): register sums up to a dozen parameters, rows through LDS up to 32, none beyond
    that or for formulas with wide sums (one wave per data row); the batched entry point is a module of its own."""
    import lsqfit_amd as amd

    def peaks(K):
        names = ['a%d' % k for k in range(K)] + ['b%d' % k for k in range(K)]
        return amd.expr(' + '.join('a%d*exp(-b%d*(x - %d.5)**2)' % (k, k, k) for k in range(K)), names)
    rc, variant, src = codegen(lib, peaks(4))                 # P = 8: sums in registers
    assert rc == 0 and variant == 0 and 'void lsqamd_jit_lm(LmArgs a)' in src
    assert 'constexpr int LP = 8,' in src and 'LROWS = 256, LRED = %d;' % (16 * (8 * 9 // 2 + 9)) in src and 'QI[LNQ]' not in src
    assert 'lsqamd_jit_lmb' not in src                        # (built on demand, from the same generator)
    rc, variant, src = codegen(lib, peaks(15))                # P = 30: rows through LDS, 128 at a time
    assert rc == 0 and variant == 0 and 'void lsqamd_jit_lm(LmArgs a)' in src
    assert 'constexpr int LP = 30,' in src and 'LROWS = 128, LRED = 16;' in src and 'QI[LNQ]' in src and 'QJ[LNQ]' in src
    assert 'lsqamd_jit_nrm' not in src                        # (the many-workgroup sums stop at a dozen parameters)
    rc, variant, src = codegen(lib, peaks(9))                 # P = 18: 256 rows at a time
    assert rc == 0 and 'LROWS = 256, LRED = 16;' in src
    rc, variant, src = codegen(lib, peaks(17), compile=0)     # P = 34: beyond the kernel
    assert rc == 0 and 'lsqamd_jit_lm' not in src
    rc, variant, src = codegen(lib, amd.models.tape_sum('a*cos(w*x)', 16), compile=0)   # 16 look-alike terms: one wave per row
    assert rc == 0 and variant == 1 and 'lsqamd_jit_lm' not in src


def test_hyperbolic_and_inverse_functions_compile(lib):
    """tan sinh cosh tanh arcsin arccos abs (LSQAMD_OP_TAN .. LSQAMD_OP_ABS): generated and built for gfx950, the whole-fit
    kernel and its hand-off included."""
    import lsqfit_amd as amd
    model = amd.expr('a*cosh(b*(x - 16))/cosh(16*b) + c*tanh(sinh(d*x)) + abs(tan(0.1*a*x)) + arcsin(0.1*b) + arccos(0.1*c) + fabs(d)',
                     ['a', 'b', 'c', 'd'])
    assert sorted(set(int(t) & 0xff for t in model.tape) & set(range(16, 23))) == list(range(16, 23))
    rc, variant, src = codegen(lib, model)
    assert rc == 0, src[:2000]
    for fn in ('cosh(', 'sinh(', 'tanh(', 'tan(', 'asin(', 'acos(', 'fabs('):
        assert fn in body(src, 'lsqamd_jit_jac')
    lm = src[src.index('static __device__ __forceinline__ void lm_fit('):]
    # the published block is self-verifying: checksum word, then the flag word that carries the sequence number
    assert 'a.pub[23] =' in lm and 'a.pub[16] =' in lm and lm.index('a.pub[23] =') < lm.index('a.pub[16] =')
    with pytest.raises(ValueError):
        amd.expr('a*erf(x)', ['a'])


def test_disk_cache_is_verified_and_private(lib, tmp_path, monkeypatch):
    """Cached code objects carry a length and a checksum: a truncated or damaged file is thrown away and rebuilt, never handed
    to the loader; a cache directory others can write to is not used at all."""
    import os
    import struct
    import lsqfit_amd as amd
    model = amd.expr('a*exp(-b*x) + 0.125*c', ['a', 'b', 'c'])
    d = tmp_path / 'cache'
    monkeypatch.setenv('LSQAMD_JIT_CACHE', str(d))
    assert codegen(lib, model)[0] == 0
    files = [f for f in os.listdir(d) if f.endswith('.hsaco')]
    assert len(files) == 1
    path = os.path.join(d, files[0])
    blob = open(path, 'rb').read()
    magic, n, h = struct.unpack('<QQQ', blob[:24])
    assert magic == int.from_bytes(b'LSQAMDJ1', 'little') and n == len(blob) - 24 and blob[24:28] == b'\x7fELF'
    for damaged in (blob[:len(blob) // 2], blob[:24] + bytes([blob[24] ^ 1]) + blob[25:], blob + b'x', b'', blob[24:]):
        with open(path, 'wb') as fh:
            fh.write(damaged)
        assert codegen(lib, model)[0] == 0
        assert open(path, 'rb').read() == blob            # rebuilt, byte for byte (hiprtc is deterministic)
    # a directory that is group / world writable (or not ours) is ignored: nothing is read from it, nothing written
    shared = tmp_path / 'shared'
    shared.mkdir()
    os.chmod(shared, 0o777)
    monkeypatch.setenv('LSQAMD_JIT_CACHE', str(shared))
    assert codegen(lib, model)[0] == 0 and os.listdir(shared) == []
