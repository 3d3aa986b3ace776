"""Host logic of the product (lsqfit_amd.whiten / models / dist) against the oracle.
No GPU needed."""
import numpy as np
import pytest

from lsqfit_amd import models, synth
from lsqfit_amd.dist import shard_rows
from lsqfit_amd.whiten import Whitening
from oracle import fit as ofit
from oracle.pdf import PDF
from tests.helpers import load

KAT = load('kat.json')


def icov_from_whitening(wh):
    N = wh.n_data
    ic = np.zeros((N, N))
    d = np.ones(N, bool)
    for b in wh.blocks:
        r0, B = b['row0'], b['size']
        W = b['Wt'].T[:b['modes']]
        ic[r0:r0 + B, r0:r0 + B] = W.T @ W
        d[r0:r0 + B] = False
    idx = np.nonzero(d)[0]
    ic[idx, idx] = wh.wdiag[idx] ** 2
    return ic


@pytest.mark.parametrize('svdcut', [1e-12, 0.0, None, 1e-2, -1e-2])
def test_whitening_matches_oracle_pdf(svdcut):
    rng = np.random.default_rng(5)
    N, P = 24, 6
    cov = np.diag(rng.uniform(0.5, 2.0, N) ** 2)
    # two correlated blocks: rows 3..8 (nearly singular) and 12..15
    Q, _ = np.linalg.qr(rng.standard_normal((6, 6)))
    c1 = (Q * np.array([3.0, 1.0, 0.5, 0.2, 1e-3, 2e-6])) @ Q.T      # cond ~ 1e6
    cov[3:9, 3:9] = c1
    B = rng.standard_normal((4, 4))
    cov[12:16, 12:16] = B @ B.T + 0.1 * np.eye(4)
    ymean = rng.standard_normal(N)
    C = rng.standard_normal((P, P))
    pcov = C @ C.T + 0.5 * np.eye(P)
    pmean = rng.standard_normal(P)
    wh = Whitening(ymean, cov, pmean, pcov, svdcut=svdcut)
    pdf = ofit.build_pdf(ymean, cov, pmean, pcov, svdcut=svdcut)
    assert wh.nchiv == pdf.nchiv
    assert wh.nmod == pdf.nmod
    assert wh.nblocks == pdf.nblocks
    assert wh.logdet == pytest.approx(pdf.logdet, rel=1e-10, abs=1e-9)
    full = pdf.icov()
    np.testing.assert_allclose(icov_from_whitening(wh), full[:N, :N], rtol=1e-7, atol=1e-9 * np.abs(full).max())
    np.testing.assert_allclose(wh.prior_prec, full[N:, N:], rtol=1e-7, atol=1e-9 * np.abs(full).max())
    if svdcut in (1e-12, 0.0, None):
        assert all(b['tri'] == 1 for b in wh.blocks)       # Cholesky route when nothing is touched
        for b in wh.blocks:
            assert np.allclose(np.tril(b['Wt'], -1), 0.0)


def test_whitening_literal_weights():
    """tests/test_lsqfit.py:955-962,:1024-1032 through the product's host code."""
    k = KAT['unpack_case4']
    y = np.array(k['y'], float)
    p = np.array(k['prior'], float)
    wh = Whitening(y[:, 0], y[:, 1], p[:, 0], p[:, 1], svdcut=0)
    assert list(wh.wdiag) == k['wgts'][:2]
    assert list(np.sqrt(wh.prior_prec)) == k['wgts'][2:]
    assert wh.nchiv == 4 and wh.nmod == 0 and wh.nblocks == {1: 4}
    assert wh.logdet == pytest.approx(np.log(np.prod(np.concatenate([y[:, 1], p[:, 1]]) ** 2)))


def test_y_vs_x_whitening_one_mode_modified():
    k = KAT['y_vs_x']
    wh = Whitening(k['ymean'], np.array(k['ycov']), [0.5, 1.0], [0.4, 0.4], svdcut=1e-12)
    assert wh.nmod == 1 and wh.nblocks == {8: 1, 1: 2}
    assert wh.blocks[0]['tri'] == 0


def test_udata_drops_correlations():
    d = synth.make_cosmix(64, 8, 3, block=16)
    wh = Whitening(d['ymean'], d['yerr'], *d['prior'], udata=True)
    assert wh.blocks == [] and wh.nblocks == {1: 64 + 8}


def test_tape_compiler_matches_numpy():
    nist = load('nist.json')
    rng = np.random.default_rng(0)
    for name, d in nist.items():
        P = d['nparam']
        xn = [c for c in d['columns'][1:]]
        m = models.expr(d['expr'], ['b%d' % (i + 1) for i in range(P)], xn)
        assert m.kind == models.MODEL_TAPE and m.n_param == P and m.n_x == len(xn)
        # run the tape on the host with a tiny interpreter and compare with python eval
        data = np.array(d['data'], float)
        p = np.array(d['start2'], float)
        x = data[:, 1:]
        ns = dict(exp=np.exp, log=np.log, sin=np.sin, cos=np.cos, arctan=np.arctan, sqrt=np.sqrt, pi=np.pi)
        ns.update({c: x[:, i] for i, c in enumerate(xn)})
        ns.update({'b%d' % (i + 1): p[i] for i in range(P)})
        want = eval(d['expr'], {'__builtins__': {}}, ns)
        got = run_tape(m, x, p)
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-300)


def run_tape(m, x, p):
    O = models.OP
    st = []
    for ins in m.tape:
        op, arg = int(ins) & 0xff, int(ins) >> 8
        if op == O['CONST']:
            st.append(np.full(x.shape[0], m.consts[arg]))
        elif op == O['X']:
            st.append(x[:, arg].copy())
        elif op == O['P']:
            st.append(np.full(x.shape[0], p[arg]))
        elif op in (O['ADD'], O['SUB'], O['MUL'], O['DIV'], O['POW']):
            b = st.pop()
            a = st.pop()
            st.append({O['ADD']: a + b, O['SUB']: a - b, O['MUL']: a * b, O['DIV']: a / b,
                       O['POW']: a ** b}[op])
        elif op == O['POWI']:
            st.append(st.pop() ** float(arg))
        else:
            a = st.pop()
            st.append({O['NEG']: lambda v: -v, O['EXP']: np.exp, O['LOG']: np.log, O['SIN']: np.sin,
                       O['COS']: np.cos, O['ATAN']: np.arctan, O['SQRT']: np.sqrt}[op](a))
    assert len(st) == 1
    return st[0]


def test_shard_rows_respects_blocks():
    blocks = [(r, 16) for r in range(0, 128, 16)]
    for world in (1, 2, 3, 4, 8):
        rs = shard_rows(128, blocks, world)
        assert rs[0][0] == 0 and rs[-1][1] == 128
        for (a, b), (c, d) in zip(rs[:-1], rs[1:]):
            assert b == c
        for a, b in rs:
            assert a % 16 == 0 and b % 16 == 0
    assert shard_rows(10, [], 4) == [(0, 2), (2, 5), (5, 7), (7, 10)]
    # more ranks than blocks: trailing ranks get empty ranges, nothing is cut
    rs = shard_rows(32, [(0, 32)], 4)
    assert sum(b - a for a, b in rs) == 32 and all((a, b) in [(0, 0), (0, 32), (32, 32)] for a, b in rs)


def test_synth_generator_is_fake_fitargs_like():
    """src/lsqfit/_extras.py:2560-2589 recipe; acceptance chi2/dof < 6 (tests/test_lsqfit.py:1199-1200)."""
    d = synth.make_cosmix(96, 8, 7, block=24)
    assert np.all(d['yerr']['sdev'] > 0)
    for r0, cov in d['yerr']['blocks']:
        s = np.sqrt(np.diag(cov))
        np.testing.assert_allclose(s, d['yerr']['sdev'][r0:r0 + 24])
        corr = cov / np.outer(s, s)
        assert np.all(corr > 0) and np.allclose(np.diag(corr), 1.0)
    from oracle import dual
    K = 4
    fcn = lambda x, p: dual.stack_sum(p[k] * dual.cos(p[K + k] * x) for k in range(K))
    cov = np.diag(d['yerr']['sdev'] ** 2)
    for r0, c in d['yerr']['blocks']:
        cov[r0:r0 + 24, r0:r0 + 24] = c
    fit = ofit.nonlinear_fit(d['x'], d['ymean'], cov, fcn, prior_mean=d['prior'][0], prior_err=d['prior'][1])
    assert fit.chi2 / fit.dof < 6
    assert np.all(np.abs(fit.pmean - d['p_true']) < 6 * fit.psdev)


def test_draws_have_the_regulated_covariance():
    """Whitening.draw_data / draw_prior (what gvar.bootstrap_iter supplies to
    simulated_data_iter, src/lsqfit/__init__.py:1534-1543): sample covariance -> C_reg, and
    whitened draws are unit normal (W S = orthogonal)."""
    from lsqfit_amd.whiten import Whitening
    rng = np.random.default_rng(3)
    N = 12
    sd = rng.uniform(0.5, 2.0, N)
    U = rng.uniform(0.1, 0.9, (5, 10))
    c1 = U @ U.T
    U = rng.uniform(0.1, 0.9, (4, 8))
    c2 = U @ U.T
    c2[1] = c2[0] * (1 + 1e-9)          # nearly singular: the svdcut floors one mode
    c2[:, 1] = c2[1]
    c2[1, 1] = c2[0, 0] * (1 + 2e-9)
    yerr = dict(sdev=sd, blocks=[(2, c1), (8, c2)])
    pcov = np.array([[1.0, 0.3], [0.3, 0.5]])
    wh = Whitening(np.zeros(N), yerr, np.zeros(2), pcov, svdcut=1e-4)
    assert wh.nmod >= 1
    for k in wh.blocks:
        W = k['Wt'].T[:k['modes']]
        np.testing.assert_allclose(W @ k['S'] @ k['S'].T @ W.T, np.eye(k['modes']), atol=1e-8)
    n = 200000
    draws = wh.draw_data(np.random.default_rng(4), n)
    emp = draws.T @ draws / n
    want = np.diag(1.0 / wh.wdiag ** 2)
    for k in wh.blocks:
        sl = slice(k['row0'], k['row0'] + k['size'])
        want[sl, sl] = k['S'] @ k['S'].T
    scale = np.sqrt(np.outer(np.diag(want), np.diag(want)))
    assert np.abs((emp - want) / scale).max() < 0.02
    pd = wh.draw_prior(np.random.default_rng(5), n)
    assert np.abs(pd.T @ pd / n - pcov).max() < 0.01


def test_joint_whitening_of_correlated_data_and_prior():
    """whiten.joint_whitening: concat(y, prior) with data-prior cross-covariance is permuted so that
    each covariance block is contiguous; the weights reproduce inv(C) of the joint vector and the
    bookkeeping (row_src, row_param, model_rows) maps rows back."""
    from lsqfit_amd.whiten import joint_whitening
    rng = np.random.default_rng(5)
    N, P = 7, 3
    A = rng.standard_normal((N + P, N + P))
    full = A @ A.T + (N + P) * np.eye(N + P)
    # make data rows 0, 1 and prior entry 2 independent of everything (1x1 components)
    for i in (0, 1, N + 2):
        full[i, :] = 0.0
        full[:, i] = 0.0
        full[i, i] = 1.0 + i
    ym, pm = rng.standard_normal(N), rng.standard_normal(P)
    wh = joint_whitening(ym, full[:N, :N], pm, full[N:, N:], full[:N, N:], svdcut=1e-12)
    assert wh.n_data == N + P and not wh.has_prior and wh.nmod == 0
    assert wh.nblocks == {1: 3, N + P - 3: 1}
    assert sorted(wh.row_src) == list(range(N + P))
    assert np.array_equal(wh.row_src[wh.model_rows], np.sort(wh.row_src[wh.model_rows]))   # data keep their order
    assert np.array_equal(wh.row_param >= 0, wh.row_src >= N)
    assert np.array_equal(wh.row_param[wh.row_param >= 0], wh.row_src[wh.row_src >= N] - N)
    np.testing.assert_allclose(wh.ymean, np.concatenate([ym, pm])[wh.row_src])
    # chi2 of a random deviation through the weights == delta^T inv(C) delta
    delta = rng.standard_normal(N + P)
    dperm = delta[wh.row_src]
    in_block = np.zeros(N + P, bool)
    chi2 = 0.0
    for b in wh.blocks:
        r0, B, m = b['row0'], b['size'], b['modes']
        in_block[r0:r0 + B] = True
        chi2 += float(np.sum((b['Wt'].T[:m] @ dperm[r0:r0 + B]) ** 2))
    chi2 += float(np.sum((wh.wdiag[~in_block] * dperm[~in_block]) ** 2))
    assert chi2 == pytest.approx(float(delta @ np.linalg.solve(full, delta)), rel=1e-10)
    sign, ld = np.linalg.slogdet(full)
    assert wh.logdet == pytest.approx(ld, rel=1e-10)


def _interleaved_cov(rng, N):
    """Components {1,4,9,10}, {2,7}, {5,6,13} (the first nearly singular); everything else 1x1."""
    cov = np.diag(rng.uniform(0.5, 2.0, N) ** 2)
    for idx, small in (([1, 4, 9, 10], 1e-5), ([2, 7], 0.3), ([5, 6, 13], 0.2)):
        B = len(idx)
        Q, _ = np.linalg.qr(rng.standard_normal((B, B)))
        ev = np.concatenate([np.linspace(2.0, 0.5, B - 1), [small]])
        cov[np.ix_(idx, idx)] = (Q * ev) @ Q.T
    return cov


@pytest.mark.parametrize('svdcut', [1e-12, 1e-3, -1e-3])
def test_interleaved_components_are_whitened_through_a_permutation(svdcut):
    """gvar's block search returns index sets, not ranges (tests/test_lsqfit.py:1011-1012): the
    whitening reorders the rows so each set is contiguous and must agree with the oracle PDF, for
    the data and for the prior."""
    rng = np.random.default_rng(17)
    N, P = 16, 14
    cov = _interleaved_cov(rng, N)
    pcov = _interleaved_cov(rng, 16)[:P, :P]          # prior components {1,4,9,10}, {2,7}, {5,6,13}
    ymean, pmean = rng.standard_normal(N), rng.standard_normal(P)
    wh = Whitening(ymean, cov, pmean, pcov, svdcut=svdcut)
    pdf = ofit.build_pdf(ymean, cov, pmean, pcov, svdcut=svdcut)
    assert wh.perm is not None and sorted(wh.perm) == list(range(N))
    assert [b['size'] for b in wh.blocks] == [4, 2, 3]
    np.testing.assert_array_equal(wh.ymean, ymean[wh.perm])
    assert (wh.nchiv, wh.nmod, wh.nblocks) == (pdf.nchiv, pdf.nmod, pdf.nblocks)
    assert wh.logdet == pytest.approx(pdf.logdet, rel=1e-10, abs=1e-9)
    full = pdf.icov()
    ic = np.empty((N, N))
    ic[np.ix_(wh.perm, wh.perm)] = icov_from_whitening(wh)        # back to the caller's row order
    scale = np.abs(full).max()
    np.testing.assert_allclose(ic, full[:N, :N], rtol=1e-7, atol=1e-9 * scale)
    np.testing.assert_allclose(wh.prior_prec, full[N:, N:], rtol=1e-7, atol=1e-9 * scale)
    kind, drows, brows = wh.prior_W
    W = np.vstack([drows] + brows)
    np.testing.assert_allclose(W.T @ W, full[N:, N:], rtol=1e-7, atol=1e-9 * scale)
    Cs = np.zeros((P, P))
    for idx, S in wh.prior_S:                          # S S^T = the regulated covariance, in place
        Cs[np.ix_(idx, idx)] = S @ S.T if np.ndim(S) == 2 else np.diag(S ** 2)
    if svdcut > 0:
        np.testing.assert_allclose(Cs @ full[N:, N:], np.eye(P), atol=1e-6)
    # contiguous layouts are untouched
    assert Whitening(ymean, np.diag(np.diag(cov)), pmean, np.sqrt(np.diag(pcov))).perm is None


def test_eps_regulation_is_a_diagonal_shift_of_the_correlation_matrix():
    """``eps`` (src/lsqfit/__init__.py:240-245, gvar.regulate's second mode): every correlated block's
    correlation matrix gets eps * ||corr||_inf on its diagonal.  The reference holds no expected value
    for it (unpinned); what is checked is the stated identity, against an inverse computed here."""
    rng = np.random.default_rng(41)
    N, P, eps = 16, 14, 1e-3
    cov = _interleaved_cov(rng, N)
    pcov = _interleaved_cov(rng, 16)[:P, :P]
    ymean, pmean = rng.standard_normal(N), rng.standard_normal(P)

    def shifted(c):
        out = c.copy()
        for comp in (np.array(k) for k in ([1, 4, 9, 10], [2, 7], [5, 6, 13])):
            sd = np.sqrt(np.diag(c)[comp])
            corr = c[np.ix_(comp, comp)] / np.outer(sd, sd)
            out[comp, comp] += eps * np.abs(corr).sum(axis=1).max() * sd ** 2
        return out
    wh = Whitening(ymean, cov, pmean, pcov, svdcut=None, eps=eps)
    assert wh.eps == eps and wh.svdcut is None
    ic = np.empty((N, N))
    ic[np.ix_(wh.perm, wh.perm)] = icov_from_whitening(wh)
    np.testing.assert_allclose(ic, np.linalg.inv(shifted(cov)), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(wh.prior_prec, np.linalg.inv(shifted(pcov)), rtol=1e-9, atol=1e-12)
    assert wh.logdet == pytest.approx(np.linalg.slogdet(shifted(cov))[1] + np.linalg.slogdet(shifted(pcov))[1], rel=1e-11)
    assert wh.nmod == 2 * (4 + 2 + 3)                  # every mode of every correlated block moved
    pdf = ofit.build_pdf(ymean, cov, pmean, pcov, svdcut=None, eps=eps)
    assert (wh.nchiv, wh.nmod, wh.nblocks) == (pdf.nchiv, pdf.nmod, pdf.nblocks)
    assert wh.logdet == pytest.approx(pdf.logdet, rel=1e-10)
    # eps is ignored when an svdcut is given (and not None)
    w2 = Whitening(ymean, cov, pmean, pcov, svdcut=1e-12, eps=eps)
    assert w2.eps is None and w2.nmod == 0
    with pytest.raises(ValueError):
        Whitening(ymean, cov, svdcut=None, eps=-1.0)


@pytest.mark.parametrize('mode', ['eps', 'svdcut'])
def test_noise_is_a_draw_from_what_the_regulation_added(mode):
    """noise=(True, False) (src/lsqfit/__init__.py:247-256): the data means move by a Gaussian draw whose
    covariance is C_regulated - C; repeated with the same seed it repeats, and over many draws its
    sample covariance is that difference."""
    rng = np.random.default_rng(43)
    N = 16
    cov = _interleaved_cov(rng, N)
    ymean = rng.standard_normal(N)
    kw = dict(svdcut=None, eps=0.05) if mode == 'eps' else dict(svdcut=0.05)
    quiet = Whitening(ymean, cov, engine='host', **kw)
    ic = np.empty((N, N))
    ic[np.ix_(quiet.perm, quiet.perm)] = icov_from_whitening(quiet)
    added = np.linalg.inv(ic) - cov
    assert np.abs(added).max() > 1e-3                # the regulation does something here
    a = Whitening(ymean, cov, engine='host', noise=(True, False), rng=7, **kw)
    b = Whitening(ymean, cov, engine='host', noise=(True, False), rng=7, **kw)
    np.testing.assert_array_equal(a.ymean, b.ymean)
    assert not np.array_equal(a.ymean, quiet.ymean)
    inv_perm = np.argsort(quiet.perm)
    shifts = np.array([(Whitening(ymean, cov, engine='host', noise=True, rng=1000 + k, **kw).ymean - quiet.ymean)[inv_perm]
                       for k in range(4000)])
    np.testing.assert_allclose(shifts.T @ shifts / len(shifts), added, atol=0.08 * np.abs(added).max())
    # prior noise: a draw from the prior itself
    pm, psd = np.zeros(5), np.full(5, 2.0)
    d = np.array([Whitening(ymean, cov, pm, psd, engine='host', noise=(False, True), rng=k, **kw).prior_mean
                  for k in range(2000)])
    assert abs(d.std() - 2.0) < 0.1 and abs(d.mean()) < 0.1
