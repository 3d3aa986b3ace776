"""Pin the oracle on the literal known answers of the reference's tests/examples
(tests/golden/kat.json, transcribed by tests/golden/make_golden.py)."""
import re

import numpy as np
import pytest

from oracle import dual, gvar_lite
from oracle import fit as ofit
from oracle import lm as olm
from oracle.pdf import PDF
from tests.helpers import load

KAT = load('kat.json')


def header(fit):
    chi2_dof = fit.chi2 / fit.dof
    return chi2_dof, fit.dof, fit.Q, fit.logGBF


def parse_header(text):
    h = re.search(r'chi2/dof \[dof\] = (\S+) \[(\d+)\]\s+Q = (\S+)\s+logGBF = (\S+)', text)
    return h.group(1), int(h.group(2)), h.group(3), h.group(4)


def check_header(fit, text, sig_gbf=5):
    c, d, q, g = parse_header(text)
    assert fit.dof == d
    assert '%.2g' % (fit.chi2 / fit.dof) == c, (fit.chi2 / fit.dof, c)
    assert '%.2g' % fit.Q == q or abs(fit.Q - float(q)) < 0.006
    assert '%.*g' % (sig_gbf, fit.logGBF) == g, (fit.logGBF, g)


def test_gvar_strings():
    assert gvar_lite.parse('0.0005502(73)') == pytest.approx((0.0005502, 73e-7))
    assert gvar_lite.parse('238.9(2.7)') == pytest.approx((238.9, 2.7))
    assert gvar_lite.parse('0(47788)') == (0.0, 47788.0)
    assert gvar_lite.parse('0.00(11)') == pytest.approx((0.0, 0.11))
    assert gvar_lite.parse('0.0(2.5)e-05') == pytest.approx((0.0, 2.5e-5))
    assert gvar_lite.parse('-1.43(28)e-06') == pytest.approx((-1.43e-6, 0.28e-6))
    assert gvar_lite.parse('0 +- 1.0187876330E-01') == pytest.approx((0.0, 0.1018787633))
    assert gvar_lite.fmt(0.9038, 0.098058) == '0.904(98)'
    assert gvar_lite.fmt(238.942, 2.707) == '238.9(2.7)'
    assert gvar_lite.fmt(-1.43e-6, 0.28e-6) == '-1.43(28)e-06'


def test_fitters_conformance():
    """tests/test_lsqfit.py:1811-1833."""
    k = KAT['test_fitters']
    ym, ys = gvar_lite.parse_array(k['data'])
    pm, ps = gvar_lite.parse_array(k['prior'])
    for solver in ['qr', 'cholesky', 'svd']:
        fit = ofit.nonlinear_fit(False, ym, ys, lambda p: p, prior_mean=pm, prior_err=ps, solver=solver)
        assert gvar_lite.fmt_array(fit.pmean, fit.psdev) == k['expected_p']
    # closed form
    w = 1 / ys ** 2 + 1 / ps ** 2
    np.testing.assert_allclose(fit.pmean, (ym / ys ** 2 + pm / ps ** 2) / w, rtol=1e-9)
    np.testing.assert_allclose(fit.cov, np.diag(1 / w), rtol=1e-10, atol=1e-18)


def test_gammaQ():
    """tests/test_lsqfit.py:1887-1901."""
    for a, x, gax, gxa in KAT['gammaQ']:
        np.testing.assert_allclose(gax, ofit.gammaQ(a, x), rtol=0.01)
        np.testing.assert_allclose(gxa, ofit.gammaQ(x, a), rtol=0.01)


def test_gsl_multifit_direct():
    """tests/test_lsqfit.py:1700-1725: singular-J^T J double root; lm stops on xtol, lmaccel on
    gtol, subspace2D on xtol."""
    k = KAT['gsl_multifit']
    xans = np.array(k['xans'])
    f = lambda x: (x - xans) ** 2 + (x - xans) ** 4
    df = lambda x: np.diag(2 * (x - xans) + 4 * (x - xans) ** 3)
    for c in k['cases']:
        for solver in ['qr', 'cholesky']:
            ans = olm.gsl_multifit(np.array(c['x0']), 3, f, df, alg=c['alg'], tol=tuple(c['tol']),
                                   solver=solver)
            np.testing.assert_allclose(ans.x, xans, rtol=c['rtol'])
            assert ans.stopping_criterion == c['stopping_criterion']
    ans = olm.gsl_multifit(np.zeros(3), 3, f, df, tol=(0.0, 1e-10, 0.0))
    np.testing.assert_allclose(ans.x, xans, rtol=1e-3)
    assert ans.stopping_criterion == 2
    for alg in ['dogleg', 'ddogleg']:
        ans = olm.gsl_multifit(np.zeros(3), 3, f, df, alg=alg, tol=(1e-10, 0.0, 0.0))
        np.testing.assert_allclose(ans.x, xans, rtol=1e-3)
        assert ans.stopping_criterion == 1
    assert olm.gsl_multifit(np.zeros(3), 3, f, df, alg='lmaccel').description == \
        'methods = lmaccel/more/qr    avmax = 0.75'                # _gsl.pyx:614-618
    with pytest.raises(ValueError):
        olm.gsl_multifit(np.zeros(3), 3, f, df, alg='cgst')        # _gsl.pyx:632-635


def test_tol_normalisation():
    assert olm.normalize_tol(1e-5) == (1e-5, 1e-10, 1e-10)
    assert olm.normalize_tol((1e-5,)) == (1e-5, 1e-10, 1e-10)
    assert olm.normalize_tol((1e-5, 1e-6)) == (1e-5, 1e-6, 1e-10)
    with pytest.raises(ValueError):
        olm.normalize_tol((1, 2, 3, 4))


def test_exceptions_propagate():
    """tests/test_lsqfit.py:1684-1698: errors raised in the fit function surface unchanged."""
    with pytest.raises(ValueError):
        ofit.nonlinear_fit(False, [1., 2.], [1., 1.], lambda p: np.array([p[0]] * 4),
                           prior_mean=[0.], prior_err=[2.])
    with pytest.raises(ZeroDivisionError):
        def f(p):
            1 / 0.
        ofit.nonlinear_fit(False, [1., 2.], [1., 1.], f, prior_mean=[0.], prior_err=[2.])


def test_format_case1():
    """tests/test_lsqfit.py:257-283."""
    k = KAT['format1']
    y = np.array(k['y'])
    pr = np.array(k['prior'])
    fit = ofit.nonlinear_fit(False, y[:, 0], y[:, 1], lambda p: dual.concatenate([p, p]),
                             prior_mean=pr[:, 0], prior_err=pr[:, 1], svdcut=k['svdcut'], tol=tuple(k['tol']))
    c, d, q, g = parse_header(k['header'])
    assert fit.dof == d and '%.1g' % (fit.chi2 / fit.dof) == c
    assert '%.2g' % fit.Q == q and '%.5g' % fit.logGBF == g
    assert gvar_lite.fmt(fit.pmean[0], fit.psdev[0]) == k['p'].replace(' ', '')


def test_logGBF_closed_form():
    """tests/test_lsqfit.py:845-868 with a fixed draw for y."""
    ygm, ygs = gvar_lite.parse_array(KAT['logGBF']['yg'])
    y = ygm + ygs * np.array([0.3, -1.1, 0.7, 0.2])
    pm, ps = ygm, 0.5 * ygs
    yvar = ygs ** 2 + ps ** 2
    logprob = np.sum(-(y - ygm) ** 2 / (2 * yvar) - 0.5 * np.log(2 * np.pi * yvar))
    chi2 = np.sum((y - ygm) ** 2 / yvar)
    fit = ofit.nonlinear_fit(False, y, ygs, lambda p: p, prior_mean=pm, prior_err=ps)
    assert fit.logGBF == pytest.approx(logprob, abs=1e-7)
    assert fit.chi2 == pytest.approx(chi2, abs=1e-7)


def test_unpack_data_weights():
    """tests/test_lsqfit.py:955-962,:1024-1032,:1000-1017."""
    for key in ['unpack_case2', 'unpack_case4']:
        k = KAT[key]
        y = np.array(k['y'], float)
        p = np.array(k['prior'], float)
        pdf = ofit.build_pdf(y[:, 0], y[:, 1], p[:, 0], p[:, 1], svdcut=0)
        assert list(pdf.i_invwgts[0][0]) == k['idx']
        assert list(pdf.i_invwgts[0][1]) == k['wgts']
        assert len(pdf.i_invwgts) == 1 and pdf.nmod == 0
    # data-prior correlation: y[0] *= one, p[0] *= one with one = 1 +- 1e-3
    cov = np.diag([4., 16., 4., 16.])
    mean = np.array([1., 10., 1., 1.])
    cov[0, 0] += 1e-6 * 1
    cov[2, 2] += 1e-6 * 1
    cov[0, 2] = cov[2, 0] = 1e-6
    pdf = PDF.from_dense(mean, cov, svdcut=0)
    assert list(pdf.i_invwgts[0][0]) == [1, 3]
    assert pdf.i_invwgts[0][1].ndim == 1
    assert list(pdf.i_invwgts[1][0]) == [0, 2]
    assert pdf.i_invwgts[1][1].ndim == 2
    np.testing.assert_allclose(pdf.icov(), np.linalg.inv(cov), rtol=1e-8, atol=1e-15)
    assert pdf.logdet == pytest.approx(np.log(np.linalg.det(cov)))
    assert pdf.nblocks == {1: 2, 2: 1}


def test_svdcut_positive_and_negative():
    """tests/test_lsqfit.py:1081-1117 (floor) and :829-841 (drop)."""
    # a = 1(1), da = 0(0.01): y = [a+da, a-da]
    cov = np.array([[1 + 1e-4, 1 - 1e-4], [1 - 1e-4, 1 + 1e-4]])
    mean = np.array([1., 1.])
    sc = 0.01
    pdf0 = PDF.from_dense(mean, cov, svdcut=0.0)
    assert pdf0.nmod == 0
    np.testing.assert_allclose(pdf0.icov(), np.linalg.inv(cov), rtol=1e-9)
    pdf = PDF.from_dense(mean, cov, svdcut=sc)
    assert pdf.nmod == 1
    creg = np.linalg.inv(pdf.icov())
    # (y1-y0)/2 now has variance svdcut (lam_max ~ 2, corr ~ cov/1.0001)
    var_diff = np.array([-.5, .5]) @ creg @ np.array([-.5, .5])
    assert var_diff == pytest.approx(sc, rel=1e-3)
    assert pdf.logdet == pytest.approx(np.log(np.linalg.det(creg)))
    # negative cut: data = [(x+dx)/2, (x-dx)/20], x=1(1), dx=0.01(1)
    k = KAT['svd_negative']
    xm, xs = gvar_lite.parse(k['x'])
    dm, ds = gvar_lite.parse(k['dx'])
    A = np.array([[0.5, 0.5], [0.05, -0.05]])
    ymean = A @ np.array([xm, dm])
    ycov = A @ np.diag([xs ** 2, ds ** 2]) @ A.T
    pm, ps = gvar_lite.parse_array(k['prior'])
    fit = ofit.nonlinear_fit(False, ymean, ycov, lambda p: p, prior_mean=pm, prior_err=ps, svdcut=k['svdcut'])
    assert fit.dof == k['dof'] and fit.svdn == k['svdn']
    w = np.array([1., 10.])
    assert gvar_lite.fmt(w @ fit.pmean, np.sqrt(w @ fit.cov @ w)).startswith('1.0(1.0') or \
        '%.1f(%.1f)' % (w @ fit.pmean, np.sqrt(w @ fit.cov @ w)) == k['combo_fmt1']


def _p_corr_problem():
    k = KAT['p_corr']
    ym, ys = gvar_lite.parse_array(k['y'])
    x = np.array(k['x'])
    pcov = np.eye(4)
    pcov[1, 1] = 400. + 0.1 ** 2
    pcov[0, 1] = pcov[1, 0] = 20.
    fcn = lambda x, p: (p[0] * (x ** 2 + p[1] * x)) / (x ** 2 + x * p[2] + p[3])
    return x, ym, ys, fcn, np.zeros(4), pcov, k['out']


def test_p_corr_example():
    """examples/p-corr.py:44-61 vs examples/p-corr.out (correlated 2x2 prior block)."""
    x, ym, ys, fcn, pm, pcov, out = _p_corr_problem()
    fit = ofit.nonlinear_fit(x, ym, ys, fcn, prior_mean=pm, prior_err=pcov)
    check_header(fit, out)
    assert gvar_lite.fmt_array(fit.pmean, fit.psdev) == '[0.149(17) 2.97(34) 1.23(61) 0.59(15)]'
    corr01 = fit.cov[0, 1] / np.sqrt(fit.cov[0, 0] * fit.cov[1, 1])
    assert '%.4f' % corr01 == '0.9571'
    assert fit.nblocks == {1: 13, 2: 1} and fit.svdn == 0
    assert fit.stopping_criterion == 1


def _y_vs_x(nexp):
    k = KAT['y_vs_x']
    x = np.array(k['x'])
    pm = np.concatenate([np.full(nexp, 0.5), np.arange(1, nexp + 1.0)])
    ps = np.full(2 * nexp, 0.4)

    def fcn(x, p):
        return dual.stack_sum(p[i] * dual.exp(-p[nexp + i] * x) for i in range(nexp))
    return x, np.array(k['ymean']), np.array(k['ycov']), fcn, pm, ps


def test_y_vs_x_example():
    """examples/y-vs-x.py vs y-vs-x.out: dense 8x8 ycov, one SVD mode modified."""
    out = KAT['y_vs_x']['out']
    blocks = re.split(r'\*+ nexp = (\d+)\n', out)[1:]
    p0 = None
    expected_p = {
        2: '[0.4024(40) 0.4471(46) 0.90104(51) 1.8282(14)]',
        3: '[0.4019(40) 0.406(14) 0.61(36) 0.90039(54) 1.8026(82) 2.83(19)]',
    }
    for nexp_s, text in zip(blocks[0::2], blocks[1::2]):
        nexp = int(nexp_s)
        x, ym, ycov, fcn, pm, ps = _y_vs_x(nexp)
        q0 = None
        if p0 is not None:      # _unpack_p0: overlap of old p0 with the new prior layout
            q0 = ofit.default_p0(pm, ps)
            n_old = p0.size // 2
            q0[:n_old] = p0[:n_old]
            q0[nexp:nexp + n_old] = p0[n_old:]
        fit = ofit.nonlinear_fit(x, ym, ycov, fcn, prior_mean=pm, prior_err=ps, p0=q0)
        assert fit.svdn == 1 and fit.nblocks[8] == 1
        c, d, q, g = parse_header(text)
        assert fit.dof == d
        assert '%.2g' % (fit.chi2 / fit.dof) == c
        assert '%.5g' % fit.logGBF == g, (nexp, fit.logGBF, g)
        if nexp in expected_p:
            assert gvar_lite.fmt_array(fit.pmean, fit.psdev) == expected_p[nexp]
        if fit.chi2 / fit.dof < 1.:
            p0 = fit.pmean


def test_empbayes_example():
    """examples/empbayes.py / .out: logGBF maximised over the prior width."""
    from scipy.optimize import minimize
    k = KAT['empbayes']
    x = np.array(k['src_inputs']['x'])
    ym, ys = gvar_lite.parse_array(k['src_inputs']['y'])
    fcn = lambda x, p: dual.exp(-p[0] - p[1] * x - p[2] * x ** 2 - p[3] * x ** 3)
    last = dict(p0=None)

    def neg_logGBF(z):
        fit = ofit.nonlinear_fit(x, ym, ys, fcn, prior_mean=np.zeros(4), prior_err=np.full(4, abs(z[0])),
                                 p0=last['p0'])
        last['p0'] = fit.pmean
        last['fit'] = fit
        return -fit.logGBF
    res = minimize(neg_logGBF, [1.0], method='Nelder-Mead', tol=1e-4)
    fit = ofit.nonlinear_fit(x, ym, ys, fcn, prior_mean=np.zeros(4), prior_err=np.full(4, abs(res.x[0])),
                             p0=last['p0'])
    check_header(fit, k['out'])
    assert '%.1f' % abs(res.x[0]) == '5.3'
    assert gvar_lite.fmt_array(fit.pmean, fit.psdev) == '[2.5904(22) -6.530(22) 7.832(65) -1.688(55)]'


def test_simple_example_header():
    """examples/simple.py:28-41 vs simple.out: mixed 2x2 data blocks + scalar."""
    ymean = np.array([1.376, 2.010, 1.329, 1.582, 2.0])
    ycov = np.zeros((5, 5))
    ycov[:2, :2] = [[0.0047, 0.01], [0.01, 0.056]]
    ycov[2:4, 2:4] = [[0.0047, 0.0067], [0.0067, 0.0136]]
    ycov[4, 4] = 0.25
    x1, x2 = np.array([0.1, 1.0]), np.array([0.1, 0.5])

    def fcn(p):
        return dual.concatenate([dual.exp(p[0] + x1 * p[1]), dual.exp(p[0] + x2 * p[1]),
                                 (p[1] / p[0]).reshape(1) if isinstance(p, dual.Dual) else np.array([p[1] / p[0]])])
    fit = ofit.nonlinear_fit(False, ymean, ycov, fcn, prior_mean=[0.5, 0.5], prior_err=[0.5, 0.5])
    check_header(fit, KAT['simple']['out'])
    assert gvar_lite.fmt_array(fit.pmean, fit.psdev) == '[0.253(32) 0.449(65)]'
    assert fit.nblocks == {1: 3, 2: 2}


def _simple_fit():
    ymean = np.array([1.376, 2.010, 1.329, 1.582, 2.0])
    ycov = np.zeros((5, 5))
    ycov[:2, :2] = [[0.0047, 0.01], [0.01, 0.056]]
    ycov[2:4, 2:4] = [[0.0047, 0.0067], [0.0067, 0.0136]]
    ycov[4, 4] = 0.25
    x1, x2 = np.array([0.1, 1.0]), np.array([0.1, 0.5])

    def fcn(p):
        return dual.concatenate([dual.exp(p[0] + x1 * p[1]), dual.exp(p[0] + x2 * p[1]),
                                 (p[1] / p[0]).reshape(1) if isinstance(p, dual.Dual) else np.array([p[1] / p[0]])])
    fit = ofit.nonlinear_fit(False, ymean, ycov, fcn, prior_mean=[0.5, 0.5], prior_err=[0.5, 0.5])
    cov_in = np.zeros((7, 7))
    cov_in[:5, :5] = ycov
    cov_in[5, 5] = cov_in[6, 6] = 0.25
    return fit, cov_in


def parse_errorbudget(out):
    """'Partial % Errors' table of a reference .out file -> {(output, input): percent}."""
    lines = out.split('Partial % Errors:')[1].strip().splitlines()
    cols = lines[0].split()
    tab = {}
    for ln in lines[1:]:
        if ':' not in ln:
            continue
        name, vals = ln.split(':')
        for c, v in zip(cols, vals.split()):
            tab[c, name.strip()] = float(v)
    return tab


def test_simple_example_error_budget():
    """fit.p derivatives (f1): examples/simple.py:49-61 vs the 'Partial % Errors' table of
    simple.out:24-32 -- pins D = dp/d[y, prior] (src/lsqfit/__init__.py:897-911)."""
    fit, cov_in = _simple_fit()
    D = ofit.dp_dinputs(fit)
    # cov_p = D C D^T (doc/source/lsqfit.rst:112-113)
    np.testing.assert_allclose(D @ cov_in @ D.T, fit.cov, rtol=1e-9)
    a, b = fit.pmean
    grads = {'a': [1.0, 0.0], 'b/a': [-b / a ** 2, 1.0 / a], 'b': [0.0, 1.0]}
    vals = {'a': a, 'b/a': b / a, 'b': b}
    groups = {'y': [0, 1, 2, 3, 4], 'prior': [5, 6], 'total': list(range(7))}
    err = ofit.partial_sdev(D, grads, groups, cov_in)
    want = parse_errorbudget(KAT['simple']['out'])
    assert len(want) == 9
    for (g, name), pct in want.items():
        got = 100.0 * err[g, name] / abs(vals[g])
        assert '%.2f' % got == '%.2f' % pct, (g, name, got, pct)


def test_partialerr_weighted_average():
    """tests/test_lsqfit.py:1474-1510 (test_partialerr1): three equal measurements of p['y'],
    a wide prior on it and an unrelated prior: d p_y / d y_i = 1/ny, d p_noty / d prior = 1."""
    ny = 3
    fcn = lambda p: dual.concatenate([p[0].reshape(1)] * ny) if isinstance(p, dual.Dual) else np.full(ny, p[0])
    fit = ofit.nonlinear_fit(False, np.full(ny, 2.0), np.full(ny, 0.125), fcn,
                             prior_mean=[0.1, 3.0], prior_err=[1e4, 0.125])
    D = ofit.dp_dinputs(fit)
    np.testing.assert_allclose(D[0, :ny], 1.0 / ny, rtol=1e-6)
    np.testing.assert_allclose(D[1, ny + 1], 1.0, rtol=1e-12)
    cov_in = np.diag(np.array([0.125] * ny + [1e4, 0.125]) ** 2)
    err = ofit.partial_sdev(D, {'y': [1, 0], 'not y': [0, 1]},
                            {'y': [0, 1, 2], 'not y': [4], 'other prior': [3]}, cov_in)
    wavg_sdev = 0.125 / np.sqrt(ny)
    assert abs(err['y', 'y'] - wavg_sdev) < 1e-7
    assert err['y', 'not y'] == 0.0 and abs(err['y', 'other prior']) < 1e-5
    assert abs(err['not y', 'not y'] - 0.125) < 1e-12
    assert err['not y', 'y'] == 0.0 and err['not y', 'other prior'] == 0.0


def test_pdf_dchi2():
    """tests/test_lsqfit.py:415-454 (test_pdf_dchi2): for a linear fit dchi2(pmean + psdev) equals
    the number of parameters moved by one sigma... as asserted there."""
    ymean, ysd = np.array([1.5, 0.8, 12.0]), np.array([1.0, 0.5, 13.0])
    fcn = lambda p: dual.concatenate([p[0].reshape(1)] * 3) if isinstance(p, dual.Dual) else np.full(3, p[0])
    fit = ofit.nonlinear_fit(False, ymean, ysd, fcn, prior_mean=[0.0, 0.0], prior_err=[2.0, 5.0])
    assert abs(ofit.pdf(fit, fit.pmean) - 1.0) < 1e-7
    assert abs(ofit.dchi2(fit, fit.pmean)) < 1e-7
    p = fit.pmean + fit.psdev
    assert abs(ofit.dchi2(fit, p) - 2) < 1e-7               # fit.prior.size
    assert abs(ofit.pdf(fit, p) - np.exp(-ofit.dchi2(fit, p) / 2)) < 1e-12
    p[0] = fit.pmean[0]
    assert abs(ofit.dchi2(fit, p) - 1) < 1e-7
    # one-parameter case (:445-454)
    fit = ofit.nonlinear_fit(False, ymean, ysd, fcn, prior_mean=[0.0], prior_err=[2.0])
    assert abs(ofit.dchi2(fit, fit.pmean + fit.psdev) - 1) < 1e-7
    # batched (lbatch) layout agrees with the single-point call
    pts = fit.pmean + np.linspace(-2, 2, 7)[:, None] * fit.psdev
    np.testing.assert_allclose(ofit.dchi2(fit, pts), [ofit.dchi2(fit, q) for q in pts], rtol=0, atol=1e-12)
    np.testing.assert_allclose(ofit.dchi2(fit, pts), np.linspace(-2, 2, 7) ** 2, atol=1e-7)


def test_maxit0_and_unusual_cases():
    """tests/test_lsqfit.py:405-413 (maxit=0: the prior comes back untouched) and :455-470
    (scalar y and prior; two data points without x): closed-form weighted averages."""
    fcn = lambda p: dual.concatenate([p[0].reshape(1)] * 2) if isinstance(p, dual.Dual) else np.full(2, p[0])
    fit = ofit.nonlinear_fit(False, [1.5, 0.8], [1.0, 0.5], fcn, prior_mean=[0.0], prior_err=[2.0], maxit=0)
    np.testing.assert_allclose(fit.pmean, [0.0]); np.testing.assert_allclose(fit.psdev, [2.0])
    assert fit.nit == 0 and fit.error is None and fit.stopping_criterion == 0
    assert abs(fit.chi2 - (1.5 ** 2 + (0.8 / 0.5) ** 2)) < 1e-12
    fit = ofit.nonlinear_fit(False, [1.5, 0.8], [1.0, 0.5], fcn, p0=[0.0], maxit=0)
    np.testing.assert_allclose(fit.pmean, [0.0])
    assert np.all(np.isinf(fit.psdev)) and fit.logGBF is None

    def wavg(m, s):
        w = 1.0 / np.asarray(s, float) ** 2
        return float(np.sum(w * m) / np.sum(w)), float(np.sum(w) ** -0.5)
    ident = lambda p: p[:1] if isinstance(p, dual.Dual) else np.array(p[:1])
    fit = ofit.nonlinear_fit(False, [1.5], [0.1], ident, prior_mean=[2.0], prior_err=[0.5])
    m, s = wavg([1.5, 2.0], [0.1, 0.5])
    assert gvar_lite.fmt(fit.pmean[0], fit.psdev[0]) == gvar_lite.fmt(m, s)
    fit = ofit.nonlinear_fit(False, [1.5, 1.7], [0.1, 0.2], fcn, prior_mean=[2.0], prior_err=[0.5], tol=1e-8)
    m, s = wavg([1.5, 1.7, 2.0], [0.1, 0.2, 0.5])
    assert gvar_lite.fmt(fit.pmean[0], fit.psdev[0]) == gvar_lite.fmt(m, s)


def _x_err_inputs():
    k = KAT['x_err']
    xm, xs = gvar_lite.parse_array(k['x'])
    ym, ys = gvar_lite.parse_array(k['y'])
    bm, bs = gvar_lite.parse_array(k['prior_b'])
    return ym, ys, np.concatenate([bm, xm]), np.concatenate([bs, xs]), k['out']


def parse_parameter_table(out):
    """Mean(sdev) strings of the 'Parameters:' table of a reference .out file, in order."""
    body = out.split('Parameters:')[1].split('Settings:')[0]
    vals = []
    for ln in body.strip().splitlines():
        m = re.search(r'(-?[\d.]+(?:e[-+]?\d+)?) \((\d[\d.]*)\)\s+\[', ln)
        assert m, ln
        vals.append('%s(%s)' % (m.group(1), m.group(2)))
    return vals


def test_x_err_example():
    """examples/x-err.py:21-48 vs x-err.out: the x_i are parameters (priors = the measured x), so row
    i of the model depends on its own parameter; 19 parameters, 15 data points, 13 iterations."""
    ym, ys, pm, ps, out = _x_err_inputs()

    def fcn(p):
        b0, b1, b2, b3 = p[0], p[1], p[2], p[3]
        return b0 / ((1. + dual.exp(b1 - b2 * p[4:])) ** (1. / b3))
    fit = ofit.nonlinear_fit(False, ym, ys, fcn, prior_mean=pm, prior_err=ps)
    check_header(fit, out)
    want = parse_parameter_table(out)
    assert len(want) == 19
    got = [gvar_lite.fmt(m, s) for m, s in zip(fit.pmean, fit.psdev)]
    assert got == want
    assert fit.nit == 13                       # 'itns/time = 13/...' in x-err.out


def _noprior_data(seed=7):
    """tests/test_lsqfit.py:671-683: nine draws of 4.00(25), neighbours averaged -> 8 correlated values."""
    rng = np.random.default_rng(seed)
    y9 = 4.0 + 0.25 * rng.standard_normal(9)
    B = np.zeros((8, 9))
    for i in range(8):
        B[i, i] = B[i, i + 1] = 0.5
    return B @ y9, B @ (0.25 ** 2 * np.eye(9)) @ B.T


def test_noprior_zero_dof():
    """tests/test_lsqfit.py:671-713 (test_noprior): fcn = p**2, as many parameters as correlated data,
    no prior: dof 0, chi2 0, logGBF None, and the fit reproduces the data with its covariance."""
    ymean, ycov = _noprior_data()
    sq = lambda p: p * p
    fit = ofit.nonlinear_fit(False, ymean, ycov, sq, p0=np.full(8, 0.1), tol=1e-14)
    assert fit.logGBF is None and fit.dof == 0
    assert abs(fit.chi2) < 1e-4
    np.testing.assert_allclose(fit.pmean ** 2, ymean, rtol=1e-4)
    g = np.diag(2 * fit.pmean)
    np.testing.assert_allclose(g @ fit.cov @ g.T, ycov, rtol=1e-4, atol=1e-4 * np.abs(ycov).max())


def _udata_inputs():
    """tests/test_lsqfit.py:1152-1167: y = 1.01(1) * [1.000(1)] * 4 -- almost fully correlated."""
    m = np.full(4, 1.01)
    cov = np.full((4, 4), 0.01 ** 2) + np.diag(np.full(4, (1.01 * 0.001) ** 2))
    return m, cov


def test_uncorrelated_data_flag():
    """tests/test_lsqfit.py:1152-1167 (udata drops the correlations, src/lsqfit/__init__.py:1892-1893):
    same mean, half the error for four (nearly) fully correlated points."""
    m, cov = _udata_inputs()
    const = lambda p: dual.concatenate([p[0].reshape(1)] * 4) if isinstance(p, dual.Dual) else np.full(4, p[0])
    for kw in (dict(prior_mean=[1.0], prior_err=[1.0]), dict(p0=[1.0])):
        f1 = ofit.nonlinear_fit(False, m, cov, const, udata=True, **kw)
        f2 = ofit.nonlinear_fit(False, m, cov, const, **kw)
        assert abs(f1.pmean[0] - f2.pmean[0]) < 5e-4
        assert abs(2 * f1.psdev[0] - f2.psdev[0]) < 5e-4


def _svd_case(seed=11):
    """tests/test_lsqfit.py:773-790: two strongly correlated pairs, y ~ [1.1, 0.9] and p**2 likewise,
    error 1e-2 on the symmetric and 1e-4 on the antisymmetric combination."""
    fac = 100.
    sig1, sig2 = 1. / fac, 1e-2 / fac
    cov = sig1 ** 2 * np.array([[1., 1.], [1., 1.]]) + sig2 ** 2 * np.array([[1., -1.], [-1., 1.]])
    mean = np.array([1.1, 0.9])
    rng = np.random.default_rng(seed)
    L = np.linalg.cholesky(cov)
    y = mean + L @ rng.standard_normal(2)
    p2 = mean + L @ rng.standard_normal(2)
    pm = np.sqrt(p2)
    Jp = np.diag(0.5 / pm)
    return y, cov, pm, Jp @ cov @ Jp.T, sig1, sig2


def check_svd_fit(fit, svdcut, sig1, sig2):
    G = np.array([[fit.pmean[0], fit.pmean[1]], [fit.pmean[0], -fit.pmean[1]]])   # d[(p0^2 +- p1^2)/2]/dp
    c = G @ fit.cov @ G.T
    sd = np.sqrt(np.diag(c))
    s2 = max(sd[0] * sig2 / sig1, svdcut ** 0.5 * sd[0])
    assert abs(sd[1] / s2 - 1.) < 0.005                       # assertAlmostEqual(..., places=2)
    assert fit.svdn == (0 if svdcut < 1e-10 else 2)
    assert fit.nblocks[2] == 2


@pytest.mark.parametrize('svdcut', [1e-20, 1e-2])
def test_svd_cut_floor(svdcut):
    """tests/test_lsqfit.py:773-826 (cases 1, 2, 5, 6): with svdcut = 1e-2 the error of the
    antisymmetric combination is set by the eigenvalue floor, sqrt(svdcut) times the symmetric one;
    svdn counts the two modified modes."""
    y, ycov, pm, pcov, sig1, sig2 = _svd_case()
    sq = lambda p: p * p
    fit = ofit.nonlinear_fit(False, y, ycov, sq, prior_mean=pm, prior_err=pcov, svdcut=svdcut)
    check_svd_fit(fit, svdcut, sig1, sig2)
    # the symmetric combination is (to 1 %) the weighted average of data and prior
    ans_p = 0.5 * (fit.pmean[0] ** 2 + fit.pmean[1] ** 2)
    ans_y, ans_pr = 0.5 * (y[0] + y[1]), 0.5 * (pm[0] ** 2 + pm[1] ** 2)
    assert abs(ans_p / (0.5 * (ans_y + ans_pr)) - 1) < 1e-2


EMPBAYES_Y = ['0.5351(54)', '0.6762(67)', '0.9227(91)', '1.3803(131)', '4.0145(399)']    # test_lsqfit.py:873-877
EMPBAYES_X = [0.1, 0.3, 0.5, 0.7, 0.95]


def test_empbayes_polynomial_prior_width():
    """tests/test_lsqfit.py:871-887 (test_empbayes): 25-coefficient polynomial, priors 0 +- exp(z);
    maximising logGBF over z gives exp(z) = 0.6012 to one decimal place."""
    from scipy.optimize import minimize
    ym, ys = gvar_lite.parse_array(EMPBAYES_Y)
    x = np.array(EMPBAYES_X)
    V = x[:, None] ** np.arange(25)[None, :]
    last = dict(p0=None)

    def neg(z):
        w = float(np.exp(z[0]))
        fit = ofit.nonlinear_fit(x, ym, ys, lambda x, p: V @ p, prior_mean=np.zeros(25), prior_err=np.full(25, w),
                                 jac=lambda x, p: V, p0=last['p0'])
        last['p0'] = fit.pmean
        return -fit.logGBF
    res = minimize(neg, [np.log(0.7)], method='Nelder-Mead', tol=1e-3)
    assert abs(np.exp(res.x[0]) - 0.6012) < 0.05


NORMAL_Y = ['-0.17(20)', '-0.03(20)', '-0.39(20)', '0.10(20)', '-0.03(20)', '0.06(20)', '-0.23(20)', '-0.23(20)',
            '-0.15(20)', '-0.01(20)', '-0.12(20)', '0.05(20)', '-0.09(20)', '-0.36(20)', '0.09(20)', '-0.07(20)',
            '-0.31(20)', '0.12(20)', '0.11(20)', '0.13(20)']          # tests/test_lsqfit.py:1581-1586

# (expression of a in terms of the fitted parameter u, prior mean/sdev of u given a = 0.02(2),
#  da/du, expected fit.p['a'].fmt())  -- tests/test_lsqfit.py:1579-1640
TRANSFORMED_PRIOR_CASES = {
    'normal': (lambda u: u, 0.02, 0.02, lambda u: 1.0, '0.004(18)'),
    'lognormal': (lambda u: dual.exp(u) if isinstance(u, dual.Dual) else np.exp(u), np.log(0.02), 0.02 / 0.02,
                  lambda u: np.exp(u), '0.012(11)'),
    'sqrtnormal': (lambda u: u * u, np.sqrt(0.02), 0.02 / (2 * np.sqrt(0.02)), lambda u: 2 * u, '0.010(13)'),
}


@pytest.mark.parametrize('kind', sorted(TRANSFORMED_PRIOR_CASES))
def test_transformed_priors(kind):
    """tests/test_lsqfit.py:1579-1640 (test_normal / test_lognormal / test_sqrtnormal): the parameter is
    u = a, log(a) or sqrt(a) with the prior of a = 0.02(2) pushed through; fit.p['a'] = a(u) is printed."""
    a_of, um, us, da, want = TRANSFORMED_PRIOR_CASES[kind]
    ym, ys = gvar_lite.parse_array(NORMAL_Y)

    def fcn(p):
        a = a_of(p[0])
        return dual.concatenate([a.reshape(1)] * ym.size) if isinstance(a, dual.Dual) else np.full(ym.size, a)
    fit = ofit.nonlinear_fit(False, ym, ys, fcn, prior_mean=[um], prior_err=[us])
    u = fit.pmean[0]
    a = a_of(u)
    assert gvar_lite.fmt(float(a), abs(da(u)) * fit.psdev[0]) == want


def test_lognormal_array_case():
    """tests/test_lsqfit.py:1613-1622: y = [[0.1(1), 1(1)], [0.2(1), 2(1)]], log(a) = log([0.1(2), 10(20)])."""
    ym = np.array([0.1, 1.0, 0.2, 2.0])
    ys = np.array([0.1, 1.0, 0.1, 1.0])
    um, us = np.log([0.1, 10.0]), np.array([2.0, 2.0])

    def fcn(p):
        a = dual.exp(p) if isinstance(p, dual.Dual) else np.exp(p)
        return dual.concatenate([a, a]) if isinstance(a, dual.Dual) else np.concatenate([a, a])
    fit = ofit.nonlinear_fit(False, ym, ys, fcn, prior_mean=um, prior_err=us)
    a = np.exp(fit.pmean)
    assert [gvar_lite.fmt(a[i], a[i] * fit.psdev[i]) for i in range(2)] == ['0.147(69)', '1.64(69)']


BASICFIT_YCOV = np.array([[2., .25], [.25, 4.]])
BASICFIT_PCOV = np.array([[2., .5], [.5, 1.]])


def basicfit_checks(fit, yfac, pfac, pmean, cov):
    """tests/test_lsqfit.py:126-168 (t_basicfit): prior-dominated and data-dominated extremes."""
    assert fit.dof == 2 and abs(fit.Q - 1.0) < 5e-8 and abs(fit.chi2) < 5e-8
    rel = lambda a, b: np.max(np.abs(np.asarray(a) - np.asarray(b))) / np.max(np.abs(np.concatenate([np.ravel(a), np.ravel(b)])))
    if yfac > 100 * pfac:
        assert rel(pmean, [4., 16.]) < 1e-5
        assert rel(cov, BASICFIT_PCOV * pfac) < 1e-5
    else:
        assert rel(np.asarray(pmean) ** 2, [1., 4.]) < 1e-5
        Jsq = np.diag(2 * np.asarray(pmean))                 # cov of p**2 by linear propagation
        assert rel(Jsq @ cov @ Jsq.T, BASICFIT_YCOV * yfac) < 1e-5


@pytest.mark.parametrize('yfac,pfac', [(1e22, 1.0), (1.0, 1e22)])
def test_basicfit_extremes(yfac, pfac):
    fit = ofit.nonlinear_fit(False, np.array([1., 4.]), BASICFIT_YCOV * yfac, lambda p: p ** 2,
                             prior_mean=np.array([4., 16.]), prior_err=BASICFIT_PCOV * pfac)
    basicfit_checks(fit, yfac, pfac, fit.pmean, fit.cov)


LINEAR_DATA = ['2.0008(10)', '1.72452(86)', '1.49030(75)', '1.29009(65)', '1.12017(57)']


def linear_case_fcn(p, t=np.arange(0., 1., 0.2)):
    """tests/test_lsqfit.py:1663-1670: c = p[:2], E = cumsum(p[2:]), sum_i c_i exp(-E_i t)."""
    c = p[:2]
    E0, E1 = p[2], p[2] + p[3]
    return c[0] * dual.exp(-E0 * t) + c[1] * dual.exp(-E1 * t)


def test_linear_array_case():
    """tests/test_lsqfit.py:1663-1682 (and :1642-1661 with dict keys): linear=[0, 1] gives the same
    chi2 (7 places), the same means (rtol 1e-5) and errors as the plain fit."""
    ym, ys = gvar_lite.parse_array(LINEAR_DATA)
    pm, ps = gvar_lite.parse_array(2 * ['1.00(1)'] + 2 * ['0.500(1)'])
    fita = ofit.nonlinear_fit(False, ym, ys, linear_case_fcn, prior_mean=pm, prior_err=ps)
    fitb = ofit.nonlinear_fit(False, ym, ys, linear_case_fcn, prior_mean=pm, prior_err=ps, linear=[0, 1])
    assert abs(fita.chi2 - fitb.chi2) < 5e-8
    np.testing.assert_allclose(fita.pmean, fitb.pmean, rtol=1e-5)
    np.testing.assert_allclose(fita.cov, fitb.cov, rtol=1e-4, atol=1e-12)
    # all parameters linear: one exact step (the reference skips the fitter altogether, :765-781)
    M = np.array([[1., 1.], [1., -2.], [3., 0.]])
    kw = dict(prior_mean=np.zeros(2), prior_err=np.ones(2) * 10, jac=lambda p: M)
    f1 = ofit.nonlinear_fit(False, np.array([1., 2., 3.]), np.ones(3), lambda p: M @ p, **kw)
    f2 = ofit.nonlinear_fit(False, np.array([1., 2., 3.]), np.ones(3), lambda p: M @ p, linear=[0, 1], **kw)
    np.testing.assert_allclose(f1.pmean, f2.pmean, rtol=1e-9)
    assert f2.nit <= 2 and abs(f1.chi2 - f2.chi2) < 1e-10


def y_noerr_fcn(x, p):
    n = p.size // 2
    return dual.stack_sum(p[i] * dual.exp(-p[n + i] * x) for i in range(n))


def test_y_noerr_out_data_prior_correlations():
    """examples/y-noerr.out: marginalising 100 - nexp exponentials into the data correlates the data
    with the fit prior (E = cumsum(dE)); concat(y, prior) is ONE dense block whose correlation matrix
    has 2-3 modes on the svdcut floor.  Every printed parameter string, chi2/dof, Q, svdn and (nexp <=
    4; the last one sits on the floor's rounding) logGBF to 4-5 digits; iteration counts within 15 %."""
    from tests.helpers import y_noerr_expected, y_noerr_joint
    k = KAT['y_noerr']
    exp = y_noerr_expected(k)
    assert len(exp) == 5
    p0 = None
    for nexp in range(1, 6):
        x, mean, cov = y_noerr_joint(k, nexp)
        P, n = 2 * nexp, len(k['x'])
        extra = [((i, n + j), cov[i, n + j]) for i in range(n) for j in range(P) if cov[i, n + j] != 0.0]
        assert extra                                                   # the cross-correlations are there
        if p0 is not None:                                             # p0 = previous fit.pmean (:46) + prior means
            p0 = np.concatenate([p0[:nexp - 1], [mean[n + nexp - 1]], p0[nexp - 1:], [mean[n + P - 1]]])
        fit = ofit.nonlinear_fit(x, mean[:n], cov[:n, :n], y_noerr_fcn, prior_mean=mean[n:], prior_err=cov[n:, n:],
                                 p0=p0, tol=k['tol'], svdcut=k['svdcut'], extra_cov=extra)
        e = exp[nexp - 1]
        assert fit.dof == e['dof'] and fit.svdn == e['svdn']
        assert '%.2g' % (fit.chi2 / fit.dof) == e['chi2dof'] and '%.2g' % fit.Q == e['Q']
        assert [gvar_lite.fmt(m, s) for m, s in zip(fit.pmean, fit.psdev)] == e['pars']
        if nexp <= 4:
            assert abs(fit.logGBF - e['logGBF']) < 2e-3
        assert abs(fit.nit - e['nit']) <= max(2, 0.15 * e['nit'])
        p0 = fit.pmean


WAVG_SVD_COV = np.array([[.5, .25, .5], [.25, .5, .5], [.5, .5, 1.]])     # cov of [(a+b)/2, (a+c)/2, a], a, b, c = 1(1)


def test_wavg_svd_literal():
    """tests/test_lsqfit.py:581-588 (test_wavg_svd): the weighted average of three correlated values is a
    one-parameter fit without a prior; svdcut = 1 - 1e-16 floors two of the three correlation
    eigenvalues -> var 0.4561552812808828; a tiny svdcut leaves 1/3."""
    for svdcut, var, nmod in ((1 - 1e-16, 0.4561552812808828, 2), (1e-18, 1. / 3., 0)):
        fit = ofit.nonlinear_fit(False, np.ones(3), WAVG_SVD_COV, lambda p: p[0] * np.ones(3), p0=[1.0],
                                 jac=lambda p: np.ones((3, 1)), svdcut=svdcut)
        assert abs(fit.cov[0, 0] - var) < 5e-8 and fit.svdn == nmod


def test_unusual_cases_print_as_weighted_averages():
    """tests/test_lsqfit.py:456-472: scalar y + scalar prior with fcn(p) = p, and y of two elements with fcn(p) = [p, p]:
    str(fit.p) == str(wavg(inputs))."""
    import numpy as np
    from oracle import fit as ofit, gvar_lite
    for ys, ysd, want in (([1.5], [0.1], '1.519(98)'), ([1.5, 1.7], [0.1, 0.2], '1.554(88)')):
        y, sd = np.array(ys), np.array(ysd)
        ref = ofit.nonlinear_fit(False, y, sd, lambda p: y.size * [p[0]], prior_mean=[2.0], prior_err=[0.5], tol=1e-8,
                                 jac=lambda p: np.ones((y.size, 1)))
        w = np.concatenate([1.0 / sd ** 2, [4.0]])
        mean, sdev = np.sum(w * np.concatenate([y, [2.0]])) / np.sum(w), 1.0 / np.sqrt(np.sum(w))
        assert gvar_lite.fmt(ref.pmean[0], ref.psdev[0]) == gvar_lite.fmt(mean, sdev) == want
