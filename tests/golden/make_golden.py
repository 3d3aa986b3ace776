#!/usr/bin/env python3
"""Generate tests/golden/*.json from the DATA the reference ships.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

It transcribes inputs and expected outputs -- never reference code:

  nist.json   <- examples/nist/*.txt  (NIST StRD data, start values, certified
                 values, model formula text) + the prior/p0/expected strings of
                 examples/nist.py and the summary lines of examples/nist.out
  kat.json    <- literal known-answer values from tests/test_lsqfit.py and the
                 example inputs/outputs of examples/{p-corr,y-vs-x,simple,
                 empbayes}.{py,out}

The reference itself cannot be imported here (gvar and GSL are absent), so no
vector is *computed* by the reference; everything is copied from its files.
"""
import json
import os
import re

REF = '/root/reference'
OUT = os.path.dirname(os.path.abspath(__file__))

FLOAT = r'[-+]?(?:\d+\.?\d*|\.\d+)(?:[eE][-+]?\d+)?'


def parse_nist_txt(path):
    txt = open(path).read()
    lines = txt.split('\n')
    d = {}
    d['name'] = re.search(r'Dataset Name:\s+(\S+)', txt).group(1)
    # model formula: between 'Model:' and 'Starting'
    m0 = next(i for i, l in enumerate(lines) if l.startswith('Model:'))
    m1 = next(i for i, l in enumerate(lines) if 'Starting' in l and i > m0)
    block = lines[m0:m1]
    nparam = int(re.search(r'(\d+)\s+Parameters', '\n'.join(block)).group(1))
    formula_lines = [l.strip() for l in block[2:] if l.strip() and 'Parameters' not in l]
    formula = ' '.join(formula_lines)
    d['formula_text'] = formula
    lhs, rhs = formula.split('=', 1)
    rhs = re.sub(r'\+\s*e\s*$', '', rhs.strip()).strip()
    rhs = rhs.replace('[', '(').replace(']', ')')
    d['lhs'] = lhs.strip().replace('[', '(').replace(']', ')')
    d['expr'] = rhs
    d['nparam'] = nparam
    # starting / certified values
    start1, start2, cert, certsd = [], [], [], []
    for l in lines[m1:]:
        m = re.match(r'\s*b(\d+)\s*=\s*(%s)\s+(%s)\s+(%s)\s+(%s)' % (FLOAT, FLOAT, FLOAT, FLOAT), l)
        if m:
            start1.append(float(m.group(2)))
            start2.append(float(m.group(3)))
            cert.append(float(m.group(4)))
            certsd.append(float(m.group(5)))
    assert len(cert) == nparam, (path, len(cert), nparam)
    d.update(start1=start1, start2=start2, certified=cert, certified_sd=certsd)
    d['rss'] = float(re.search(r'Residual Sum of Squares:\s+(%s)' % FLOAT, txt).group(1))
    d['rsd'] = float(re.search(r'Residual Standard Deviation:\s+(%s)' % FLOAT, txt).group(1))
    d['dof'] = int(re.search(r'Degrees of Freedom:\s+(\d+)', txt).group(1))
    d['nobs'] = int(re.search(r'Number of Observations:\s+(\d+)', txt).group(1))
    # data
    di = max(i for i, l in enumerate(lines) if re.match(r'^Data:\s+y', l))
    cols = lines[di].split()[1:]
    rows = []
    for l in lines[di + 1:]:
        parts = l.split()
        if len(parts) == len(cols):
            try:
                rows.append([float(p) for p in parts])
            except ValueError:
                pass
    assert len(rows) == d['nobs'], (path, len(rows), d['nobs'])
    d['columns'] = cols
    d['data'] = rows
    return d


def parse_nist_py():
    src = open(os.path.join(REF, 'examples/nist.py')).read()
    out = {}
    for m in re.finditer(r'\ndef (\w+)\(\):\n(.*?)(?=\ndef |\Z)', src, re.S):
        name, body = m.group(1), m.group(2)
        if 'nonlinear_fit' not in body or name in ('main', 'assert_equal'):
            continue
        pri = re.search(r'prior = gv\.gvar\(\[(.*?)\]\)', body, re.S).group(1)
        priors = re.findall(r"'([^']*)'", pri)
        p0 = re.search(r'p0 = np\.array\(\[(.*?)\]\)', body, re.S).group(1)
        p0 = [float(v) for v in re.findall(FLOAT, p0)]
        yerr = re.search(r"\['0 \+- (%s)'\]" % FLOAT, body).group(1)
        exp = [e for e in re.findall(r"\n    assert_equal\(fit\.p, '([^']*)'\)", body)][-1]
        tol = float(re.search(r'tol=(%s)' % FLOAT, body).group(1))
        out[name] = dict(prior=priors, p0=p0, yerr=float(yerr), expected_p=exp, tol=tol,
                         log_y='y = log(y)' in body)
    return out


def parse_nist_out():
    txt = open(os.path.join(REF, 'examples/nist.out')).read()
    out = {}
    for m in re.finditer(r'=+ (\w+)\n(.*?)(?=\n=+ \w+\n|\Z)', txt, re.S):
        name, body = m.group(1), m.group(2)
        h = re.search(r'chi2/dof \[dof\] = (\S+) \[(\d+)\]\s+Q = (\S+)\s+logGBF = (\S+)', body)
        it = re.search(r'itns/time = (\d+)(\*?)/', body)
        out[name] = dict(chi2_dof=h.group(1), dof=int(h.group(2)), Q=h.group(3),
                         logGBF=h.group(4), itns=int(it.group(1)),
                         tol_line=re.search(r'tol = (\([^)]*\))', body).group(1))
    return out


def make_nist():
    py = parse_nist_py()
    outs = parse_nist_out()
    res = {}
    for fn in sorted(os.listdir(os.path.join(REF, 'examples/nist'))):
        if not fn.endswith('.txt'):
            continue
        d = parse_nist_txt(os.path.join(REF, 'examples/nist', fn))
        key = fn[:-4]
        d['harness'] = py[key]
        d['lsqfit_out'] = outs[key]
        res[key] = d
    json.dump(res, open(os.path.join(OUT, 'nist.json'), 'w'), indent=0)
    print('nist.json:', len(res), 'problems')


def make_kat():
    kat = {}
    # tests/test_lsqfit.py:1811-1833 (test_fitters)
    kat['test_fitters'] = dict(data=['0.9(1)', '2.2(2)'], prior=['1.0(5)', '2.0(5)'],
                               expected_p='[0.904(98) 2.17(19)]')
    # tests/test_lsqfit.py:1887-1894 (test_gammaQ): (a, x, Q(a,x), Q(x,a)), rtol 1e-2
    kat['gammaQ'] = [
        [2.371, 5.243, 0.05371580082389009, 0.9266599665892222],
        [20.12, 20.3, 0.4544782602230986, 0.4864172139106905],
        [100.1, 105.2, 0.29649013488390663, 0.6818457585776236],
        [1004., 1006., 0.4706659307021259, 0.5209695379094582],
    ]
    # tests/test_lsqfit.py:1700-1725 (test_gsl_multifit): f=(x-x*)^2+(x-x*)^4
    kat['gsl_multifit'] = dict(xans=[1., 2., 3.], cases=[
        dict(x0=[1., 1., 1.], alg='lm', tol=[1e-10, 0.0, 0.0], stopping_criterion=1, rtol=1e-3),
        dict(x0=[0., 0., 0.], alg='lmaccel', tol=[0.0, 1e-10, 0.0], stopping_criterion=2, rtol=1e-3),
        dict(x0=[0., 0., 0.], alg='subspace2D', tol=[1e-10, 0.0, 0.0], stopping_criterion=1, rtol=1e-3)])
    # tests/test_lsqfit.py:257-283 (test_format case 1): header + parameter line
    kat['format1'] = dict(y=[[1.5, 1.0], [0.8, 0.5]], prior=[[0.0, 2.0]], svdcut=1e-15,
                          tol=[1e-15, 1e-15, 1e-15],
                          header='chi2/dof [dof] = 0.3 [2]    Q = 0.74    logGBF = -2.9682',
                          p='0.90 (44)')
    # tests/test_lsqfit.py:845-868 (test_logGBF): closed form, fixed y instead of a random draw
    kat['logGBF'] = dict(yg=['2(1)', '4(6)', '-0.62(1)', '-100(10)'])
    # tests/test_lsqfit.py:955-962, :1024-1032 (test_unpack_data literal weights)
    kat['unpack_case2'] = dict(y=[[1, 2], [10, 4]], prior=[[2, 4]],
                               idx=[0, 1, 2], wgts=[0.5, 0.25, 0.25])
    kat['unpack_case4'] = dict(y=[[1, 2], [10, 4]], prior=[[1, 2], [1, 4]],
                               idx=[0, 1, 2, 3], wgts=[0.5, 0.25, 0.5, 0.25])
    # tests/test_lsqfit.py:829-841 (negative svdcut)
    kat['svd_negative'] = dict(x='1(1)', dx='0.01(1)', prior=['1(10)', '0.05(50)'],
                               svdcut=-0.2 ** 2, dof=1, svdn=1, combo_fmt1='1.0(1.0)')
    # examples/p-corr.py:44-61 and examples/p-corr.out
    kat['p_corr'] = dict(
        x=[4., 2., 1., 0.5, 0.25, 0.167, 0.125, 0.1, 0.0833, 0.0714, 0.0625],
        y=['0.198(14)', '0.216(15)', '0.184(23)', '0.156(44)', '0.099(49)', '0.142(40)',
           '0.108(32)', '0.065(26)', '0.044(22)', '0.041(19)', '0.044(16)'],
        prior_note='p=gvar(4*["0(1)"]); p[1]=20*p[0]+gvar("0.0(1)")',
        expr='(b1*(x**2+b2*x))/(x**2+x*b3+b4)',
        out=open(os.path.join(REF, 'examples/p-corr.out')).read())
    # examples/y-vs-x.py:58-99 and examples/y-vs-x.out
    src = open(os.path.join(REF, 'examples/y-vs-x.py')).read()
    x = re.search(r'x = np\.array\(\[(.*?)\]\)', src, re.S).group(1)
    ym = re.search(r'ymean = np\.array\(\s*\[(.*?)\]\s*\)', src, re.S).group(1)
    yc = re.search(r'ycov = np\.array\(\s*\[(.*?)\]\]\s*\)', src, re.S).group(1)
    xs = [float(v) for v in re.findall(FLOAT, x)]
    yms = [float(v) for v in re.findall(FLOAT, ym)]
    ycs = [float(v) for v in re.findall(FLOAT, yc)]
    assert len(xs) == 8 and len(yms) == 8 and len(ycs) == 64
    kat['y_vs_x'] = dict(x=xs, ymean=yms, ycov=[ycs[i * 8:(i + 1) * 8] for i in range(8)],
                         prior_note='a[i]=0.5(4), E[i]=(i+1)(0.4)  (examples/y-vs-x.py:63-67)',
                         out=open(os.path.join(REF, 'examples/y-vs-x.out')).read())
    # examples/empbayes.py / .out
    kat['empbayes'] = dict(src_inputs=_grab(os.path.join(REF, 'examples/empbayes.py')),
                           out=open(os.path.join(REF, 'examples/empbayes.out')).read())
    # examples/simple.py / .out
    kat['simple'] = dict(out=open(os.path.join(REF, 'examples/simple.out')).read())
    # examples/x-err.py:21-42 / x-err.out ("x has error bars": the x_i are fit parameters with the
    # measured x as their priors; 4 + 15 parameters): data literals are lines 22-26 and 27-31
    src = open(os.path.join(REF, 'examples/x-err.py')).read()
    lists = re.findall(r"gv\.gvar\(\[(.*?)\]\s*\)", src, flags=re.S)
    strs = [re.findall(r"'([^']+)'", l) for l in lists]
    kat['x_err'] = dict(x=strs[0], y=strs[1], prior_b=strs[2],
                        out=open(os.path.join(REF, 'examples/x-err.out')).read())
    # examples/y-noerr.py:82-90 / y-noerr.out ("y has no error bars": 100 - nexp exponentials of a
    # 100-term prior are marginalised into the data, which makes data and fit prior CORRELATED
    # through E = cumsum(dE)): data literals only; the prior is a[i] = 0.5(5), dE[i] = 1.0(1) (:74-79)
    src = open(os.path.join(REF, 'examples/y-noerr.py')).read()
    body = src[src.index('def make_data'):]
    xs = [float(v) for v in re.findall(FLOAT, re.search(r'x = np\.array\(\[(.*?)\]\)', body, re.S).group(1))]
    ys = [float(v) for v in re.findall(FLOAT, re.search(r'y = np\.array\(\[(.*?)\]\)', body, re.S).group(1))]
    assert len(xs) == 9 and len(ys) == 9
    kat['y_noerr'] = dict(x=xs, y=ys, n_terms=100, prior_a='0.5(5)', prior_dE='1.0(1)', tol=1e-15, svdcut=1e-12,
                          out=open(os.path.join(REF, 'examples/y-noerr.out')).read())
    json.dump(kat, open(os.path.join(OUT, 'kat.json'), 'w'), indent=1)
    print('kat.json:', len(kat), 'entries')


def _grab(path):
    """Numeric literals of the example's data statements (inputs only)."""
    src = open(path).read()
    out = {}
    for m in re.finditer(r'\n\s*(\w+) = (?:np\.array|gv\.gvar)\(\s*\[(.*?)\]\s*\)', src, re.S):
        body = m.group(2)
        strs = re.findall(r"'([^']*)'", body)
        out[m.group(1)] = strs if strs else [float(v) for v in re.findall(FLOAT, body)]
    return out


if __name__ == '__main__':
    make_nist()
    make_kat()
