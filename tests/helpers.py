"""Shared fixture loaders for the test-suite (tests only)."""
import json
import os

import numpy as np

from oracle import dual, gvar_lite

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def nist_problem(name, nist=None):
    """-> dict(x, y, ysd, fcn(x,p), prior_mean, prior_sd, p0, ...)."""
    d = (nist or load('nist.json'))[name]
    cols = d['columns']
    data = np.array(d['data'], float)
    y = data[:, 0]
    if d['lhs'].startswith('log'):
        y = np.log(y)
    xs = {c: data[:, i] for i, c in enumerate(cols) if i > 0}
    expr = d['expr']
    code = compile(expr, '<nist:%s>' % name, 'eval')
    P = d['nparam']

    def fcn(x, p):
        ns = dict(dual.NAMESPACE)
        ns.update(x)
        for k in range(P):
            ns['b%d' % (k + 1)] = p[k]
        return eval(code, {'__builtins__': {}}, ns)

    h = d['harness']
    pm, ps = gvar_lite.parse_array(h['prior'])
    return dict(name=name, x=xs, y=y, ysd=np.full(y.size, h['yerr']), fcn=fcn, P=P,
                prior_mean=pm, prior_sd=ps, p0=np.array(h['p0'], float), tol=h['tol'],
                expected_p=h['expected_p'], out=d['lsqfit_out'], certified=np.array(d['certified']),
                certified_sd=np.array(d['certified_sd']), rss=d['rss'], rsd=d['rsd'],
                dof=d['dof'], start1=np.array(d['start1']), start2=np.array(d['start2']),
                expr=expr, columns=cols)


def y_noerr_joint(k, nexp):
    """examples/y-noerr.py:27-36: mean and covariance of z = [ymod (9); a[:nexp]; E[:nexp]] where
    ymod = y - sum_{k >= nexp} a_k exp(-E_k x), E = cumsum(dE), by first-order propagation (what gvar
    does) from the 2 x n_terms independent primaries a_k = 0.5(5), dE_k = 1.0(1)."""
    from oracle import gvar_lite
    x, y, NT = np.array(k['x']), np.array(k['y']), int(k['n_terms'])
    am, asd = gvar_lite.parse(k['prior_a'])
    dm, dsd = gvar_lite.parse(k['prior_dE'])
    abar, Ebar = np.full(NT, am), np.cumsum(np.full(NT, dm))
    n = x.size
    M = np.zeros((n + 2 * nexp, 2 * NT))
    ex = np.exp(-np.outer(x, Ebar))
    M[:n, nexp:NT] = -ex[:, nexp:]                                   # d ymod / d a_k
    for i in range(NT):                                              # d ymod / d dE_i
        ks = np.arange(max(i, nexp), NT)
        M[:n, NT + i] = (abar[ks] * x[:, None] * ex[:, ks]).sum(1)
    for j in range(nexp):
        M[n + j, j] = 1.0
        M[n + nexp + j, NT:NT + j + 1] = 1.0
    var = np.concatenate([np.full(NT, asd ** 2), np.full(NT, dsd ** 2)])
    cov = (M * var) @ M.T
    mean = np.concatenate([y - (abar[nexp:] * ex[:, nexp:]).sum(1), abar[:nexp], Ebar[:nexp]])
    return x, mean, cov


def y_noerr_expected(k):
    """Per nexp block of examples/y-noerr.out: (chi2/dof string, dof, Q, logGBF, parameter strings, svdn)."""
    import re
    out = []
    for blk in k['out'].split('*' * 37)[1:]:
        m = re.search(r'chi2/dof \[dof\] = (\S+) \[(\d+)\]\s+Q = (\S+)\s+logGBF = (\S+)', blk)
        pars = re.findall(r'^\s+(?:[aE] )?\d+\s+(\S+ \(\d+\))\s+\[', blk, flags=re.M)
        svdn = int(re.search(r'svdcut/n = \S+?/(\d+)', blk).group(1))
        nit = int(re.search(r'itns/time = (\d+)/', blk).group(1))
        out.append(dict(chi2dof=m.group(1), dof=int(m.group(2)), Q=m.group(3), logGBF=float(m.group(4)),
                        pars=[p.replace(' ', '') for p in pars], svdn=svdn, nit=nit))
    return out
