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
