"""Minimal restatement of gvar's number <-> string conventions (ORACLE ONLY).

gvar is a third-party dependency of the reference (setup.cfg:21,
``gvar>=13.1.5``) and is NOT under /root/reference; the rules below are
restated from its published behaviour and are anchored on the literal
strings the reference's own tests/examples hold, e.g.
``'[0.904(98) 2.17(19)]'`` (tests/test_lsqfit.py:1833) and
``'[238.9(2.7) 0.0005502(73)]'`` (examples/nist.py:120).
"""
import math
import re

import numpy as np

_PAREN = re.compile(r'^\s*([-+]?[0-9]*\.?[0-9]*(?:[eE][-+]?[0-9]+)?)\s*\(\s*([0-9]*\.?[0-9]*)\s*\)\s*(?:[eE]([-+]?[0-9]+))?\s*$')
_PM = re.compile(r'^\s*(\S+?)\s*(?:\+-|\+/-|±)\s*(\S+)\s*$')


def parse(s):
    """'1.23(45)' | '0(47788)' | '0.00(11)' | '0 +- 1.0e-1' -> (mean, sdev)."""
    if not isinstance(s, str):
        m, sd = s
        return float(m), float(sd)
    m = _PM.match(s)
    if m:
        return float(m.group(1)), float(m.group(2))
    m = _PAREN.match(s)
    if not m:
        raise ValueError('cannot parse gvar string: %r' % (s,))
    ms, ss, es = m.group(1), m.group(2), m.group(3)
    mantissa = ms.lower().split('e')[0]
    mean = float(ms)
    sdev = float(ss)
    if '.' not in ss and '.' in mantissa:
        ndec = len(mantissa.split('.')[1])
        sdev *= 10.0 ** (-ndec)
    if 'e' in ms.lower():
        sdev *= 10.0 ** int(ms.lower().split('e')[1])
    if es is not None:
        f = 10.0 ** int(es)
        mean *= f
        sdev *= f
    return mean, sdev


def parse_array(strs):
    a = np.array([parse(s) for s in strs], float)
    return a[:, 0].copy(), a[:, 1].copy()


def _ndec(x, offset=2):
    return int(math.floor(-math.log10(x))) + offset


def fmt(mean, sdev):
    """Default ``str(GVar)``: two significant digits of the error."""
    v, dv = float(mean), abs(float(sdev))
    if math.isnan(v) or math.isnan(dv):
        return '%g +- %g' % (v, dv)
    if dv == float('inf'):
        return '%g +- inf' % v
    if v == 0 and (dv >= 1e5 or dv < 1e-4):
        if dv == 0:
            return '0(0)'
        ans = ('%.1e' % dv).split('e')
        return '0.0(' + ans[0] + ')e' + ans[1]
    if v == 0:
        if dv >= 9.95:
            return '0(%.0f)' % dv
        if dv >= 0.995:
            return '0.0(%.1f)' % dv
        nd = _ndec(dv)
        return '%.*f(%.0f)' % (nd, v, dv * 10. ** nd)
    if dv == 0:
        ans = ('%g' % v).split('e')
        return ans[0] + '(0)' + ('e' + ans[1] if len(ans) == 2 else '')
    if dv < 1e-6 * abs(v):
        return '%g +- %.2g' % (v, dv)
    if dv > 1e4 * abs(v):
        return '%.1g +- %.2g' % (v, dv)
    if abs(v) >= 1e6 or abs(v) < 1e-5:
        exponent = math.floor(math.log10(abs(v)))
        fac = 10. ** exponent
        return fmt(v / fac, dv / fac) + 'e' + ('%.0e' % fac).split('e')[-1]
    if dv >= 9.95:
        if abs(v) >= 9.5:
            return '%.0f(%.0f)' % (v, dv)
        nd = _ndec(abs(v))
        return '%.*f(%.*f)' % (nd, v, nd, dv)
    if dv >= 0.995:
        if abs(v) >= 0.95:
            return '%.1f(%.1f)' % (v, dv)
        nd = _ndec(abs(v))
        return '%.*f(%.*f)' % (nd, v, nd, dv)
    # the mean contributes one digit less than the error (pinned by tests/test_lsqfit.py:1592,
    # '0.004(18)', and :1640, '0.010(13)': an error larger than the mean fixes the decimals)
    nd = max(_ndec(abs(v), 1), _ndec(dv))
    return '%.*f(%.0f)' % (nd, v, dv * 10. ** nd)


def fmt_array(mean, sdev):
    return '[' + ' '.join(fmt(m, s) for m, s in zip(mean, sdev)) + ']'
