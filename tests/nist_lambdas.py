"""The 27 NIST StRD models as ordinary numpy functions ``fcn(x, b)`` -- what a user of lsqfit writes (cf. the generated
functions of /root/reference/examples/nist.py; typed here from the model definitions of the NIST data files, the same
definitions tests/golden/nist.json carries as text).  ``x`` is the predictor column (a dictionary ``{'x1':, 'x2':}`` for
nelson), ``b`` the parameter array, 0-based."""
import numpy as np
from numpy import arctan, cos, exp, pi, sin

_gauss = lambda x, b: b[0]*exp(-b[1]*x) + b[2]*exp(-(x-b[3])**2 / b[4]**2) + b[5]*exp(-(x-b[6])**2 / b[7]**2)   # noqa: E731
_lanczos = lambda x, b: b[0]*exp(-b[1]*x) + b[2]*exp(-b[3]*x) + b[4]*exp(-b[5]*x)                                # noqa: E731
_chwirut = lambda x, b: exp(-b[0]*x)/(b[1]+b[2]*x)                                                              # noqa: E731
_misra_a = lambda x, b: b[0]*(1-exp(-b[1]*x))                                                                   # noqa: E731

MODELS = dict(
    bennett5=lambda x, b: b[0] * (b[1]+x)**(-1/b[2]),
    boxbod=_misra_a,
    chwirut1=_chwirut,
    chwirut2=_chwirut,
    danwood=lambda x, b: b[0]*x**b[1],
    eckerle4=lambda x, b: (b[0]/b[1]) * exp(-0.5*((x-b[2])/b[1])**2),
    enso=lambda x, b: (b[0] + b[1]*cos(2*pi*x/12) + b[2]*sin(2*pi*x/12) + b[4]*cos(2*pi*x/b[3]) + b[5]*sin(2*pi*x/b[3])
                       + b[7]*cos(2*pi*x/b[6]) + b[8]*sin(2*pi*x/b[6])),
    gauss1=_gauss, gauss2=_gauss, gauss3=_gauss,
    hahn1=lambda x, b: (b[0]+b[1]*x+b[2]*x**2+b[3]*x**3) / (1+b[4]*x+b[5]*x**2+b[6]*x**3),
    kirby2=lambda x, b: (b[0] + b[1]*x + b[2]*x**2) / (1 + b[3]*x + b[4]*x**2),
    lanczos1=_lanczos, lanczos2=_lanczos, lanczos3=_lanczos,
    mgh09=lambda x, b: b[0]*(x**2+x*b[1]) / (x**2+x*b[2]+b[3]),
    mgh10=lambda x, b: b[0] * exp(b[1]/(x+b[2])),
    mgh17=lambda x, b: b[0] + b[1]*exp(-x*b[3]) + b[2]*exp(-x*b[4]),
    misra1a=_misra_a,
    misra1b=lambda x, b: b[0] * (1-(1+b[1]*x/2)**(-2)),
    misra1c=lambda x, b: b[0] * (1-(1+2*b[1]*x)**(-.5)),
    misra1d=lambda x, b: b[0]*b[1]*x*((1+b[1]*x)**(-1)),
    nelson=lambda x, b: b[0] - b[1]*x['x1'] * exp(-b[2]*x['x2']),
    rat42=lambda x, b: b[0] / (1+exp(b[1]-b[2]*x)),
    rat43=lambda x, b: b[0] / ((1+exp(b[1]-b[2]*x))**(1/b[3])),
    roszman1=lambda x, b: b[0] - b[1]*x - arctan(b[2]/(x-b[3]))/pi,
    thurber=lambda x, b: (b[0] + b[1]*x + b[2]*x**2 + b[3]*x**3) / (1 + b[4]*x + b[5]*x**2 + b[6]*x**3),
)
