"""lsqfit_amd -- MI355X-native Levenberg-Marquardt backend for lsqfit's fitter plugin.

Host side (Python, like the reference) of ONE path of gplepage/lsqfit: the
``nonlinear_fit(..., fitter=...)`` hot path (src/lsqfit/__init__.py:539-575,
:662-725).  All numerics below the plugin boundary run in hand-written gfx950
HIP kernels behind the C ABI declared in ``include/lsqfit_amd.h``; there is no
CPU fallback -- importing the compute entry points without the built library
raises.

  models     row models the kernels can differentiate (replaces user fcn + gvar AD)
  trace      ``trace(fcn, x, p0)``: the user's Python fit function recorded into a device tape (one call on tracer arrays)
  whiten     host mirror of gvar.PDF: block structure, svdcut, whitening weights
  fitter     ``mi355x_lm``: the fitter plugin class (mirror of gsl_multifit);
             ``mi355x_trf``: bounded fits (mirror of scipy_least_squares, method 'trf')
  fit        ``nonlinear_fit``: problem setup + chi2/dof/Q/logGBF reduction
  sweep      ``empbayes_fit`` / prior-width sweeps on one resident problem
  batched    ``BatchedFits``: many same-shape fits in lockstep, device-resident LM state, hipGraph
  dist       row sharding across GPUs + the all-reduce hook (torch.distributed/RCCL)
  synth      fake_fitargs-style synthetic problems (benchmark generator)
"""
from .models import Model, cosmix, multiexp, identity, expr, piecewise  # noqa: F401
from .whiten import Whitening  # noqa: F401
from .trace import trace, trace_residual, TraceError  # noqa: F401
from .fitter import mi355x_lm, mi355x_trf, DeviceProblem, register  # noqa: F401
from .fit import nonlinear_fit, gammaQ  # noqa: F401
from .sweep import empbayes_fit, prior_width_sweep  # noqa: F401
from .batched import BatchedFits  # noqa: F401

__version__ = '0.1.0'
