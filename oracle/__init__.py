"""CPU oracle for the lsqfit LM hot path -- TEST INFRASTRUCTURE ONLY.

This package is a from-scratch numpy restatement of the reference algorithm
(gplepage/lsqfit 13.3.1 + the parts of GSL ``multifit_nlinear`` and ``gvar``
it leans on).  It exists to *check* the HIP backend in ``lsqfit_amd`` and to
serve as the timed CPU baseline in ``bench.py``.  Nothing in the product
package ``lsqfit_amd`` may import it: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do.

Parity status: PINNED for converged (p, sdev/cov, chi2, dof, Q, logGBF) on
the golden vectors the reference ships (NIST StRD certified values,
``examples/*.out`` header lines, literal assertions in
``tests/test_lsqfit.py``) -- see ``tests/test_oracle_*.py``.  UNPINNED:
iteration counts / trajectories (no reference test asserts them) and the
``eps`` regulation mode (no literal expected values anywhere in the
reference).  GSL and gvar sources are NOT under /root/reference; their
algorithms are restated from the published code and documentation
(GSL >= 2.2.1 per INSTALLATION.txt:6, gvar >= 13.1.5 per setup.cfg:21) and
anchored on the reference's own call sites and expected outputs.

Modules
  gvar_lite  -- "1.23(45)" parsing / formatting          (gvar, 3rd party)
  dual       -- forward-mode AD carrier (gvar.valder)     (_gsl.pyx:671,742-760)
  pdf        -- whitening of [y; prior] (gvar.PDF)        (__init__.py:1892-1900)
  chiv       -- whitened residual builder                 (_utilities.pyx:50-94)
  lm         -- gsl_multifit_nlinear trust/LM driver      (_gsl.pyx:563-723)
  fit        -- nonlinear_fit problem setup + reductions  (__init__.py:455-737)
  synth      -- fake_fitargs-style generators             (_extras.py:2508-2589)
  trf, minpack -- scipy's least_squares methods restated (the scipy plugin, _scipy.py:115-181; pinned on scipy itself)
  csrc/qrpt_unblocked.c + build_c -- the ONE C file: an unblocked column-pivoted Householder QR (the algorithm class
               of gsl's default `qr` solver, _gsl.pyx:646-647), gcc -O2 -> oracle/_build/liboracle_c.so; checked against
               LAPACK (tests/test_oracle_qr_c.py) and timed by bench.py's cpu_baseline "faithful" mode
"""
