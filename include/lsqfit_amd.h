/*
 * lsqfit_amd.h -- C ABI of the MI355X-native Levenberg-Marquardt backend.
 *
 * This is the drop-in boundary for ONE path of gplepage/lsqfit: the fitter
 * plugin that `nonlinear_fit` constructs at src/lsqfit/__init__.py:662-664
 *
 *     fit = FITTERS[name](p0, nf, chiv, tol=tol, maxit=maxit, **fitterargs)
 *
 * and whose attributes it reads back at :665-679 (error, cov, f, J, nit, tol,
 * stopping_criterion, description, results, x).  In the reference that plugin
 * is `gsl_multifit` (src/lsqfit/_gsl.pyx:414-723), a Cython wrapper around
 * GSL's C API; the entry points below are what a binding for this backend
 * binds instead (INTEGRATION.md shows the ctypes stub).  Every function cites
 * the reference interface it replaces.
 *
 * Conventions
 *   - plain C: pointers + sizes, no C++/torch types, no exceptions, no abort();
 *   - every call returns int: 0 success; GSL numbering for the codes the
 *     reference inspects (_gsl.pyx:686-701,:714): 11 EMAXITER, 27 ENOPROG,
 *     29 ETOLF, 30 ETOLX, 31 ETOLG; negative = backend failure
 *     (LSQAMD_E*); lsqamd_last_error() gives the text;
 *   - all state is handle-scoped (the reference keeps module globals
 *     _valder/_p_f/_pyerr, _gsl.pyx:397-399): handles are re-entrant and may be
 *     used from different host threads, one handle = one HIP stream.  Give every
 *     handle a NON-BLOCKING stream of its own (hipStreamNonBlocking): handles
 *     replay their LM step from captured graphs, and a capture cannot coexist
 *     with work on the legacy default stream -- on ROCm 7 a legacy-stream call
 *     (hipMemcpy, a NULL-stream launch) from ANY thread fails with
 *     hipErrorStreamCaptureImplicit while a capture is open and invalidates it
 *     (the library then falls back to eager launches for that handle).  The
 *     library itself never uses the legacy stream; a NULL stream argument is
 *     accepted but then nothing is ordered against the caller's legacy-stream work.
 *     A handle belongs to the device that was current at lsqamd_create: that
 *     device must be current in the calling thread for every later call on it
 *     (the library does not switch devices; lsqamd_init / _run / _step refuse
 *     with LSQAMD_EINVAL otherwise); kernel attributes are kept per device;
 *   - no C++ exception crosses the boundary: std::bad_alloc comes back as
 *     LSQAMD_ENOMEM, anything else as LSQAMD_EINTERNAL (csrc/common.h
 *     LSQAMD_ABI_CATCH on every export);
 *   - host buffers passed to lsqamd_set_* are copied before the call returns
 *     and never retained; output buffers are caller-allocated with an element
 *     capacity; device memory is ONE caller-provided workspace
 *     (lsqamd_workspace_bytes) that the handle carves and never frees;
 *   - matrices are row-major float64, like gsl_matrix (tda = size2,
 *     _gsl.pyx:60-64).
 */
#ifndef LSQFIT_AMD_H
#define LSQFIT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LSQAMD_ABI_VERSION 8

/* error codes (negative = backend, positive = GSL numbering) */
#define LSQAMD_SUCCESS 0
#define LSQAMD_EMAXITER 11 /* GSL_EMAXITER, _gsl.pyx:714 */
#define LSQAMD_ENOPROG 27  /* GSL_ENOPROG,  _gsl.pyx:698 */
#define LSQAMD_ETOLF 29
#define LSQAMD_ETOLX 30
#define LSQAMD_ETOLG 31
#define LSQAMD_EINVAL (-1)    /* bad argument / call order               */
#define LSQAMD_EHIP (-2)      /* HIP runtime error                       */
#define LSQAMD_ENOMEM (-3)    /* workspace too small, or a host allocation failed (std::bad_alloc) */
#define LSQAMD_ENOTPD (-4)    /* J^T J + mu D^2 not positive definite    */
#define LSQAMD_ENONFINITE (-5)/* residual or Jacobian not finite         */
#define LSQAMD_EUNSUPPORTED (-6)
#define LSQAMD_EREDUCE (-7)   /* the all-reduce hook reported failure    */
#define LSQAMD_ECAPACITY (-8) /* caller's output buffer too small        */
#define LSQAMD_EINTERNAL (-10) /* a C++ exception other than bad_alloc was caught at the ABI (never propagated:
                                 * every export is a function-try-block, csrc/common.h LSQAMD_ABI_CATCH) */
#define LSQAMD_EINACCURATE (-9) /* summary.cov_status only: the covariance was delivered, but the factorisation
                                 * behind it missed its own accuracy test (solver = qr: Q not orthogonal to 1e-6
                                 * after six passes) */

/* Row models f(x_i; p) the kernels evaluate with forward-mode AD.  They take
 * the place of the user's Python fit function + gvar.valder derivative
 * propagation (_gsl.pyx:671,742-760; _utilities.pyx:74-93), which a device
 * cannot call. */
enum {
  LSQAMD_MODEL_COSMIX = 1,   /* sum_k a_k cos(w_k x); p=[a_0..a_{K-1}, w_0..w_{K-1}] */
  LSQAMD_MODEL_MULTIEXP = 2, /* sum_k a_k exp(-E_k x); p=[a.., E..] (examples/y-vs-x.py:58-61) */
  LSQAMD_MODEL_TAPE = 3,     /* RPN expression tape over x[0..n_x) and p[0..P): ANY formula -- the stand-in for the
                              * user's Python fit function (examples/nist.py models); Jacobian by reverse-mode AD */
  LSQAMD_MODEL_IDENTITY = 4  /* f_i = p_i (tests/test_lsqfit.py:1815) */
};

enum { LSQAMD_SCALE_MORE = 0, LSQAMD_SCALE_LEVENBERG = 1, LSQAMD_SCALE_MARQUARDT = 2 }; /* _gsl.pyx:637-644 */
/* _gsl.pyx:646-653.  CHOLESKY: damped normal equations, covariance from their factor.  QR (the
 * reference's default; 'svd' maps here too): LM steps on the normal equations, the post-fit
 * covariance and log det J^T J through a column-equilibrated CholeskyQR factorisation of the
 * whitened Jacobian with re-orthogonalisation -- error ~ cond(J) eps like gsl's QR route, where the
 * normal equations give cond(J)^2 eps (examples/y-noerr.out at nexp = 5).  Needs lsqamd_set_qr_work. */
enum { LSQAMD_SOLVER_CHOLESKY = 0, LSQAMD_SOLVER_QR = 1 };
/* trust-region sub-problem solvers: gsl_multifit's `alg` keyword (_gsl.pyx:622-635) */
enum { LSQAMD_TRS_LM = 0, LSQAMD_TRS_LMACCEL = 1, LSQAMD_TRS_DOGLEG = 2, LSQAMD_TRS_DDOGLEG = 3,
       LSQAMD_TRS_SUBSPACE2D = 4,
       /* scipy_least_squares' method='trf' (src/lsqfit/_scipy.py:56-60,:135-139): Trust Region
        * Reflective; honours lsqamd_set_bounds.  With it: maxit is the cap on
        * function evaluations (max_nfev, _scipy.py:157), scaler LEVENBERG = x_scale 1.0 (default
        * there) and MORE = x_scale 'jac', summary.nit counts function evaluations (:161) and
        * summary.info = LSQAMD_INFO_TRF + scipy's status (0 max_nfev, 1 gtol, 2 ftol, 3 xtol,
        * 4 ftol and xtol), mapped to stopping_criterion as at :176-181. */
       LSQAMD_TRS_TRF = 5,
       /* method='dogbox' (_scipy.py:62-63): dogleg in a rectangular trust region with an active set;
        * same conventions as LSQAMD_TRS_TRF (bounds, maxit, scaler, nit, info). */
       LSQAMD_TRS_DOGBOX = 6,
       /* method='lm' (_scipy.py:64-67): MINPACK's lmder as scipy calls it (factor 100; scaler
        * LEVENBERG = diag 1/x_scale = 1, MORE = MINPACK's own column-norm scaling for x_scale 'jac');
        * no bounds; all three tolerances must exceed machine epsilon; maxit, nit, info as for TRF. */
       LSQAMD_TRS_MINPACK_LM = 7 };
#define LSQAMD_INFO_TRF 100

/* tape opcodes (LSQAMD_MODEL_TAPE); operands in `arg` */
enum {
  LSQAMD_OP_CONST = 0, /* push consts[arg] */
  LSQAMD_OP_X = 1,     /* push x[row][arg]  */
  LSQAMD_OP_P = 2,     /* push p[arg] (derivative seed e_arg) */
  LSQAMD_OP_ADD = 3, LSQAMD_OP_SUB = 4, LSQAMD_OP_MUL = 5, LSQAMD_OP_DIV = 6,
  LSQAMD_OP_POW = 7,   /* a ** b */
  LSQAMD_OP_NEG = 8, LSQAMD_OP_EXP = 9, LSQAMD_OP_LOG = 10, LSQAMD_OP_SIN = 11,
  LSQAMD_OP_COS = 12, LSQAMD_OP_ATAN = 13, LSQAMD_OP_SQRT = 14,
  LSQAMD_OP_POWI = 15, /* a ** (int)arg */
  /* the rest of what a gvar-overloaded fit function may call on a parameter (gvar's tan sinh cosh tanh arcsin arccos and
   * abs / fabs -- numpy.fabs on fit-function output: src/lsqfit/_extras.py:2569; cosh: the periodic two-point correlator
   * models lsqfit's documentation fits).  d|a|/da = +1 at a = 0, as gvar's GVar.__abs__ (returns self when mean >= 0). */
  LSQAMD_OP_TAN = 16, LSQAMD_OP_SINH = 17, LSQAMD_OP_COSH = 18, LSQAMD_OP_TANH = 19, LSQAMD_OP_ASIN = 20,
  LSQAMD_OP_ACOS = 21, LSQAMD_OP_ABS = 22,
  LSQAMD_OP_LAST = 22
};
#define LSQAMD_TAPE_MAX_PARAM 4096 /* parameters of a tape model */
#define LSQAMD_TAPE_CHUNK 16       /* interpreter fallback of batched fits: differentiated 16 at a time, ceil(P/16) forward passes per row */
#define LSQAMD_TAPE_MAX_STACK 16
#define LSQAMD_TAPE_MAX_CODE 16384 /* instructions.  Tapes are compiled (hiprtc) at set_tape time, single and batched fits alike;
                                    * the interpreter fallback of batched fits takes at most 1024 instructions */

/* Devices this process can use (one process per GPU; a host in another language picks its
 * device with the HIP runtime before lsqamd_create).  *count <- visible devices (0 without a GPU:
 * not an error); for device `index`, when it exists: arch[cap] <- its gfx name ("gfx950:..."),
 * *hbm_bytes <- its memory.  arch / hbm_bytes may be NULL. */
int lsqamd_query_devices(int32_t *count, int32_t index, char *arch, size_t cap, int64_t *hbm_bytes);

typedef struct lsqamd_fit lsqamd_fit; /* opaque handle (replaces gsl_multifit_nlinear_workspace, _gsl.pyx:672) */

/* Problem shape.  n_data rows are the LOCAL rows of this process when the fit
 * is row-sharded; the prior is replicated. */
typedef struct {
  int32_t abi_version;   /* LSQAMD_ABI_VERSION */
  int32_t model;         /* LSQAMD_MODEL_* */
  int64_t n_data;        /* N (local) : len(fcn(p)) */
  int64_t n_param;       /* P : len(x0), _gsl.pyx:618 */
  int32_t n_x;           /* predictors per data row */
  int32_t has_prior;     /* 0: chi2 over data only (prior=None, _utilities.pyx:72-73) */
  int32_t prior_dense;   /* 0: diagonal prior precision; 1: dense P x P */
  int32_t n_blocks;      /* correlated data blocks (contiguous row ranges); 0 = all rows 1x1 */
  int64_t max_block;     /* largest block size */
  int64_t sum_block_sq;  /* sum over blocks of B_b * B_b (whitening storage) */
  int32_t want_jacobian_out; /* keep the whitened J retrievable (fit.J, __init__.py:668) */
  int32_t n_batch;       /* independent fits sharing shape (1 unless batched sweep) */
  int32_t tape_len;      /* LSQAMD_MODEL_TAPE: instructions of the tape lsqamd_set_tape will bring (0 = up to
                          * 1024); sizes the per-row partial-derivative store of the reverse sweep */
  int32_t reserved1;
} lsqamd_config;

/* Driver options: gsl_multifit.__init__ keyword arguments (_gsl.pyx:563-575). */
typedef struct {
  double xtol, gtol, ftol;  /* tol normalised to a 3-tuple, _gsl.pyx:594-603 */
  int32_t maxit;            /* _gsl.pyx:568 */
  int32_t scaler;           /* LSQAMD_SCALE_* */
  int32_t solver;           /* LSQAMD_SOLVER_* */
  int32_t trs;              /* LSQAMD_TRS_*: alg, _gsl.pyx:569,:622-635 */
  double factor_up;         /* 3.0, _gsl.pyx:573 */
  double factor_down;       /* 2.0, _gsl.pyx:574 */
  double avmax;             /* 0.75, _gsl.pyx:575,:658 (lmaccel: reject when |a|/|v| exceeds it) */
} lsqamd_options;

/* What nonlinear_fit reads from the plugin (__init__.py:665-679) plus the
 * counters the benchmark reports. */
typedef struct {
  int32_t status;             /* GSL-style driver status (0, 11, ...) -> fit.error */
  int32_t info;               /* raw convergence info (1 xtol, 2 gtol, 27 ...) */
  int32_t stopping_criterion; /* 0..4, _gsl.pyx:690-701 */
  int32_t nit;                /* gsl_multifit_nlinear_niter, _gsl.pyx:713 */
  int32_t nfev;               /* residual evaluations (trial steps + 1) */
  int32_t njev;               /* Jacobian evaluations */
  int32_t ntrial;             /* damped solves attempted */
  int32_t chol_fail;          /* factorizations that hit a non-positive pivot */
  int32_t cov_status;         /* 0; k > 0: the final Jacobian is rank deficient and cov is what the reference's plugin returns there
                               * with k directions dropped (gsl_multifit_nlinear_covar's pivoted-QR form, _gsl.pyx:704-706; the
                               * thresholded-SVD pseudo-inverse of _scipy.py:170-175 for the TRF / dogbox / MINPACK methods),
                               * logdet_jtj = -inf; LSQAMD_EINACCURATE; or LSQAMD_ENOTPD: J^T J is not positive definite at the end point --
                               * cov and logdet_jtj (NaN) are undefined (gsl_multifit_nlinear_covar has no
                               * such report: its QR-based inverse returns garbage silently) */
  int32_t qr_trials;          /* solver = qr: trial steps whose damped system had no Cholesky factor and was solved from the
                               * orthogonal factorisation of [J ; sqrt(mu) D] instead (what gsl's qr solver does every time) */
  double chi2;                /* sum f**2, __init__.py:667 */
  double mu;                  /* final LM parameter */
  double logdet_jtj;          /* log det(J^T J) at the end (for logGBF, __init__.py:719) */
  double t_setup_ms, t_run_ms;
} lsqamd_summary;

/* All-reduce hook: sum `count` doubles at device address `dev_buf` over all
 * ranks, in place, and return only when the result is visible to work queued
 * afterwards on the handle's stream.  Return 0 on success.  (No counterpart in
 * the reference, which has no distributed path: SURVEY.md 5.) */
typedef int (*lsqamd_reduce_fn)(void *user, void *dev_buf, int64_t count);

/* ---- lifecycle ------------------------------------------------------------ */
int lsqamd_abi_version(void);
/* bytes of device workspace lsqamd_create needs for this shape */
size_t lsqamd_workspace_bytes(const lsqamd_config *cfg);
/* replaces gsl_multifit_nlinear_alloc (_gsl.pyx:672); `stream` is a hipStream_t (NULL = default) */
int lsqamd_create(const lsqamd_config *cfg, void *dev_workspace, size_t workspace_bytes,
                  void *stream, lsqamd_fit **out);
/* replaces gsl_multifit_nlinear_free (_gsl.pyx:720) */
int lsqamd_destroy(lsqamd_fit *fit);
/* replaces gsl_strerror (_gsl.pyx:687); valid until the next call on the handle */
const char *lsqamd_last_error(const lsqamd_fit *fit);

/* Replace the data means only (same covariance / whitening): the data of a simulated or
 * bootstrap copy of the fit (nonlinear_fit.simulated_data_iter, src/lsqfit/__init__.py:
 * 1470-1543: `y += f - mean(y)` then gvar.bootstrap_iter keeps the covariance). */
int lsqamd_set_ymean(lsqamd_fit *fit, const double *ymean);

/* ---- whitening set-up on the device ------------------------------------------------------
 * The O(B^3) part of what nonlinear_fit obtains from gvar.PDF(...) (src/lsqfit/__init__.py:
 * 1892-1900; svdcut semantics doc/source/overview.rst:1546-1606), for n_blocks covariance
 * blocks of ONE size B given back to back (cov[n_blocks][B][B], row-major, host or device
 * memory): correlation matrix -> Cholesky -> triangular inverse, all on the device.  Outputs:
 *   wt_out      device, [n_blocks][B][B]: the TRANSPOSED weights lsqamd_set_data takes
 *               (Wt_b = D^-1 inv(U_b), upper triangular; W_b^T W_b = inv(C_b));
 *   prec_out    device, [n_blocks][B][B] or NULL: inv(C_b) (what lsqamd_set_prior takes for a
 *               correlated prior block);
 *   logdet_out  host [n_blocks]: log det C_b;
 *   lam_min_out / lam_max_out  host [n_blocks] or NULL: a lower / an upper bound on the extreme
 *               eigenvalues of the block's correlation matrix;
 *   status_out  host [n_blocks]: 0 = weights valid and no eigenvalue lies below
 *               |svdcut| * lambda_max (gvar would leave the block untouched too); 1 = the svdcut
 *               floor may bind: the caller must take the eigen-mode route for this block;
 *               2 = not positive definite (same remedy, or an error).
 * lsqamd_set_data / lsqamd_set_prior / lsqamdb_set_blocks accept device pointers for their
 * weight / precision arguments, so the results never visit the host. */
size_t lsqamd_whiten_work_bytes(int64_t block_size, int32_t n_blocks);
int lsqamd_whiten_blocks(void *stream, int64_t block_size, int32_t n_blocks, const double *cov, double svdcut,
                         double *wt_out, double *prec_out, void *dev_work, size_t work_bytes,
                         double *logdet_out, double *lam_min_out, double *lam_max_out, int32_t *status_out);

/* ---- problem data (copied host -> device) ----------------------------------- */
/* x[n_data][n_x]: what the closure `flatfcn` hides (__init__.py:566-568,:1997-2042) */
int lsqamd_set_x(lsqamd_fit *fit, const double *x, int64_t n_rows, int32_t n_x);
/* tape for LSQAMD_MODEL_TAPE: code[n_code] = opcode | arg << 8 */
int lsqamd_set_tape(lsqamd_fit *fit, const int32_t *code, int32_t n_code,
                    const double *consts, int32_t n_consts);
/* Several formulas, each for its own contiguous range of data rows: the reference's fit function may
 * return a dictionary (or an array assembled from different expressions), which nonlinear_fit flattens
 * into ONE residual vector (_unpack_fcn / flatfcn, src/lsqfit/__init__.py:1997-2042; examples/simple.py:
 * `dict(data1=exp(a + x*b), data2=..., "b/a"=b/a)`): output i is computed by whatever expression the
 * user wrote for it.  Program k = code[code_off[k] .. code_off[k + 1]) (encoding as in lsqamd_set_tape)
 * covers rows row0[k] .. row0[k + 1] - 1; row0[0] = 0, row0[n_prog] = n_data, code_off[0] = 0; all
 * programs share the parameter vector, the predictors' layout and consts[].  Replaces a previous
 * lsqamd_set_tape (and vice versa).  lsqamd_config.tape_len must cover code_off[n_prog].  Single fits. */
int lsqamd_set_tape_programs(lsqamd_fit *fit, int32_t n_prog, const int64_t *row0, const int32_t *code,
                             const int32_t *code_off, const double *consts, int32_t n_consts);
/* Whitening of the data rows = PDF.mean / PDF.i_invwgts (_utilities.pyx:58-61):
 *   ymean[n_data];  wdiag[n_data]: 1/sdev for 1x1 rows (ignored inside blocks);
 *   block b covers rows [block_row0[b], block_row0[b]+block_size[b]) and has
 *   block_modes[b] <= block_size[b] kept modes; wt holds, back to back, each
 *   block's TRANSPOSED weights Wt_b[B_b][B_b] (row j = column j of W_b,
 *   columns >= block_modes[b] zero) so that W_b^T W_b = inv(C_b regulated).
 *   block_tri[b] != 0 promises Wt_b is upper triangular (W_b = inv(chol)).
 *   wt may point to host or device memory (lsqamd_whiten_blocks leaves its result on the device). */
int lsqamd_set_data(lsqamd_fit *fit, const double *ymean, const double *wdiag, int32_t n_blocks,
                    const int64_t *block_row0, const int64_t *block_size,
                    const int64_t *block_modes, const int32_t *block_tri, const double *wt);
/* Prior: mean[P] and precision = inv(C_prior regulated): prec[P] (diagonal) or
 * prec[P*P] (dense, symmetric) as cfg.prior_dense says.  The prior rows of
 * chiv (_utilities.pyx:76-77) enter J^T J / J^T f / chi2 through it. */
int lsqamd_set_prior(lsqamd_fit *fit, const double *mean, const double *prec);   /* prec: host or device memory */
/* Data-prior cross-correlations (the reference whitens concat(y, prior) as ONE vector,
 * src/lsqfit/__init__.py:1892-1900; examples/y-noerr.py): create the fit with has_prior = 0 and
 * n_data = N + (number of prior entries), give the prior entries as extra rows of ymean / wdiag /
 * the covariance blocks (any block may mix both kinds), and flag them here: row_param[n_data] holds
 * -1 for a model row and j >= 0 for a row whose "model" is the parameter p_j itself (its x is
 * ignored).  NULL clears.  Such rows shard like any others (a shard is a range of whole covariance
 * blocks); lsqamd_eval_fcn returns p_j for them; lsqamd_dpdy returns columns for the rows as given
 * (data and prior entries alike). */
int lsqamd_set_param_rows(lsqamd_fit *fit, const int32_t *row_param);
int lsqamd_set_options(lsqamd_fit *fit, const lsqamd_options *opt);
/* Device scratch of the QR-grade covariance (LSQAMD_SOLVER_QR): a transposed copy of J, the
 * orthogonalised Q and a few P x P matrices -- lsqamd_qr_work_bytes(fit) bytes the caller owns and
 * lends to the handle (NULL takes it back).  lsqamd_qr_info: orthogonalisation passes of the last
 * covariance and max |Q^T Q - I| seen before the final factor. */
size_t lsqamd_qr_work_bytes(const lsqamd_fit *fit);
int lsqamd_set_qr_work(lsqamd_fit *fit, void *dev_work, size_t work_bytes);
int lsqamd_qr_info(const lsqamd_fit *fit, int32_t *passes, double *delta);
/* Box bounds lower[P] < upper[P] (+-INFINITY = open side; NULL array = open everywhere; both
 * NULL clears them): the flattened `bounds` pair nonlinear_fit hands to scipy_least_squares
 * (src/lsqfit/__init__.py:641-655, tests/test_lsqfit.py:1780-1808).  LSQAMD_TRS_TRF and
 * LSQAMD_TRS_DOGBOX read them; p0 must lie inside (else lsqamd_run returns LSQAMD_EINVAL, as scipy raises). */
int lsqamd_set_bounds(lsqamd_fit *fit, const double *lower, const double *upper);
/* scipy_least_squares' other pass-through options (src/lsqfit/_scipy.py:76-79 names them, :147-153 forwards
 * **extra_args verbatim to scipy.optimize.least_squares).
 * loss / f_scale: scipy's robust losses rho(z), z = (f_i / f_scale)^2, over the elements of the whitened residual
 * vector -- LSQAMD_TRS_TRF and LSQAMD_TRS_DOGBOX only (scipy's 'lm' refuses them too: LSQAMD_EINVAL).  Every
 * residual must then be a ROW (a prior goes in through lsqamd_set_param_rows, not lsqamd_set_prior:
 * LSQAMD_EUNSUPPORTED otherwise).  As in the reference the fit point minimises the robust cost, cov comes from the
 * loss-scaled Jacobian scipy returns (:165-169), chi2 and logdet_jtj from the true residuals and Jacobian
 * (src/lsqfit/__init__.py:667,:719), which is also what the getters return afterwards.
 * x_scale[P] > 0 (NULL: scipy's default 1.0): characteristic scale of each parameter; read with scaler
 * LSQAMD_SCALE_LEVENBERG by the three scipy methods (scaler MORE is x_scale = 'jac'). */
enum { LSQAMD_LOSS_LINEAR = 0, LSQAMD_LOSS_SOFT_L1 = 1, LSQAMD_LOSS_HUBER = 2, LSQAMD_LOSS_CAUCHY = 3, LSQAMD_LOSS_ARCTAN = 4 };
int lsqamd_set_loss(lsqamd_fit *fit, int32_t loss, double f_scale);
int lsqamd_set_x_scale(lsqamd_fit *fit, const double *x_scale);
/* nonlinear_fit's `linear=` (src/lsqfit/__init__.py:738-787, _varpro_fit; tests/test_lsqfit.py:1642-1682):
 * index[n] names the parameters the fit function is linear in (n = 0 clears): variable projection.
 * The reference wraps the residual so that every evaluation first solves for them exactly; so does
 * the device -- every trial point gets a full evaluation followed by the exact linear solve
 * A_aa da = -g_a, and the step of the other parameters is the LM step of the projected functional
 * (Schur complement of the normal matrix; Kaufman's form).  Same minimum, covariance and chi2 as
 * the plain fit in fewer iterations on multi-exponential problems, at two evaluations per
 * accepted step.  Plain LSQAMD_TRS_LM only; nfev / njev count every full evaluation. */
int lsqamd_set_linear(lsqamd_fit *fit, const int32_t *index, int32_t n);
int lsqamd_set_reduce(lsqamd_fit *fit, lsqamd_reduce_fn fn, void *user);
/* The exchange inside the library (what a host in any language uses; the hook above stays for
 * transports RCCL does not cover, e.g. the CPU-side gloo tests): a persistent RCCL communicator
 * per process and id (shared by the handles that name it, see lsqamd_comm_stats), one rank per GPU.  Rank 0 calls lsqamd_comm_unique_id and ships the
 * LSQAMD_COMM_ID_BYTES bytes to the other ranks by whatever channel the host has (MPI, a file,
 * torch.distributed); then EVERY rank calls lsqamd_comm_init (collective: returns when all
 * nranks have joined).  From then on the sums of the row-sharded fit -- the packed
 * [J^T J | J^T f | chi2] buffer per Jacobian evaluation, one scalar per trial step -- are
 * enqueued on the handle's stream as ncclReduceScatter + ncclAllGather (short vectors:
 * ncclAllReduce): no stream synchronisation, no host callback, every rank receives identical
 * bytes.  Takes precedence over lsqamd_set_reduce.  LSQAMD_EUNSUPPORTED: no librccl.so in the
 * process or on the loader path (LSQAMD_RCCL_PATH names one).  (No counterpart in the
 * reference, which has no distributed path: SURVEY.md 5, 8e.) */
#define LSQAMD_COMM_ID_BYTES 128
int lsqamd_comm_unique_id(void *id_out, size_t cap);
int lsqamd_comm_init(lsqamd_fit *fit, const void *id, size_t id_bytes, int32_t rank, int32_t nranks);
int lsqamd_comm_destroy(lsqamd_fit *fit);
/* *rank / *nranks of the handle's communicator (-1 / 0 when there is none) */
int lsqamd_comm_info(const lsqamd_fit *fit, int32_t *rank, int32_t *nranks);
/* Communicators belong to the PROCESS: the first lsqamd_comm_init that names an id creates it (ncclCommInitRank, collective),
 * every later handle that names the same id on the same device shares it (no collective call; handles sharing one must not
 * run their sums at the same time), lsqamd_comm_destroy / lsqamd_destroy only drop the handle's reference.
 * lsqamd_comm_stats: *init_ms = what creating the handle's communicator took, *handles = handles holding it now.
 * lsqamd_comm_shutdown: destroys every communicator no handle holds; returns how many are still held. */
int lsqamd_comm_stats(const lsqamd_fit *fit, double *init_ms, int32_t *handles);
int lsqamd_comm_shutdown(void);
/* Row-sharded fits: exactly one rank (on != 0) contributes the replicated prior
 * terms to the sums before the all-reduce.  Default on. */
int lsqamd_set_adds_prior(lsqamd_fit *fit, int32_t on);

/* ---- the hot path ------------------------------------------------------------ */
/* replaces gsl_multifit_nlinear_init + _driver (_gsl.pyx:676-677) */
int lsqamd_run(lsqamd_fit *fit, const double *p0, lsqamd_summary *out);
/* the same, one piece at a time (benchmark timing, sweeps):
 *   init  = gsl_multifit_nlinear_init      (f, J, g, D, mu at p0)
 *   step  = gsl_multifit_nlinear_iterate + _test; *converged = info (0 = continue)
 *   finish= covariance + logdet at the current point */
int lsqamd_init(lsqamd_fit *fit, const double *p0);
int lsqamd_step(lsqamd_fit *fit, int32_t *info);
int lsqamd_finish(lsqamd_fit *fit, lsqamd_summary *out);

/* ---- kernel-level entry points (parity tests, roofline measurement) ----------- */
/* _c_f  (_gsl.pyx:727-740): whitened residual at p; returns chi2 = |f|^2 */
int lsqamd_eval_residual(lsqamd_fit *fit, const double *p, double *chi2);
/* _c_df (_gsl.pyx:742-760) + solver.init: J, then J^T J, J^T f, chi2 at p */
int lsqamd_eval_normal(lsqamd_fit *fit, const double *p, double *chi2);
/* out[N] = fcn(x; p), unwhitened: what simulated_data_iter evaluates at pexact
 * (src/lsqfit/__init__.py:1524 `f = self.fcn(self.x, pexact)`) */
int lsqamd_eval_fcn(lsqamd_fit *fit, const double *p, double *out, size_t cap);
/* damped solve (J^T J + mu D^2) v = J^T f with D = diag (host, P); v -> host */
int lsqamd_solve_damped(lsqamd_fit *fit, double mu, const double *diag, double *v);
/* raw dense ops on device pointers (row-major, see gemm_tn_f64.hip) */
int lsqamd_op_gemm_tn(void *stream, int64_t M, int64_t N, int64_t K, double alpha, const double *X,
                      int64_t ldx, const double *Y, int64_t ldy, double beta, double *C, int64_t ldc,
                      int32_t upper_only, int32_t x_upper_tri);
int lsqamd_op_potrf_upper(void *stream, double *A, int64_t n, int64_t lda, int64_t n_cols,
                          double *work, size_t work_bytes, int32_t *dev_info);
size_t lsqamd_op_potrf_work_bytes(int64_t n);
/* host arithmetic of the rank-deficient covariance (rankdef.hip; what lsqamd_run / lsqamd_finish fall back to when
 * J^T J has no Cholesky factor at the end point): G[n*n] = J^T J, n_rows = rows of J; scipy_form 0: gsl_multifit_nlinear_covar's
 * pivoted recipe (_gsl.pyx:704-706), 1: the thresholded pseudo-inverse of _scipy.py:170-175.  Host pointers, no GPU. */
int lsqamd_op_truncated_inverse(const double *G, int64_t n, int64_t n_rows, int32_t scipy_form, double *cov_out,
                                int32_t *dropped);

/* ---- results (replace vector2array / matrix2array, _gsl.pyx:77-86,:104-120) ---- */
int lsqamd_get_x(lsqamd_fit *fit, double *out, size_t cap);       /* P        : fit.x  */
int lsqamd_get_f(lsqamd_fit *fit, double *out, size_t cap);       /* nf       : fit.f  */
int lsqamd_get_J(lsqamd_fit *fit, double *out, size_t cap);       /* nf * P   : fit.J  */
int lsqamd_get_jtj(lsqamd_fit *fit, double *out, size_t cap);     /* P * P    : J^T J  */
int lsqamd_get_grad(lsqamd_fit *fit, double *out, size_t cap);    /* P        : J^T f  */
int lsqamd_get_cov(lsqamd_fit *fit, double *out, size_t cap);     /* P * P    : fit.cov (gsl_multifit_nlinear_covar, _gsl.pyx:704-706) */

/* ---- chi**2 at many parameter points -------------------------------------
 * Replaces the Python loop / numpy batch behind nonlinear_fit.dchi2 / .pdf
 * (_fit_dchi2, _fit_pdf: src/lsqfit/__init__.py:1648-1816, `sum(chiv(p)**2)`) and the
 * batched layout of vegas_fit._chiv (src/lsqfit/_extras.py:2467-2486,
 * `chiv[:, iw] = delta[:, iw].dot(wgt.T)`): chi2_out[i] = |chiv(p_i)|^2 for m points
 * p[m x P] (host, row-major), prior term included.  dev_scratch: device memory from the
 * caller; lsqamd_chi2_points_work_bytes(fit, m) processes all m points in one pass, a smaller
 * buffer makes the call work in chunks.  Row-sharded fits: the per-point sums go through
 * the all-reduce hook. */
size_t lsqamd_chi2_points_work_bytes(const lsqamd_fit *fit, int64_t m);
int lsqamd_chi2_points(lsqamd_fit *fit, const double *p, int64_t m, void *dev_scratch, size_t scratch_bytes,
                       double *chi2_out);

/* ---- sensitivity of the best-fit parameters to the inputs ------------
 * Replaces the matrix D[a,i] = d pmean[a] / d buf[i], buf = concat(y, prior), that
 * nonlinear_fit._getp builds column by column from chivw and cov
 * (src/lsqfit/__init__.py:897-911, `D[:, i] = chivw_i.mdotder(self.cov)`;
 * chivw = inv(C_reg) . delta, src/lsqfit/_utilities.pyx:96-139):
 *     D = cov . [J_f ; I]^T . inv(C_reg)          (doc/source/lsqfit.rst:105-117)
 * evaluated at the current point from the whitened Jacobian resident on the device.
 *   gt     host, P x m row-major: m directions in parameter space (gradients of the
 *          outputs whose error budget is wanted); NULL = identity (m must equal P).
 *   out_t  host, (N + P) x m row-major (N x m without a prior) = D^T . gt : rows 0..N-1
 *          are this handle's data rows in the caller's order, then the P prior entries.
 *   dev_scratch  device memory of lsqamd_dpdy_work_bytes(fit, m) bytes from the caller.
 * The handle's covariance is (re)computed if needed.  Row-sharded fits: every rank gets
 * the rows of its own shard (cov is replicated); no collective. */
size_t lsqamd_dpdy_work_bytes(const lsqamd_fit *fit, int64_t m);
int lsqamd_dpdy(lsqamd_fit *fit, const double *gt, int64_t m, void *dev_scratch, size_t scratch_bytes,
                double *out_t, size_t cap);
int64_t lsqamd_nf(const lsqamd_fit *fit);                         /* nchiv (__init__.py:574) */

/* ---- batched fits (SURVEY.md 8a row a8 / BASELINE.json config 5) --------------------
 * B independent fits of ONE shape -- same model, x, data means and diagonal data whitening,
 * different (diagonal) priors and starting points -- advanced in lockstep with all LM state
 * on the device and one round of kernels captured in a hipGraph.  Device counterpart of the
 * Python loop of whole fits in lsqfit.empbayes_fit (src/lsqfit/_extras.py:153-174); per fit
 * the driver semantics are those of lsqamd_run. */
typedef struct lsqamdb_fits lsqamdb_fits;
size_t lsqamdb_workspace_bytes(const lsqamd_config *cfg, int32_t n_fits);
int lsqamdb_create(const lsqamd_config *cfg, int32_t n_fits, void *dev_workspace, size_t workspace_bytes,
                   void *stream, lsqamdb_fits **out);
int lsqamdb_destroy(lsqamdb_fits *fits);
const char *lsqamdb_last_error(const lsqamdb_fits *fits);
int lsqamdb_set_x(lsqamdb_fits *fits, const double *x, int64_t n_rows, int32_t n_x);
int lsqamdb_set_tape(lsqamdb_fits *fits, const int32_t *code, int32_t n_code, const double *consts,
                     int32_t n_consts);
int lsqamdb_set_data(lsqamdb_fits *fits, const double *ymean, const double *wdiag);     /* shared, [N] each */
/* correlated data blocks shared by the fits: arguments as in lsqamd_set_data (whitening
 * weights of gvar.PDF, src/lsqfit/_utilities.pyx:58-61); required when cfg.n_blocks > 0 */
int lsqamdb_set_blocks(lsqamdb_fits *fits, int32_t n_blocks, const int64_t *row0, const int64_t *size,
                       const int64_t *modes, const int32_t *tri, const double *wt);
/* per-fit data means ymean[B*N]: simulated / bootstrap copies of one data set keep its
 * covariance and replace the means (simulated_fit_iter / bootstrapped_fit_iter,
 * src/lsqfit/__init__.py:1391-1469,1548-1642) */
int lsqamdb_set_data_means(lsqamdb_fits *fits, const double *ymean);
/* mean[B*P]; prec[B*P] = 1/sdev^2 per fit, or -- cfg.prior_dense -- ONE dense P x P precision shared by
 * all fits (simulated / bootstrap copies differ in their prior means, not in the prior covariance) */
int lsqamdb_set_priors(lsqamdb_fits *fits, const double *mean, const double *prec);
int lsqamdb_set_options(lsqamdb_fits *fits, const lsqamd_options *opt);
/* p0[B*P]; summaries[B] or NULL (t_setup_ms carries the number of graph-replayed rounds) */
int lsqamdb_run(lsqamdb_fits *fits, const double *p0, lsqamd_summary *summaries, int32_t use_graph);
int lsqamdb_get_x(lsqamdb_fits *fits, double *out, size_t cap);                         /* B*P */
int lsqamdb_covariance(lsqamdb_fits *fits, double *logdet_jtj_out, size_t cap);         /* B */
int lsqamdb_get_cov(lsqamdb_fits *fits, int32_t fit, double *out, size_t cap);          /* P*P */
int lsqamdb_get_cov_all(lsqamdb_fits *fits, double *out, size_t cap);                    /* n_fits*P*P, one copy */
int32_t lsqamdb_rounds(const lsqamdb_fits *fits);
/* phase timers of the batched engine (the measurement side of SURVEY.md 8d for BASELINE config 5): with timing on, every
 * round runs eagerly with HIP events around the batched J^T J launch (which = LSQAMD_T_SYRK) and the batched
 * factorisation (LSQAMD_T_CHOLESKY) on the engine's stream; _get waits for the stream and returns the totals since _enable. */
int lsqamdb_timing_enable(lsqamdb_fits *fits, int32_t on);
int lsqamdb_timing_get(lsqamdb_fits *fits, int32_t which, double *total_ms, int64_t *count);

/* ---- measurement ------------------------------------------------------------ */
enum {
  LSQAMD_T_RESIDUAL = 0, LSQAMD_T_JACOBIAN = 1, LSQAMD_T_WHITEN = 2, LSQAMD_T_SYRK = 3,
  LSQAMD_T_GRAD = 4, LSQAMD_T_REDUCE = 5, LSQAMD_T_CHOLESKY = 6, LSQAMD_T_SOLVE = 7,
  LSQAMD_T_COVAR = 8,
  /* sharded fits: the collective itself (HIP events around it on the stream it runs on: the step's stream, or the handle's
   * exchange stream when LSQAMD_EXCHANGE_GROUPS > 1) and the time the STEP's stream spent waiting for it -- their ratio is the
   * exposed share of the exchange (1 when nothing overlaps it) */
  LSQAMD_T_EXCH_COLL = 9, LSQAMD_T_EXCH_WAIT = 10, LSQAMD_T_COUNT = 11
};
/* HIP-event timing of each phase on the handle's stream; off by default */
int lsqamd_timing_enable(lsqamd_fit *fit, int32_t on);
int lsqamd_timing_get(lsqamd_fit *fit, int32_t which, double *total_ms, int64_t *count);
int lsqamd_timing_reset(lsqamd_fit *fit);
/* introspection for tests: bit 0 = batched (uniform-block) whitening in use, bit 1 = Jacobian rows
 * synthesised inside the whitening product, bit 4 = the normal equations of the last accepted step were formed without writing the Jacobian (few
 * parameters, compiled formula, uncorrelated rows), bit 3 = the tape model's formula runs as compiled code (hiprtc) rather than through the
 * interpreter kernels, bit 2 = LM steps replayed from captured graphs (small
 * single-rank problems without phase timing; LSQAMD_STEP_GRAPH=0 disables), bit 5 = the last lsqamd_run was ONE
 * kernel launch (compiled formula; <= 12 parameters and <= 8192 uncorrelated or <= 256 correlated rows, or <= 32
 * parameters and a few hundred to two thousand rows, fewer when correlated; plain lm: every iteration of
 * gsl_multifit_nlinear_init + _driver + _covar, src/lsqfit/_gsl.pyx:676-677,:706, by one workgroup;
 * LSQAMD_ONE_LAUNCH_FIT=0 disables),
 * bits 8..31 = split-K factor of the J^T J kernel, bits 32.. = block count */
int64_t lsqamd_debug_flags(const lsqamd_fit *fit);
/* Small fits hand their results to the host through pinned memory the host polls (no reference counterpart: the
 * reference's driver runs on the host, src/lsqfit/_gsl.pyx:676-701).  Device stores to host memory arrive in no
 * particular order, so every such block carries a sequence number and a checksum and the host acts on a snapshot
 * only when both fit.  out3[0] = snapshots that did not verify yet (polled again), out3[1] = hand-offs served from
 * device memory after a stream synchronisation instead, out3[2] = words the test knob LSQAMD_VERIFY_HANDOFF=1 found
 * different from the device's own copy after the host had acted on them (must stay 0).  Process-wide counters. */
int lsqamd_handoff_stats(int64_t *out3);
/* self-tests of the boundary (no GPU needed; tests/test_abi.py).  lsqamd_debug_throw raises a C++ exception INSIDE the
 * library (kind 1 std::bad_alloc, 2 std::runtime_error, 3 a non-std object, 4 a real over-sized allocation) and must come
 * back as LSQAMD_ENOMEM / LSQAMD_EINTERNAL with the text in lsqamd_last_error(fit) -- the reference's callbacks are
 * `noexcept` and stash Python errors (src/lsqfit/_gsl.pyx:726-760,:680-685); nothing may unwind into the host language.
 * lsqamd_debug_per_device_once(dev, 0) returns how many times the per-device "set kernel attributes" action has run for
 * device `dev` after this call (1 however often it is called, separately per device; reset != 0 clears the counters). */
int lsqamd_debug_throw(lsqamd_fit *fit, int32_t kind);
/* GPU self-test of the recovery from an invalidated graph capture (csrc/common.h capture_reset): a capture on
 * `stream` (non-blocking, idle), ONE legacy-stream hipMemcpy from another thread while it is open, the reset, an eager copy.
 * report[6]: intruder's hipMemcpy result, hipStreamEndCapture's result, capture status after it, status after the reset,
 * result of the eager copy + synchronise, the value it delivered (42). */
int lsqamd_debug_capture_selftest(void *stream, int32_t *report);
int lsqamd_debug_per_device_once(int32_t dev, int32_t reset);
/* the work list of the J^T J launch as the library builds it (host arithmetic; tests/test_abi.py): entries (tile row, tile
 * column, K-split, exchange group or 0) in launch order -- workgroup b takes entry  run_start(b % 8) + b / 8  of eight
 * contiguous runs.  G = 0: the plain list; G >= 1: rows[G + 1] tile-row bounds of the exchange groups, count[G] filled. */
int64_t lsqamd_debug_syrk_work(int64_t n_param, int32_t splits, int32_t n_groups, const int32_t *rows, int32_t *out4, int32_t *count);
/* The process-wide cache of compiled formulas (lsqamd_set_tape compiles with hiprtc and keeps the loaded code object):
 * out3[0] = kernels loaded now, out3[1] = of those, held by a live handle, out3[2] = kernels unloaded so far.  At most
 * LSQAMD_JIT_CACHE_CAP (default 1024) stay loaded; beyond that the ones no handle holds go, least recently used first. */
int lsqamd_jit_cache_stats(int64_t *out3);
/* The formula of a tape model as the straight-line gfx950 code lsqamd_set_tape builds with hiprtc in
 * place of interpreting the tape (bit 3 of lsqamd_debug_flags: the compiled route is in use; it stands
 * in for the Python fit function the reference differentiates with gvar.valder, src/lsqfit/_gsl.pyx:
 * 742-760).  Needs no GPU: src_out[cap] <- the generated source (or the reason it was declined),
 * *variant <- 0 one lane per data row, 1 one wave per data row with the lanes striding look-alike
 * terms of its sums; compile != 0 also runs hiprtc on it.  Returns 0, LSQAMD_EUNSUPPORTED (formula
 * outside what the generator handles / no hiprtc: the interpreter kernels run instead), LSQAMD_EHIP
 * (hiprtc rejected the source: a bug). */
int lsqamd_tape_codegen(const int32_t *code, int32_t n_code, const double *consts, int32_t n_consts, int32_t n_param,
                        int32_t n_x, char *src_out, size_t cap, int32_t *variant, int32_t compile);
/* developer builds (-DLSQAMD_POTF2_TIMING): device buffer of 32 int64 cycle stamps written by the
 * diagonal-block Cholesky kernel; NULL (default) disables */
void lsqamd_debug_set_potf2_stamps(void *dev_ptr);
#ifdef __cplusplus
}
#endif
#endif /* LSQFIT_AMD_H */
