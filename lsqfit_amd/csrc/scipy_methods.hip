// scipy_least_squares' three methods on the device's normal equations:
// Trust Region Reflective and dogbox (box bounds), MINPACK's lmder.  Host-side step logic only --
// every O(N P), O(P^2) and O(P^3) operation is one of the building blocks of api.hip
// (fit_state.h); the reference reaches these methods through src/lsqfit/_scipy.py:115-181.
#include "fit_state.h"

using namespace lsqamd;

namespace lsqamd_host {

// ---- Trust Region Reflective with box bounds --------------------------
// What src/lsqfit/_scipy.py:115-181 gets from scipy.optimize.least_squares(method='trf', bounds=...)
// (Branch, Coleman and Li), restated on the normal equations this library already forms: scipy's
// exact sub-problem solver works from the SVD of the scaled, augmented Jacobian (J d | sqrt C);
// every quantity it takes from that SVD is a function of B = d A d + C with A = J^T J, so here
//   p(alpha)   = -(A + E)^-1 g,  E = (C + alpha) / d^2           (Cholesky on the device)
//   phi(alpha) = |p_h| - Delta,  phi' = -p_h.(B + alpha)^-1 p_h / |p_h|   (second solve, same factor)
// and the model values come from A-products (one device GEMV each) or, for the sub-problem
// solution itself, from the linear system it satisfies.  Coleman-Li scaling, step selection
// (cut back / reflected / gradient), radius update and the ftol / xtol / gtol tests are O(P)
// host work on vectors that are already on the host.
struct TrfOuter {       // fixed during the trial steps of one outer iteration
  std::vector<double> d, C, gh, Aag;   // Aag = A (d ag_h), ag_h = -g_h (lazily)
  bool have_Aag = false;
  bool gn_tried = false, full_rank = false;
  std::vector<double> p_gn;            // Gauss-Newton point (original variables)
  double gn_norm = 0.0, phi0_slope = 0.0;   // |p_gn,h| and phi'(0)
  double theta = 0.0;
};

double norm_h(const std::vector<double> &v) { return std::sqrt(dot_h(v, v)); }

// (A + (C + alpha)/d^2) p = -g -> p (original variables); LSQAMD_ENOTPD when the factorisation fails
int trf_solve(lsqamd_fit *f, const TrfOuter &o, double alpha, std::vector<double> &p) {
  const int64_t P = f->P;
  std::vector<double> e(P);
  for (int64_t j = 0; j < P; ++j) e[j] = std::sqrt((o.C[j] + alpha)) / o.d[j];
  const int rc = solve_damped_dev(f, 1.0, e.data());
  if (rc) return rc;
  for (int64_t j = 0; j < P; ++j) p[j] = -f->hv[j];
  return 0;
}

// with the factor of (A + E) still in place: -> p_h.(B + alpha)^-1 p_h  for p_h = p / d
int trf_slope_term(lsqamd_fit *f, const TrfOuter &o, const std::vector<double> &p, double *out) {
  const int64_t P = f->P;
  std::vector<double> rhs(P), z(P);
  for (int64_t j = 0; j < P; ++j) rhs[j] = p[j] / (o.d[j] * o.d[j]);
  const int rc = solve_with_factor(f, rhs.data(), z.data());
  if (rc) return rc;
  *out = dot_h(rhs, z);
  return 0;
}

// solve_lsq_trust_region: p_h with |p_h| <= Delta; *quad = p_h.B p_h of the returned step
int trf_subproblem(lsqamd_fit *f, TrfOuter &o, double Delta, double *alpha_io, std::vector<double> &p_h,
                   double *quad) {
  const int64_t P = f->P;
  std::vector<double> p(P);
  auto to_hat = [&](const std::vector<double> &v) { for (int64_t j = 0; j < P; ++j) p_h[j] = v[j] / o.d[j]; };
  if (!o.gn_tried) {
    o.gn_tried = true;
    o.p_gn.assign(P, 0.0);
    int rc = trf_solve(f, o, 0.0, o.p_gn);
    if (rc < 0 && rc != LSQAMD_ENOTPD) return rc;
    o.full_rank = rc == 0;
    if (o.full_rank) {
      to_hat(o.p_gn);
      o.gn_norm = norm_h(p_h);
      double t = 0.0;
      rc = trf_slope_term(f, o, o.p_gn, &t);
      if (rc) return rc;
      o.phi0_slope = -t / o.gn_norm;
    }
  }
  if (o.full_rank && o.gn_norm <= Delta) {
    to_hat(o.p_gn);
    *alpha_io = 0.0;
    *quad = -dot_h(f->hg, o.p_gn);          // B p_h = -g_h
    return 0;
  }
  double hi = norm_h(o.gh) / Delta;
  double lo = o.full_rank ? -(o.gn_norm - Delta) / o.phi0_slope : 0.0;
  auto restart = [&]() { return std::fmax(0.001 * hi, std::sqrt(lo * hi)); };
  double alpha = (!o.full_rank && *alpha_io == 0.0) ? restart() : *alpha_io;
  // A singular B (a dead Jacobian column, say) is where scipy's SVD iteration may wander to
  // negative shifts; a Cholesky factorisation needs B + alpha positive definite, so the shift is
  // kept in (lo, hi] and the last successfully factored one is the fallback.
  double good_alpha = -1.0;
  for (int it = 0; it < 10; ++it) {
    if (alpha < lo || alpha > hi || (!o.full_rank && !(alpha > 0.0))) alpha = restart();
    int rc = trf_solve(f, o, alpha, p);
    if (rc == LSQAMD_ENOTPD) {               // not positive definite at this shift: move up
      lo = std::fmax(lo, alpha);
      alpha = std::fmax(2.0 * alpha, restart());
      continue;
    }
    if (rc) return rc;
    good_alpha = alpha;
    to_hat(p);
    const double pn = norm_h(p_h);
    double t = 0.0;
    rc = trf_slope_term(f, o, p, &t);
    if (rc) return rc;
    const double phi = pn - Delta, slope = -t / pn;
    if (phi < 0.0) hi = alpha;
    const double ratio = phi / slope;
    lo = std::fmax(lo, alpha - ratio);
    alpha -= (phi + Delta) * ratio / Delta;
    if (std::fabs(phi) < 0.01 * Delta) break;
  }
  if (!o.full_rank && !(alpha > 0.0)) alpha = good_alpha > 0.0 ? good_alpha : restart();
  int rc = trf_solve(f, o, alpha, p);
  if (rc == LSQAMD_ENOTPD && good_alpha >= 0.0 && good_alpha != alpha) {
    alpha = good_alpha;
    rc = trf_solve(f, o, alpha, p);
  }
  if (rc == LSQAMD_ENOTPD) FAIL(f, LSQAMD_ENOTPD, "trf: the shifted normal matrix is not positive definite");
  if (rc) return rc;
  to_hat(p);
  const double pn = norm_h(p_h);
  // B p_h = -g_h - alpha p_h before the rescaling onto the boundary
  double gp = 0.0;
  for (int64_t j = 0; j < P; ++j) gp += o.gh[j] * p_h[j];
  const double c = Delta / pn;
  *quad = c * c * (-gp - alpha * pn * pn);
  for (int64_t j = 0; j < P; ++j) p_h[j] *= c;
  *alpha_io = alpha;
  return 0;
}

// largest t with x + t s inside the box; hit[j] != 0 where that bound is reached first
double trf_to_bound(const lsqamd_fit *f, const std::vector<double> &x, const std::vector<double> &s,
                    std::vector<char> *hit) {
  const int64_t P = f->P;
  double t = INFINITY;
  std::vector<double> steps(P, INFINITY);
  for (int64_t j = 0; j < P; ++j) {
    if (s[j] == 0.0) continue;
    steps[j] = std::fmax((f->lb[j] - x[j]) / s[j], (f->ub[j] - x[j]) / s[j]);
    if (steps[j] < t) t = steps[j];
  }
  if (hit) {
    hit->assign(P, 0);
    for (int64_t j = 0; j < P; ++j) (*hit)[j] = (steps[j] == t && s[j] != 0.0) ? 1 : 0;
  }
  return t;
}

void quad_min_1d(double a, double b, double lo, double hi, double c, double *t_out, double *y_out) {
  double ts[3] = {lo, hi, 0.0};
  int n = 2;
  if (a != 0.0) {
    const double t0 = -0.5 * b / a;
    if (lo < t0 && t0 < hi) ts[n++] = t0;
  }
  int best = 0;
  double yb = ts[0] * (a * ts[0] + b) + c;
  for (int k = 1; k < n; ++k) {
    const double y = ts[k] * (a * ts[k] + b) + c;
    if (y < yb) { yb = y; best = k; }
  }
  *t_out = ts[best];
  *y_out = yb;
}

// s_h.B t_h given A (d t_h)
double trf_curv(const TrfOuter &o, const std::vector<double> &s_h, const std::vector<double> &t_h,
                const std::vector<double> &A_dt) {
  double q = 0.0;
  for (size_t j = 0; j < s_h.size(); ++j) q += o.d[j] * s_h[j] * A_dt[j] + s_h[j] * o.C[j] * t_h[j];
  return q;
}

// select_step: step (original variables), step_h, predicted reduction
int trf_choose_step(lsqamd_fit *f, TrfOuter &o, const std::vector<double> &p_h_in, double quad_pp, double Delta,
                    std::vector<double> &step, std::vector<double> &step_h, double *predicted) {
  const int64_t P = f->P;
  std::vector<double> p(P), p_h(p_h_in);
  bool inside = true;
  for (int64_t j = 0; j < P; ++j) {
    p[j] = o.d[j] * p_h[j];
    const double xn = f->hx[j] + p[j];
    if (!(xn >= f->lb[j] && xn <= f->ub[j])) inside = false;
  }
  const double g_p = dot_h(o.gh, p_h);
  if (inside) {
    step = p; step_h = p_h;
    *predicted = -(0.5 * quad_pp + g_p);
    return 0;
  }
  std::vector<char> hit;
  const double t_hit = trf_to_bound(f, f->hx, p, &hit);
  std::vector<double> r_h(p_h), r(P), x_hit(P), tmp(P), A_r(P);
  for (int64_t j = 0; j < P; ++j) {
    if (hit[j]) r_h[j] = -r_h[j];
    r[j] = o.d[j] * r_h[j];
    p[j] *= t_hit; p_h[j] *= t_hit;
    x_hit[j] = f->hx[j] + p[j];
  }
  const double pp = t_hit * t_hit * quad_pp, gp = t_hit * g_p;   // p_h.B p_h and g_h.p_h after the cut
  // exit of the reflected ray from the trust region (intersect_trust_region, larger root)
  double t_tr;
  {
    const double a = dot_h(r_h, r_h), b = dot_h(p_h, r_h), c = dot_h(p_h, p_h) - Delta * Delta;
    const double disc = std::sqrt(b * b - a * c);
    const double q = -(b + std::copysign(disc, b));
    t_tr = std::fmax(q / a, c / q);
  }
  const double t_box = trf_to_bound(f, x_hit, r, nullptr);
  const double t_r = std::fmin(t_box, t_tr);
  double r_lo = 0.0, r_hi = -1.0;
  if (t_r > 0.0) {
    r_lo = (1.0 - o.theta) * t_hit / t_r;
    r_hi = t_r == t_box ? o.theta * t_box : t_tr;
  }
  double r_value = INFINITY;
  if (r_lo <= r_hi) {
    for (int64_t j = 0; j < P; ++j) tmp[j] = o.d[j] * r_h[j];
    const int rc = symv_host(f, tmp.data(), A_r.data());
    if (rc) return rc;
    const double rr = trf_curv(o, r_h, r_h, A_r), pr = trf_curv(o, p_h, r_h, A_r);
    const double a = 0.5 * rr, b = dot_h(o.gh, r_h) + pr, c = 0.5 * pp + gp;
    double t;
    quad_min_1d(a, b, r_lo, r_hi, c, &t, &r_value);
    for (int64_t j = 0; j < P; ++j) {
      r_h[j] = p_h[j] + t * r_h[j];
      r[j] = r_h[j] * o.d[j];
    }
  }
  for (int64_t j = 0; j < P; ++j) { p[j] *= o.theta; p_h[j] *= o.theta; }   // strictly interior
  const double p_value = 0.5 * o.theta * o.theta * pp + o.theta * gp;
  // scaled anti-gradient
  std::vector<double> ag_h(P), ag(P);
  for (int64_t j = 0; j < P; ++j) { ag_h[j] = -o.gh[j]; ag[j] = o.d[j] * ag_h[j]; }
  if (!o.have_Aag) {
    o.Aag.assign(P, 0.0);
    const int rc = symv_host(f, ag.data(), o.Aag.data());
    if (rc) return rc;
    o.have_Aag = true;
  }
  const double t_tr_ag = Delta / norm_h(ag_h);
  const double t_box_ag = trf_to_bound(f, f->hx, ag, nullptr);
  const double t_max = t_box_ag < t_tr_ag ? o.theta * t_box_ag : t_tr_ag;
  double t_ag, ag_value;
  quad_min_1d(0.5 * trf_curv(o, ag_h, ag_h, o.Aag), dot_h(o.gh, ag_h), 0.0, t_max, 0.0, &t_ag, &ag_value);
  if (p_value < r_value && p_value < ag_value) {
    step = p; step_h = p_h; *predicted = -p_value;
  } else if (r_value < p_value && r_value < ag_value) {
    step = r; step_h = r_h; *predicted = -r_value;
  } else {
    for (int64_t j = 0; j < P; ++j) { ag[j] *= t_ag; ag_h[j] *= t_ag; }
    step = ag; step_h = ag_h; *predicted = -ag_value;
  }
  return 0;
}

// make_strictly_feasible
void trf_feasible(const lsqamd_fit *f, std::vector<double> &x, double rstep) {
  for (size_t j = 0; j < x.size(); ++j) {
    const double lb = f->lb[j], ub = f->ub[j];
    if (rstep == 0.0) {
      if (x[j] >= ub) x[j] = std::nextafter(ub, lb);
      else if (x[j] <= lb) x[j] = std::nextafter(lb, ub);
    } else {
      const double dlo = x[j] - lb, dhi = ub - x[j];
      const bool hi = std::isfinite(ub) && dhi <= std::fmin(dlo, rstep * std::fmax(1.0, std::fabs(ub)));
      const bool lo = std::isfinite(lb) && dlo <= std::fmin(dhi, rstep * std::fmax(1.0, std::fabs(lb)));
      if (hi) x[j] = ub - rstep * std::fmax(1.0, std::fabs(ub));
      else if (lo) x[j] = lb + rstep * std::fmax(1.0, std::fabs(lb));
    }
    if (x[j] < lb || x[j] > ub) x[j] = 0.5 * (lb + ub);
  }
}

// Coleman-Li scaling vector at (x, g): -> v, dv; returns |g v|_inf
double trf_cl_scaling(const lsqamd_fit *f, std::vector<double> &v, std::vector<double> &dv) {
  double gn = 0.0;
  for (int64_t j = 0; j < f->P; ++j) {
    v[j] = 1.0; dv[j] = 0.0;
    const double g = f->hg[j];
    if (g < 0.0 && std::isfinite(f->ub[j])) { v[j] = f->ub[j] - f->hx[j]; dv[j] = -1.0; }
    if (g > 0.0 && std::isfinite(f->lb[j])) { v[j] = f->hx[j] - f->lb[j]; dv[j] = 1.0; }
    gn = std::fmax(gn, std::fabs(g * v[j]));
  }
  return gn;
}

// trf_bounds / trf_no_bounds.  *status_out: 0 max_nfev, 1 gtol, 2 ftol, 3 xtol, 4 ftol and xtol
int run_trf(lsqamd_fit *f, const double *p0, int *status_out) {
  const int64_t P = f->P;
  if (f->opt.scaler == LSQAMD_SCALE_MARQUARDT)
    FAIL(f, LSQAMD_EINVAL, "trf: x_scale is 1 (scaler levenberg) or 'jac' (scaler more)");
  if (f->lb.empty()) { f->lb.assign(P, -INFINITY); f->ub.assign(P, INFINITY); }
  const double xtol = f->opt.xtol, gtol = f->opt.gtol, ftol = f->opt.ftol;
  const double eps = 2.220446049250313e-16;
  if (ftol < eps && xtol < eps && gtol < eps)
    FAIL(f, LSQAMD_EINVAL, "trf: at least one of the tolerances must be higher than machine epsilon");
  std::vector<double> x0(p0, p0 + P);
  for (int64_t j = 0; j < P; ++j)
    if (!(x0[j] >= f->lb[j] && x0[j] <= f->ub[j]))
      FAIL(f, LSQAMD_EINVAL, "trf: initial guess is outside of provided bounds (parameter %lld)", (long long)j);
  trf_feasible(f, x0, 1e-10);
  int rc = do_init(f, x0.data());     // f, J, A = J^T J, g, column norms at x0; nfev = njev = 1
  if (rc) return rc;
  const int max_nfev = f->opt.maxit;
  const bool jac_scale = f->opt.scaler == LSQAMD_SCALE_MORE;   // hdiag = 1 / scale (scale_init / scale_update)
  std::vector<double> v(P), dv(P), p_h(P), step(P), step_h(P), x_new(P);
  (void)trf_cl_scaling(f, v, dv);
  double Delta = 0.0;
  for (int64_t j = 0; j < P; ++j) {
    if (dv[j] != 0.0) v[j] *= f->hdiag[j];
    const double t = f->hx[j] * f->hdiag[j] / std::sqrt(v[j]);
    Delta += t * t;
  }
  Delta = std::sqrt(Delta);
  if (Delta == 0.0) Delta = 1.0;
  double alpha = 0.0;
  int status = -1;
  TrfOuter o;
  o.d.resize(P); o.C.resize(P); o.gh.resize(P);
  const bool trace = std::getenv("LSQAMD_TRF_TRACE") != nullptr;   // developer knob
  while (true) {
    const double g_norm = trf_cl_scaling(f, v, dv);
    if (trace) fprintf(stderr, "trf nfev %d cost %.12e optimality %.3e Delta %.3e alpha %.3e\n", f->nfev, 0.5 * f->chi2, g_norm, Delta, alpha);
    if (g_norm < gtol) status = 1;
    if (status >= 0 || f->nfev >= max_nfev) break;
    for (int64_t j = 0; j < P; ++j) {
      if (dv[j] != 0.0) v[j] *= f->hdiag[j];
      o.d[j] = std::sqrt(v[j]) / f->hdiag[j];
      o.C[j] = f->hg[j] * dv[j] / f->hdiag[j];
      o.gh[j] = o.d[j] * f->hg[j];
    }
    o.have_Aag = false;
    o.gn_tried = false;
    o.theta = std::fmax(0.995, 1.0 - g_norm);
    const double cost = 0.5 * f->chi2;
    double actual = -1.0, cost_new = cost;
    while (actual <= 0.0 && f->nfev < max_nfev) {
      double quad = 0.0, predicted = 0.0;
      rc = trf_subproblem(f, o, Delta, &alpha, p_h, &quad);
      if (rc) return rc;
      rc = trf_choose_step(f, o, p_h, quad, Delta, step, step_h, &predicted);
      if (rc) return rc;
      for (int64_t j = 0; j < P; ++j) x_new[j] = f->hx[j] + step[j];
      trf_feasible(f, x_new, 0.0);
      std::memcpy(f->pin_x, x_new.data(), sizeof(double) * P);
      HIPCHK(f, hipMemcpyAsync(f->p_trial, f->pin_x, sizeof(double) * P, hipMemcpyHostToDevice, f->st));
      double chi2_new = 0.0;
      rc = eval_residual_dev(f, f->p_trial, &chi2_new);
      if (rc) return rc;
      const double sh_norm = norm_h(step_h);
      if (!std::isfinite(chi2_new)) {
        Delta = 0.25 * sh_norm;
        continue;
      }
      cost_new = 0.5 * chi2_new;
      actual = cost - cost_new;
      double ratio;
      if (predicted > 0.0) ratio = actual / predicted;
      else if (predicted == 0.0 && actual == 0.0) ratio = 1.0;
      else ratio = 0.0;
      double Delta_new = Delta;
      if (ratio < 0.25) Delta_new = 0.25 * sh_norm;
      else if (ratio > 0.75 && sh_norm > 0.95 * Delta) Delta_new = 2.0 * Delta;
      const bool f_ok = actual < ftol * cost && ratio > 0.25;
      const bool x_ok = norm_h(step) < xtol * (xtol + norm_h(f->hx));
      if (trace) fprintf(stderr, "    trial: actual %.3e predicted %.3e ratio %.3f |step_h| %.3e Delta %.3e -> %.3e\n", actual, predicted, ratio, sh_norm, Delta, Delta_new);
      if (f_ok && x_ok) status = 4;
      else if (f_ok) status = 2;
      else if (x_ok) status = 3;
      if (status >= 0) break;
      alpha *= Delta / Delta_new;
      Delta = Delta_new;
    }
    if (actual > 0.0) {
      rc = eval_normal_dev(f, f->p_trial);
      if (rc) return rc;
      f->hx = x_new;
      f->hdx = step;
      std::swap(f->p_dev, f->p_trial);
      if (jac_scale) scale_update(f);
    }
  }
  *status_out = status < 0 ? 0 : status;
  return 0;
}

// ---- dogbox: dogleg in a rectangular trust region with an active set ----------
// scipy_least_squares' method='dogbox' (src/lsqfit/_scipy.py:62-63; scipy optimize/_lsq/dogbox.py).
// Per outer iteration the device does ONE factorisation -- the Gauss-Newton step of the free
// parameters, (A_ff) n = -g_f, as the full system with the active rows/columns replaced by the
// identity -- and one A g product for the Cauchy step; every trial step lies in span(n, g), so its
// predicted reduction follows from three scalars (n.A n = -n.g, n.A g = -g.g, g.A g) and a
// trial costs one residual evaluation.
void dogbox_step(const std::vector<double> &x, const std::vector<double> &newton, const std::vector<double> &g,
                 const std::vector<char> &free_set, double a, double b, double Delta,
                 const std::vector<double> &scale_inv, const std::vector<double> &lb, const std::vector<double> &ub,
                 std::vector<double> &step, std::vector<int> &lands, bool *tr_hit, double *alpha_n, double *beta_g) {
  const size_t P = x.size();
  std::vector<double> lo_t(P), hi_t(P), lo_c(P), hi_c(P), trb(P);
  bool newton_inside = true;
  for (size_t j = 0; j < P; ++j) {
    lands[j] = 0;
    step[j] = 0.0;
    if (!free_set[j]) continue;
    trb[j] = Delta / scale_inv[j];
    lo_c[j] = lb[j] - x[j]; hi_c[j] = ub[j] - x[j];
    lo_t[j] = std::fmax(lo_c[j], -trb[j]); hi_t[j] = std::fmin(hi_c[j], trb[j]);
    if (!(newton[j] >= lo_t[j] && newton[j] <= hi_t[j])) newton_inside = false;
  }
  *tr_hit = false;
  if (newton_inside) {
    for (size_t j = 0; j < P; ++j) if (free_set[j]) step[j] = newton[j];
    *alpha_n = 1.0; *beta_g = 0.0;
    return;
  }
  // largest t with t * (-g) inside the region
  double t_max = INFINITY;
  for (size_t j = 0; j < P; ++j) {
    if (!free_set[j] || g[j] == 0.0) continue;
    const double s = -g[j];
    t_max = std::fmin(t_max, std::fmax(lo_t[j] / s, hi_t[j] / s));
  }
  double t_c, y;
  quad_min_1d(a, b, 0.0, t_max, 0.0, &t_c, &y);
  // from the Cauchy point towards the Newton point until the region's edge
  double t = INFINITY;
  std::vector<double> steps(P, INFINITY);
  for (size_t j = 0; j < P; ++j) {
    if (!free_set[j]) continue;
    const double c = -t_c * g[j], dj = newton[j] - c;
    if (dj == 0.0) continue;
    steps[j] = std::fmax((lo_t[j] - c) / dj, (hi_t[j] - c) / dj);
    if (steps[j] < t) t = steps[j];
  }
  for (size_t j = 0; j < P; ++j) {
    if (!free_set[j]) continue;
    const double c = -t_c * g[j], dj = newton[j] - c;
    step[j] = c + t * dj;
    if (steps[j] == t && dj != 0.0) {
      if (dj < 0.0) {
        if (lo_t[j] == lo_c[j]) lands[j] = -1;
        if (lo_t[j] == -trb[j]) *tr_hit = true;
      } else {
        if (hi_t[j] == hi_c[j]) lands[j] = 1;
        if (hi_t[j] == trb[j]) *tr_hit = true;
      }
    }
  }
  *alpha_n = t;
  *beta_g = -t_c * (1.0 - t);
}

int run_dogbox(lsqamd_fit *f, const double *p0, int *status_out) {
  const int64_t P = f->P;
  if (f->opt.scaler == LSQAMD_SCALE_MARQUARDT)
    FAIL(f, LSQAMD_EINVAL, "dogbox: x_scale is 1 (scaler levenberg) or 'jac' (scaler more)");
  if (f->lb.empty()) { f->lb.assign(P, -INFINITY); f->ub.assign(P, INFINITY); }
  const double xtol = f->opt.xtol, gtol = f->opt.gtol, ftol = f->opt.ftol;
  const double eps = 2.220446049250313e-16;
  if (ftol < eps && xtol < eps && gtol < eps)
    FAIL(f, LSQAMD_EINVAL, "dogbox: at least one of the tolerances must be higher than machine epsilon");
  for (int64_t j = 0; j < P; ++j)
    if (!(p0[j] >= f->lb[j] && p0[j] <= f->ub[j]))
      FAIL(f, LSQAMD_EINVAL, "dogbox: initial guess is outside of provided bounds (parameter %lld)", (long long)j);
  int rc = do_init(f, p0);
  if (rc) return rc;
  const int max_nfev = f->opt.maxit;
  const bool jac_scale = f->opt.scaler == LSQAMD_SCALE_MORE;   // hdiag = 1 / scale
  double Delta = 0.0;
  for (int64_t j = 0; j < P; ++j) Delta = std::fmax(Delta, std::fabs(f->hx[j] * f->hdiag[j]));
  if (Delta == 0.0) Delta = 1.0;
  std::vector<int> on_bound(P, 0), lands(P, 0);
  for (int64_t j = 0; j < P; ++j) {
    if (f->hx[j] == f->lb[j]) on_bound[j] = -1;
    if (f->hx[j] == f->ub[j]) on_bound[j] = 1;
  }
  std::vector<char> free_set(P, 1);
  std::vector<double> gz(P), frozen(P), newton(P), Ag(P), step(P), x_new(P);
  int status = -1;
  while (true) {
    bool any_active = false;
    double g_norm = 0.0;
    for (int64_t j = 0; j < P; ++j) {
      const bool active = on_bound[j] * f->hg[j] < 0.0;
      free_set[j] = !active;
      frozen[j] = active ? 1.0 : 0.0;
      any_active |= active;
      gz[j] = active ? 0.0 : f->hg[j];
      g_norm = std::fmax(g_norm, std::fabs(gz[j]));
    }
    if (g_norm < gtol) status = 1;
    if (status >= 0 || f->nfev >= max_nfev) break;
    rc = solve_damped_dev(f, 0.0, nullptr, any_active ? frozen.data() : nullptr);
    if (rc == LSQAMD_ENOTPD)
      FAIL(f, LSQAMD_ENOTPD, "dogbox: J^T J of the free parameters is not positive definite (rank-deficient Jacobian)");
    if (rc) return rc;
    for (int64_t j = 0; j < P; ++j) newton[j] = free_set[j] ? -f->hv[j] : 0.0;
    rc = symv_host(f, gz.data(), Ag.data());
    if (rc) return rc;
    const double gAg = dot_h(gz, Ag), gg = dot_h(gz, gz), ng = dot_h(newton, gz);
    const double a = 0.5 * gAg, b = -gg;
    const double nAn = -ng, nAg = -gg;      // (A n)_free = -g_free
    const double cost = 0.5 * f->chi2;
    double actual = -1.0;
    while (actual <= 0.0 && f->nfev < max_nfev) {
      bool tr_hit = false;
      double al = 0.0, be = 0.0;
      dogbox_step(f->hx, newton, gz, free_set, a, b, Delta, f->hdiag, f->lb, f->ub, step, lands, &tr_hit, &al, &be);
      // step = al n + be g on the free set
      const double sAs = al * al * nAn + 2.0 * al * be * nAg + be * be * gAg;
      const double predicted = -(0.5 * sAs + al * ng + be * gg);
      double sh_norm = 0.0, s_norm = 0.0;
      for (int64_t j = 0; j < P; ++j) {
        x_new[j] = std::fmin(std::fmax(f->hx[j] + step[j], f->lb[j]), f->ub[j]);
        sh_norm = std::fmax(sh_norm, std::fabs(step[j] * f->hdiag[j]));
        s_norm += step[j] * step[j];
      }
      s_norm = std::sqrt(s_norm);
      std::memcpy(f->pin_x, x_new.data(), sizeof(double) * P);
      HIPCHK(f, hipMemcpyAsync(f->p_trial, f->pin_x, sizeof(double) * P, hipMemcpyHostToDevice, f->st));
      double chi2_new = 0.0;
      rc = eval_residual_dev(f, f->p_trial, &chi2_new);
      if (rc) return rc;
      if (!std::isfinite(chi2_new)) {
        Delta = 0.25 * sh_norm;
        continue;
      }
      actual = cost - 0.5 * chi2_new;
      double ratio;
      if (predicted > 0.0) ratio = actual / predicted;
      else if (predicted == 0.0 && actual == 0.0) ratio = 1.0;
      else ratio = 0.0;
      if (ratio < 0.25) Delta = 0.25 * sh_norm;
      else if (ratio > 0.75 && tr_hit) Delta *= 2.0;
      const bool f_ok = actual < ftol * cost && ratio > 0.25;
      const bool x_ok = s_norm < xtol * (xtol + norm_h(f->hx));
      if (f_ok && x_ok) status = 4;
      else if (f_ok) status = 2;
      else if (x_ok) status = 3;
      if (status >= 0) break;
    }
    if (actual > 0.0) {
      for (int64_t j = 0; j < P; ++j) {
        if (free_set[j]) on_bound[j] = lands[j];
        if (on_bound[j] == -1) x_new[j] = f->lb[j];
        if (on_bound[j] == 1) x_new[j] = f->ub[j];
      }
      std::memcpy(f->pin_x, x_new.data(), sizeof(double) * P);   // variables set exactly on their walls
      HIPCHK(f, hipMemcpyAsync(f->p_trial, f->pin_x, sizeof(double) * P, hipMemcpyHostToDevice, f->st));
      rc = eval_normal_dev(f, f->p_trial);
      if (rc) return rc;
      f->hx = x_new;
      f->hdx = step;
      std::swap(f->p_dev, f->p_trial);
      if (jac_scale) scale_update(f);
    }
  }
  *status_out = status < 0 ? 0 : status;
  return 0;
}

// ---- MINPACK's lmder (scipy_least_squares' method='lm') ------------------------
// src/lsqfit/_scipy.py:64-67: scipy hands method 'lm' to MINPACK (lmder.f / lmpar.f, factor = 100,
// diag = 1/x_scale or MINPACK's own column-norm scaling).  qrfac / qrsolv only ever deliver the
// solution of [J; sqrt(par) D] x = [f; 0] and triangular solves with its R factor -- functions of
// A = J^T J, g = J^T f and D -- so lmpar runs on the device's Cholesky: per Newton step on the
// secular equation one factorisation of A + par D^2, the solve, and a second solve for
// q.(A + par D^2)^-1 q with q = D^2 x / |D x|; |J p|^2 comes from the system p satisfies.
struct LmparOuter {   // par = 0 quantities, fixed while the Jacobian is
  bool tried = false, full_rank = false;
  std::vector<double> x_gn;
  double dx_gn = 0.0, form_gn = 0.0;   // |D x_gn| and q.A^-1 q
};

// q.(A + par D^2)^-1 q, q = D^2 x / |D x|, factor of the matrix in place
int lmpar_form(lsqamd_fit *f, const std::vector<double> &x, double dxnorm, double *out) {
  const int64_t P = f->P;
  std::vector<double> q(P), z(P);
  for (int64_t j = 0; j < P; ++j) q[j] = f->hdiag[j] * f->hdiag[j] * x[j] / dxnorm;
  const int rc = solve_with_factor(f, q.data(), z.data());
  if (rc) return rc;
  *out = dot_h(q, z);
  return 0;
}

// lmpar: x with |D x| within 10 % of delta (or the Gauss-Newton step when that is shorter)
int lmpar_dev(lsqamd_fit *f, LmparOuter &o, double delta, double *par_io, std::vector<double> &x) {
  const int64_t P = f->P;
  const double dwarf = 2.2250738585072014e-308;
  double par = *par_io;
  if (!o.tried) {
    o.tried = true;
    const int rc = solve_damped_dev(f, 0.0, f->hdiag.data());
    if (rc < 0 && rc != LSQAMD_ENOTPD) return rc;
    o.full_rank = rc == 0;
    if (o.full_rank) {
      o.x_gn = f->hv;
      o.dx_gn = scaled_norm(f->hdiag, o.x_gn);
      const int rc2 = lmpar_form(f, o.x_gn, o.dx_gn, &o.form_gn);
      if (rc2) return rc2;
    }
  }
  double dxnorm = INFINITY, fp = INFINITY, parl = 0.0;
  if (o.full_rank) {
    dxnorm = o.dx_gn;
    fp = dxnorm - delta;
    if (fp <= 0.1 * delta) {
      x = o.x_gn;
      *par_io = 0.0;
      return 0;
    }
    parl = (fp / delta) / o.form_gn;
  }
  double gn = 0.0;
  for (int64_t j = 0; j < P; ++j) { const double t = f->hg[j] / f->hdiag[j]; gn += t * t; }
  gn = std::sqrt(gn);
  double paru = gn / delta;
  if (paru == 0.0) paru = dwarf / std::fmin(delta, 0.1);
  par = std::fmin(std::fmax(par, parl), paru);
  if (par == 0.0) par = gn / dxnorm;
  for (int it = 1; it <= 10; ++it) {
    if (par == 0.0) par = std::fmax(dwarf, 0.001 * paru);
    int rc = solve_damped_dev(f, par, f->hdiag.data());
    if (rc == LSQAMD_ENOTPD) FAIL(f, LSQAMD_ENOTPD, "minpack lm: J^T J + par D^2 is not positive definite (par %.3e)", par);
    if (rc) return rc;
    x = f->hv;
    dxnorm = scaled_norm(f->hdiag, x);
    const double prev = fp;
    fp = dxnorm - delta;
    if (std::fabs(fp) <= 0.1 * delta || (parl == 0.0 && fp <= prev && prev < 0.0) || it == 10) break;
    double form = 0.0;
    rc = lmpar_form(f, x, dxnorm, &form);
    if (rc) return rc;
    const double parc = (fp / delta) / form;
    if (fp > 0.0) parl = std::fmax(parl, par);
    if (fp < 0.0) paru = std::fmin(paru, par);
    par = std::fmax(parl, par + parc);
  }
  *par_io = par;
  return 0;
}

// lmder.  *status_out in scipy's numbering (FROM_MINPACK_TO_COMMON): info 1, 2, 3, 4, 5 -> 2, 3, 4, 1, 0
int run_minpack(lsqamd_fit *f, const double *p0, int *status_out) {
  const int64_t P = f->P;
  if (f->opt.scaler == LSQAMD_SCALE_MARQUARDT)
    FAIL(f, LSQAMD_EINVAL, "minpack lm: x_scale is 1 (scaler levenberg) or 'jac' (scaler more)");
  const double xtol = f->opt.xtol, gtol = f->opt.gtol, ftol = f->opt.ftol;
  const double epsmch = 2.220446049250313e-16, factor = 100.0;
  if (ftol < epsmch || xtol < epsmch || gtol < epsmch)
    FAIL(f, LSQAMD_EINVAL, "minpack lm: all tolerances must be higher than machine epsilon");
  if (!f->lb.empty())
    for (int64_t j = 0; j < P; ++j)
      if (std::isfinite(f->lb[j]) || std::isfinite(f->ub[j]))
        FAIL(f, LSQAMD_EINVAL, "minpack lm: method 'lm' doesn't support bounds");
  int rc = do_init(f, p0);   // fvec, J, A, g, column norms, D (mode 1: column norms, 1 where 0); nfev = njev = 1
  if (rc) return rc;
  const int maxfev = f->opt.maxit;
  const bool mode1 = f->opt.scaler == LSQAMD_SCALE_MORE;
  double fnorm = std::sqrt(f->chi2);
  double xnorm = scaled_norm(f->hdiag, f->hx);
  double delta = factor * xnorm;
  if (delta == 0.0) delta = factor;
  double par = 0.0;
  int iter = 1, info = 0;
  std::vector<double> p(P), xt(P);
  while (true) {
    double gnorm = 0.0;
    if (fnorm != 0.0)
      for (int64_t j = 0; j < P; ++j)
        if (f->hcoln[j] != 0.0) gnorm = std::fmax(gnorm, std::fabs(f->hg[j] / fnorm / f->hcoln[j]));
    if (gnorm <= gtol) { info = 4; break; }
    LmparOuter o;
    while (true) {
      rc = lmpar_dev(f, o, delta, &par, p);
      if (rc) return rc;
      double pg = 0.0;
      for (int64_t j = 0; j < P; ++j) { xt[j] = f->hx[j] - p[j]; pg += p[j] * f->hg[j]; }
      const double pnorm = scaled_norm(f->hdiag, p);
      if (iter == 1) delta = std::fmin(delta, pnorm);
      std::memcpy(f->pin_x, xt.data(), sizeof(double) * P);
      HIPCHK(f, hipMemcpyAsync(f->p_trial, f->pin_x, sizeof(double) * P, hipMemcpyHostToDevice, f->st));
      double chi2_t = 0.0;
      rc = eval_residual_dev(f, f->p_trial, &chi2_t);
      if (rc) return rc;
      const double fnorm1 = std::sqrt(chi2_t);     // NaN compares false everywhere below: rejected
      double actred = -1.0;
      if (0.1 * fnorm1 < fnorm) actred = 1.0 - (fnorm1 / fnorm) * (fnorm1 / fnorm);
      // |J p|^2 = p.g - par |D p|^2 since (A + par D^2) p = g
      const double jp2 = std::fmax(pg - par * pnorm * pnorm, 0.0);
      const double t1sq = jp2 / (fnorm * fnorm), t2sq = par * pnorm * pnorm / (fnorm * fnorm);
      const double prered = t1sq + t2sq / 0.5, dirder = -(t1sq + t2sq);
      const double ratio = prered != 0.0 ? actred / prered : 0.0;
      if (ratio <= 0.25) {
        double temp = actred >= 0.0 ? 0.5 : 0.5 * dirder / (dirder + 0.5 * actred);
        if (0.1 * fnorm1 >= fnorm || temp < 0.1) temp = 0.1;
        delta = temp * std::fmin(delta, pnorm / 0.1);
        par /= temp;
      } else if (par == 0.0 || ratio >= 0.75) {
        delta = pnorm / 0.5;
        par *= 0.5;
      }
      const bool accepted = ratio >= 1e-4;
      if (accepted) {
        rc = eval_normal_dev(f, f->p_trial);    // also the Jacobian of the next outer pass
        if (rc) return rc;
        f->hx = xt;
        for (int64_t j = 0; j < P; ++j) f->hdx[j] = -p[j];
        std::swap(f->p_dev, f->p_trial);
        xnorm = scaled_norm(f->hdiag, f->hx);   // with the D of this pass (lmder.f), before its update
        if (mode1) scale_update(f);
        fnorm = fnorm1;
        ++iter;
      }
      const bool small = std::fabs(actred) <= ftol && prered <= ftol && 0.5 * ratio <= 1.0;
      if (small) info = 1;
      if (delta <= xtol * xnorm) info = 2;
      if (small && info == 2) info = 3;
      if (info != 0) break;
      if (f->nfev >= maxfev) info = 5;
      if (std::fabs(actred) <= epsmch && prered <= epsmch && 0.5 * ratio <= 1.0) info = 6;
      if (delta <= epsmch * xnorm) info = 7;
      if (gnorm <= epsmch) info = 8;
      if (info != 0 || accepted) break;
    }
    if (info != 0) break;
  }
  // lmder stops BEFORE evaluating the Jacobian at an accepted final point; here it is already in
  // place (one evaluation more than MINPACK counts -- the caller needs J there anyway, _scipy.py:160)
  static const int to_scipy[9] = {-1, 2, 3, 4, 1, 0, 2, 3, 1};   // 6, 7, 8: the tests 1, 2, 4 at machine precision
  *status_out = to_scipy[info];
  return 0;
}

}  // namespace lsqamd_host
