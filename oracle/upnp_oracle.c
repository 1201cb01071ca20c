/*
 * ORACLE — test infrastructure only.
 * CPU restatement of uncertainty-PnP, core/csrc/uncertainty_pnp/src/uncertainty_pnp.cpp:
 *   cost functor ReprojectionErrorArray::operator() :16-34
 *   problem set-up + ceres::Solve with default options + DENSE_SCHUR :61-92
 *
 * The solver itself lives in a third-party dependency that is NOT in the tree as
 * a library: Ceres Solver (headers 2.0.0 vendored at include/ceres/version.h:34-36,
 * libceres.so absent, build_ceres.sh:3 names 1.14.0).  Its published algorithm is
 * restated here:
 *   - automatic differentiation by dual numbers with 6 infinitesimals (ceres/jet.h),
 *   - AngleAxisRotatePoint incl. the small-angle branch (ceres/rotation.h),
 *   - Levenberg-Marquardt trust-region minimizer with Ceres' defaults:
 *     max 50 iterations, initial radius 1e4, max radius 1e16, min radius 1e-32,
 *     min_relative_decrease 1e-3, LM diagonal clamp [1e-6, 1e32], Jacobi
 *     column scaling 1/(1+||J_i||), radius update r/max(1/3, 1-(2q-1)^3),
 *     rejected step: r/=d, d*=2; invalid step: r*=0.5 (max 5 in a row);
 *     termination on parameter (1e-8), function (1e-6) and gradient (1e-10)
 *     tolerances in the order the minimizer tests them.  A one-parameter-block
 *     problem has no Schur complement, so DENSE_SCHUR reduces to a dense
 *     factorisation of the 6x6 damped system; normal equations + Cholesky here.
 * The cost/Jacobian part is pinned in this container against the vendored
 * ceres/jet.h + ceres/rotation.h (oracle/_ref/upnp_ref, built by oracle/build_ref.py);
 * the minimizer is PARITY UNPINNED against libceres (not buildable here: needs
 * cmake + glog + Eigen install) — see DESIGN.md.
 */
#include <math.h>
#include <string.h>

#define NP 6
typedef struct { double v; double d[NP]; } jet;

static jet jc(double c) { jet r; r.v = c; memset(r.d, 0, sizeof r.d); return r; }
static jet jvar(double c, int k) { jet r = jc(c); r.d[k] = 1.0; return r; }
static jet jadd(jet a, jet b) { jet r; r.v = a.v + b.v; for (int i = 0; i < NP; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
static jet jsub(jet a, jet b) { jet r; r.v = a.v - b.v; for (int i = 0; i < NP; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
static jet jmul(jet a, jet b) { jet r; r.v = a.v * b.v; for (int i = 0; i < NP; ++i) r.d[i] = a.v * b.d[i] + a.d[i] * b.v; return r; }
static jet jdiv(jet a, jet b) { /* jet.h: h = 1/g; f*h, (f' - f*h*g')*h */
  jet r; double inv = 1.0 / b.v; double q = a.v * inv; r.v = q;
  for (int i = 0; i < NP; ++i) r.d[i] = (a.d[i] - q * b.d[i]) * inv; return r; }
static jet jsqrt(jet a) { jet r; r.v = sqrt(a.v); double t = 1.0 / (2.0 * r.v); for (int i = 0; i < NP; ++i) r.d[i] = t * a.d[i]; return r; }
static jet jcos(jet a) { jet r; r.v = cos(a.v); double s = -sin(a.v); for (int i = 0; i < NP; ++i) r.d[i] = s * a.d[i]; return r; }
static jet jsin(jet a) { jet r; r.v = sin(a.v); double c = cos(a.v); for (int i = 0; i < NP; ++i) r.d[i] = c * a.d[i]; return r; }

/* ceres/rotation.h AngleAxisRotatePoint */
static void angle_axis_rotate_point(const jet aa[3], const jet pt[3], jet out[3]) {
  jet theta2 = jadd(jadd(jmul(aa[0], aa[0]), jmul(aa[1], aa[1])), jmul(aa[2], aa[2]));
  if (theta2.v > 2.220446049250313e-16) {
    jet theta = jsqrt(theta2), costheta = jcos(theta), sintheta = jsin(theta);
    jet theta_inverse = jdiv(jc(1.0), theta);
    jet w[3] = {jmul(aa[0], theta_inverse), jmul(aa[1], theta_inverse), jmul(aa[2], theta_inverse)};
    jet wxp[3] = {jsub(jmul(w[1], pt[2]), jmul(w[2], pt[1])), jsub(jmul(w[2], pt[0]), jmul(w[0], pt[2])),
                  jsub(jmul(w[0], pt[1]), jmul(w[1], pt[0]))};
    jet tmp = jmul(jadd(jadd(jmul(w[0], pt[0]), jmul(w[1], pt[1])), jmul(w[2], pt[2])), jsub(jc(1.0), costheta));
    for (int i = 0; i < 3; ++i) out[i] = jadd(jadd(jmul(pt[i], costheta), jmul(wxp[i], sintheta)), jmul(w[i], tmp));
  } else {
    jet wxp[3] = {jsub(jmul(aa[1], pt[2]), jmul(aa[2], pt[1])), jsub(jmul(aa[2], pt[0]), jmul(aa[0], pt[2])),
                  jsub(jmul(aa[0], pt[1]), jmul(aa[1], pt[0]))};
    for (int i = 0; i < 3; ++i) out[i] = jadd(pt[i], wxp[i]);
  }
}

/* residuals r[2] and jacobian J[2][6] of one observation (uncertainty_pnp.cpp:16-34) */
void oracle_upnp_residual(const double* pose, const double* p2, const double* p3, const double* w, const double* K,
                          double* r, double* J) {
  jet ps[6];
  for (int k = 0; k < 6; ++k) ps[k] = jvar(pose[k], k);
  jet pt[3] = {jc(p3[0]), jc(p3[1]), jc(p3[2])}, tp[3];
  angle_axis_rotate_point(ps, pt, tp);
  tp[0] = jadd(tp[0], ps[3]); tp[1] = jadd(tp[1], ps[4]); tp[2] = jadd(tp[2], ps[5]);
  double fx = K[0], fy = K[4], px = K[2], py = K[5];
  jet proj_x = jadd(jdiv(jmul(jc(fx), tp[0]), tp[2]), jc(px));
  jet proj_y = jadd(jdiv(jmul(jc(fy), tp[1]), tp[2]), jc(py));
  jet dx = jsub(proj_x, jc(p2[0])), dy = jsub(proj_y, jc(p2[1]));
  jet r0 = jadd(jmul(jc(w[0]), dx), jmul(jc(w[1]), dy));
  jet r1 = jadd(jmul(jc(w[1]), dx), jmul(jc(w[2]), dy));
  r[0] = r0.v; r[1] = r1.v;
  for (int k = 0; k < 6; ++k) { J[k] = r0.d[k]; J[6 + k] = r1.d[k]; }
}

/* cost = 0.5||r||^2; H = J^T J (unscaled), g = J^T r */
static double evaluate(const double* x, const double* pts2d, const double* pts3d, const double* wgt, const double* K,
                       int pn, double* H, double* g) {
  double cost = 0.0;
  if (H) memset(H, 0, sizeof(double) * 36);
  if (g) memset(g, 0, sizeof(double) * 6);
  for (int i = 0; i < pn; ++i) {
    double r[2], J[12];
    oracle_upnp_residual(x, pts2d + 2 * i, pts3d + 3 * i, wgt + 3 * i, K, r, J);
    cost += r[0] * r[0] + r[1] * r[1];
    if (H)
      for (int a = 0; a < 6; ++a) {
        g[a] += J[a] * r[0] + J[6 + a] * r[1];
        for (int b = 0; b < 6; ++b) H[a * 6 + b] += J[a] * J[b] + J[6 + a] * J[6 + b];
      }
  }
  cost = 0.5 * cost;
  if (!isfinite(cost)) cost = 1.7976931348623157e308; /* evaluation failure -> step rejected */
  return cost;
}

/* Cholesky solve of the SPD 6x6 system A y = b; returns 0 on failure */
static int chol_solve6(const double* A, const double* b, double* y) {
  double L[36];
  memset(L, 0, sizeof L);
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = A[i * 6 + j];
      for (int k = 0; k < j; ++k) s -= L[i * 6 + k] * L[j * 6 + k];
      if (i == j) {
        if (!(s > 0.0)) return 0;
        L[i * 6 + i] = sqrt(s);
      } else {
        L[i * 6 + j] = s / L[j * 6 + j];
      }
    }
  double z[6];
  for (int i = 0; i < 6; ++i) {
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= L[i * 6 + k] * z[k];
    z[i] = s / L[i * 6 + i];
  }
  for (int i = 5; i >= 0; --i) {
    double s = z[i];
    for (int k = i + 1; k < 6; ++k) s -= L[k * 6 + i] * y[k];
    y[i] = s / L[i * 6 + i];
  }
  for (int i = 0; i < 6; ++i)
    if (!isfinite(y[i])) return 0;
  return 1;
}

/* info[0] = iterations, info[1] = termination: 0 convergence(param) 1 (function) 2 (gradient)
 * 3 max iterations 4 min radius 5 too many invalid steps */
void oracle_uncertainty_pnp(const double* pts2d, const double* pts3d, const double* wgt2d, const double* K,
                            const double* init_rt, double* result_rt, int pn, int* info) {
  const int max_it = 50;
  const double min_rel_dec = 1e-3, ftol = 1e-6, gtol = 1e-10, ptol = 1e-8;
  const double min_diag = 1e-6, max_diag = 1e32, max_radius = 1e16, min_radius = 1e-32;
  double x[6], H[36], g[6], scale[6];
  memcpy(x, init_rt, sizeof x);
  double radius = 1e4, decrease_factor = 2.0;
  int invalid = 0, iter = 0, term = 3;

  double cost = evaluate(x, pts2d, pts3d, wgt2d, K, pn, H, g);
  for (int i = 0; i < 6; ++i) scale[i] = 1.0 / (1.0 + sqrt(H[i * 6 + i]));
  double gmax = 0.0;
  for (int i = 0; i < 6; ++i) gmax = fmax(gmax, fabs(g[i]));
  int reuse_diag = 0;
  double diag[6] = {0, 0, 0, 0, 0, 0};
  if (gmax <= gtol) { term = 2; goto done; }

  for (;;) {
    /* FinalizeIterationAndCheckIfMinimizerCanContinue */
    if (iter >= max_it) { term = 3; break; }
    if (gmax <= gtol) { term = 2; break; }
    if (radius < min_radius) { term = 4; break; }
    ++iter;
    /* scaled system: Hs = S H S, gs = S g */
    double Hs[36], gs[6], A[36], y[6], step[6];
    for (int a = 0; a < 6; ++a) {
      gs[a] = scale[a] * g[a];
      for (int b = 0; b < 6; ++b) Hs[a * 6 + b] = scale[a] * H[a * 6 + b] * scale[b];
    }
    if (!reuse_diag)
      for (int a = 0; a < 6; ++a) diag[a] = fmin(fmax(Hs[a * 6 + a], min_diag), max_diag);
    memcpy(A, Hs, sizeof A);
    for (int a = 0; a < 6; ++a) A[a * 6 + a] += diag[a] / radius; /* (sqrt(diag/radius))^2 */
    reuse_diag = 1;
    int ok = chol_solve6(A, gs, y);
    double model_change = 0.0;
    if (ok) {
      for (int a = 0; a < 6; ++a) step[a] = -y[a];
      /* -(J s)^T (f + J s / 2) = -s^T gs - s^T Hs s / 2 */
      double sg = 0.0, sHs = 0.0;
      for (int a = 0; a < 6; ++a) {
        sg += step[a] * gs[a];
        double t = 0.0;
        for (int b = 0; b < 6; ++b) t += Hs[a * 6 + b] * step[b];
        sHs += step[a] * t;
      }
      model_change = -sg - 0.5 * sHs;
    }
    if (!ok || !(model_change > 0.0)) {
      if (++invalid >= 5) { term = 5; break; }
      radius *= 0.5; reuse_diag = 0;
      continue;
    }
    invalid = 0;
    double cand[6], delta2 = 0.0, xn2 = 0.0;
    for (int a = 0; a < 6; ++a) {
      double d = step[a] * scale[a];
      cand[a] = x[a] + d;
      delta2 += (x[a] - cand[a]) * (x[a] - cand[a]);
      xn2 += x[a] * x[a];
    }
    double Hc[36], gc[6];
    double cand_cost = evaluate(cand, pts2d, pts3d, wgt2d, K, pn, Hc, gc);
    if (sqrt(delta2) <= ptol * (sqrt(xn2) + ptol)) { term = 0; break; }
    if (fabs(cost - cand_cost) <= ftol * cost) { term = 1; break; }
    double rel_dec = (cost - cand_cost) / model_change;
    if (rel_dec > min_rel_dec) {
      memcpy(x, cand, sizeof x); memcpy(H, Hc, sizeof H); memcpy(g, gc, sizeof g);
      cost = cand_cost;
      gmax = 0.0;
      for (int i = 0; i < 6; ++i) gmax = fmax(gmax, fabs(g[i]));
      double t = 2.0 * rel_dec - 1.0;
      radius = radius / fmax(1.0 / 3.0, 1.0 - t * t * t);
      radius = fmin(max_radius, radius);
      decrease_factor = 2.0; reuse_diag = 0;
    } else {
      radius = radius / decrease_factor; decrease_factor *= 2.0; reuse_diag = 1;
    }
  }
done:
  memcpy(result_rt, x, sizeof x);
  if (info) { info[0] = iter; info[1] = term; }
}

/* batched driver used by the CPU baseline */
void oracle_uncertainty_pnp_batched(const double* pts2d, const double* pts3d, const double* wgt2d, const double* K,
                                    const double* init_rt, double* result_rt, int* info, int b, int pn) {
  for (int i = 0; i < b; ++i)
    oracle_uncertainty_pnp(pts2d + (size_t)i * pn * 2, pts3d + (size_t)i * pn * 3, wgt2d + (size_t)i * pn * 3,
                           K + 9 * i, init_rt + 6 * i, result_rt + 6 * i, pn, info ? info + 2 * i : 0);
}
