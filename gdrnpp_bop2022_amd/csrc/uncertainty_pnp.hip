// Uncertainty-PnP (covariance-weighted reprojection LM) on gfx950 — SURVEY.md §8 row a11.
//
// Behavioural spec: core/csrc/uncertainty_pnp/src/uncertainty_pnp.cpp
//   residual  W * (pi(R(w) X + t) - x),  W = [[wxx,wxy],[wxy,wyy]]   :16-34
//   ceres::Solve, default options, DENSE_SCHUR, 6-dof angle-axis pose  :61-92
// Ceres itself is not in the tree as a library; its published LM trust-region
// schedule is restated (see oracle/upnp_oracle.c header for the constants) and
// this kernel follows the same schedule step by step in fp64.
//
// Design: one wave64 per problem (B problems -> B single-wave workgroups, no
// barriers, wave-uniform control flow).  R(w) and dR/dw_k are evaluated once per
// LM iteration with 3-infinitesimal dual numbers (the same rotation.h formula the
// reference differentiates, incl. its small-angle branch); lanes stride over the
// correspondences, each accumulating the 21 unique entries of J^T J, the 6 of
// J^T r and the cost in registers; a 6-step shuffle tree reduces the 28 doubles;
// the damped 6x6 system is solved by Cholesky redundantly in every lane.
// Per iteration the correspondences (64 B/point in the fp64 ABI) are re-read from
// L2; algorithmic HBM bytes = 64*pn + 72 + 48 in, 48 out per problem.
#include "common.hpp"
#include <cfloat>

namespace {

struct J3 {  // dual number with 3 infinitesimals (d/dw0, d/dw1, d/dw2)
  double v, d[3];
};
__device__ __forceinline__ J3 jc(double c) { return J3{c, {0.0, 0.0, 0.0}}; }
__device__ __forceinline__ J3 operator+(J3 a, J3 b) { return J3{a.v + b.v, {a.d[0] + b.d[0], a.d[1] + b.d[1], a.d[2] + b.d[2]}}; }
__device__ __forceinline__ J3 operator-(J3 a, J3 b) { return J3{a.v - b.v, {a.d[0] - b.d[0], a.d[1] - b.d[1], a.d[2] - b.d[2]}}; }
__device__ __forceinline__ J3 operator*(J3 a, J3 b) {
  return J3{a.v * b.v, {a.v * b.d[0] + a.d[0] * b.v, a.v * b.d[1] + a.d[1] * b.v, a.v * b.d[2] + a.d[2] * b.v}};
}
__device__ __forceinline__ J3 jinv(J3 g) {  // 1/g
  double inv = 1.0 / g.v;
  return J3{inv, {-inv * g.d[0] * inv, -inv * g.d[1] * inv, -inv * g.d[2] * inv}};
}

// R (row-major) and dR[k] = dR/dw_k, rotation.h AngleAxisRotatePoint in matrix form
__device__ void rotation_and_derivs(const double* w, double* R, double dR[3][9]) {
  J3 a[3] = {J3{w[0], {1, 0, 0}}, J3{w[1], {0, 1, 0}}, J3{w[2], {0, 0, 1}}};
  J3 theta2 = a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
  J3 M[9];
  if (theta2.v > 2.220446049250313e-16) {
    double th = sqrt(theta2.v);
    double t = 1.0 / (2.0 * th);
    J3 theta{th, {t * theta2.d[0], t * theta2.d[1], t * theta2.d[2]}};
    double c = cos(th), s = sin(th);
    J3 ct{c, {-s * theta.d[0], -s * theta.d[1], -s * theta.d[2]}};
    J3 st{s, {c * theta.d[0], c * theta.d[1], c * theta.d[2]}};
    J3 ti = jinv(theta);
    J3 u[3] = {a[0] * ti, a[1] * ti, a[2] * ti};
    J3 omc = jc(1.0) - ct;
    // R = c I + s [u]x + (1-c) u u^T
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) M[i * 3 + j] = u[i] * u[j] * omc;
    M[0] = M[0] + ct; M[4] = M[4] + ct; M[8] = M[8] + ct;
    M[1] = M[1] - u[2] * st; M[2] = M[2] + u[1] * st;
    M[3] = M[3] + u[2] * st; M[5] = M[5] - u[0] * st;
    M[6] = M[6] - u[1] * st; M[7] = M[7] + u[0] * st;
  } else {
    // first-order branch: R = I + [w]x
    for (int i = 0; i < 9; ++i) M[i] = jc(0.0);
    M[0] = M[4] = M[8] = jc(1.0);
    M[1] = jc(0.0) - a[2]; M[2] = a[1];
    M[3] = a[2];           M[5] = jc(0.0) - a[0];
    M[6] = jc(0.0) - a[1]; M[7] = a[0];
  }
  for (int i = 0; i < 9; ++i) {
    R[i] = M[i].v;
    dR[0][i] = M[i].d[0]; dR[1][i] = M[i].d[1]; dR[2][i] = M[i].d[2];
  }
}

__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off, 64);
  return x;
}

// One pass over the correspondences: cost, g = J^T r, H = J^T J (full 6x6, symmetric)
template <typename T>
__device__ double evaluate(const double* x, const T* __restrict__ p2, const T* __restrict__ p3,
                           const T* __restrict__ wg, double fx, double fy, double px, double py, int pn,
                           double* H, double* g) {
  double R[9], dR[3][9];
  rotation_and_derivs(x, R, dR);
  double acc[28];
#pragma unroll
  for (int i = 0; i < 28; ++i) acc[i] = 0.0;
  const int lane = threadIdx.x & 63;
  for (int i = lane; i < pn; i += 64) {
    const double X = (double)p3[3 * i], Y = (double)p3[3 * i + 1], Z = (double)p3[3 * i + 2];
    const double tx = R[0] * X + R[1] * Y + R[2] * Z + x[3];
    const double ty = R[3] * X + R[4] * Y + R[5] * Z + x[4];
    const double tz = R[6] * X + R[7] * Y + R[8] * Z + x[5];
    double dX[6], dY[6], dZ[6];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      dX[k] = dR[k][0] * X + dR[k][1] * Y + dR[k][2] * Z;
      dY[k] = dR[k][3] * X + dR[k][4] * Y + dR[k][5] * Z;
      dZ[k] = dR[k][6] * X + dR[k][7] * Y + dR[k][8] * Z;
    }
    dX[3] = 1; dX[4] = 0; dX[5] = 0;
    dY[3] = 0; dY[4] = 1; dY[5] = 0;
    dZ[3] = 0; dZ[4] = 0; dZ[5] = 1;
    const double inv = 1.0 / tz;
    const double qx = fx * tx * inv, qy = fy * ty * inv;  // jet division: f*h, (f' - f*h*g')*h
    const double ex = (qx + px) - (double)p2[2 * i], ey = (qy + py) - (double)p2[2 * i + 1];
    // wg == nullptr: identity weights = plain reprojection error (cv2.solvePnP ITERATIVE's cost)
    const double wxx = wg ? (double)wg[3 * i] : 1.0, wxy = wg ? (double)wg[3 * i + 1] : 0.0, wyy = wg ? (double)wg[3 * i + 2] : 1.0;
    const double r0 = wxx * ex + wxy * ey, r1 = wxy * ex + wyy * ey;
    double J0[6], J1[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const double du = (fx * dX[k] - qx * dZ[k]) * inv;
      const double dv = (fy * dY[k] - qy * dZ[k]) * inv;
      J0[k] = wxx * du + wxy * dv;
      J1[k] = wxy * du + wyy * dv;
    }
    int q = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int b = a; b < 6; ++b) acc[q++] += J0[a] * J0[b] + J1[a] * J1[b];
#pragma unroll
    for (int a = 0; a < 6; ++a) acc[21 + a] += J0[a] * r0 + J1[a] * r1;
    acc[27] += r0 * r0 + r1 * r1;
  }
#pragma unroll
  for (int i = 0; i < 28; ++i) acc[i] = wave_sum(acc[i]);
  int q = 0;
  for (int a = 0; a < 6; ++a)
    for (int b = a; b < 6; ++b) { H[a * 6 + b] = acc[q]; H[b * 6 + a] = acc[q]; ++q; }
  for (int a = 0; a < 6; ++a) g[a] = acc[21 + a];
  double cost = 0.5 * acc[27];
  if (!isfinite(cost)) cost = DBL_MAX;  // evaluation failure -> step rejected
  return cost;
}

__device__ bool chol_solve6(const double* A, const double* b, double* y) {
  double L[36];
  for (int i = 0; i < 36; ++i) L[i] = 0.0;
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = A[i * 6 + j];
      for (int k = 0; k < j; ++k) s -= L[i * 6 + k] * L[j * 6 + k];
      if (i == j) {
        if (!(s > 0.0)) return false;
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
    if (!isfinite(y[i])) return false;
  return true;
}

// Levenberg-Marquardt with Ceres' trust-region schedule (see oracle/upnp_oracle.c); x0 -> x, wave-uniform
template <typename T>
__device__ void lm_minimise(const double* x0, double* x, const T* __restrict__ p2, const T* __restrict__ p3,
                            const T* __restrict__ wg, double fx, double fy, double px, double py, int pn, int& iter_out,
                            int& term_out) {
  const int max_it = 50;
  const double min_rel_dec = 1e-3, ftol = 1e-6, gtol = 1e-10, ptol = 1e-8;
  const double min_diag = 1e-6, max_diag = 1e32, max_radius = 1e16, min_radius = 1e-32;

  double H[36], g[6], scale[6], diag[6];
  for (int k = 0; k < 6; ++k) { x[k] = x0[k]; diag[k] = 0.0; }
  double radius = 1e4, decrease_factor = 2.0;
  int invalid = 0, iter = 0, term = 3;
  bool reuse_diag = false;

  double cost = evaluate(x, p2, p3, wg, fx, fy, px, py, pn, H, g);
  double gmax = 0.0;
  for (int i = 0; i < 6; ++i) {
    scale[i] = 1.0 / (1.0 + sqrt(H[i * 6 + i]));
    gmax = fmax(gmax, fabs(g[i]));
  }
  bool done = false;
  if (gmax <= gtol) { term = 2; done = true; }

  while (!done) {
    if (iter >= max_it) { term = 3; break; }
    if (gmax <= gtol) { term = 2; break; }
    if (radius < min_radius) { term = 4; break; }
    ++iter;
    double Hs[36], gs[6], A[36], y[6], step[6];
    for (int a = 0; a < 6; ++a) {
      gs[a] = scale[a] * g[a];
      for (int b = 0; b < 6; ++b) Hs[a * 6 + b] = scale[a] * H[a * 6 + b] * scale[b];
    }
    if (!reuse_diag)
      for (int a = 0; a < 6; ++a) diag[a] = fmin(fmax(Hs[a * 6 + a], min_diag), max_diag);
    for (int i = 0; i < 36; ++i) A[i] = Hs[i];
    for (int a = 0; a < 6; ++a) A[a * 6 + a] += diag[a] / radius;
    reuse_diag = true;
    const bool ok = chol_solve6(A, gs, y);
    double model_change = 0.0;
    if (ok) {
      double sg = 0.0, sHs = 0.0;
      for (int a = 0; a < 6; ++a) step[a] = -y[a];
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
      radius *= 0.5;
      reuse_diag = false;
      continue;
    }
    invalid = 0;
    double cand[6], delta2 = 0.0, xn2 = 0.0;
    for (int a = 0; a < 6; ++a) {
      cand[a] = x[a] + step[a] * scale[a];
      delta2 += (x[a] - cand[a]) * (x[a] - cand[a]);
      xn2 += x[a] * x[a];
    }
    double Hc[36], gc[6];
    const double cand_cost = evaluate(cand, p2, p3, wg, fx, fy, px, py, pn, Hc, gc);
    if (sqrt(delta2) <= ptol * (sqrt(xn2) + ptol)) { term = 0; break; }
    if (fabs(cost - cand_cost) <= ftol * cost) { term = 1; break; }
    const double rel_dec = (cost - cand_cost) / model_change;
    if (rel_dec > min_rel_dec) {
      gmax = 0.0;
      for (int i = 0; i < 6; ++i) { x[i] = cand[i]; g[i] = gc[i]; gmax = fmax(gmax, fabs(g[i])); }
      for (int i = 0; i < 36; ++i) H[i] = Hc[i];
      cost = cand_cost;
      const double t = 2.0 * rel_dec - 1.0;
      radius = fmin(max_radius, radius / fmax(1.0 / 3.0, 1.0 - t * t * t));
      decrease_factor = 2.0;
      reuse_diag = false;
    } else {
      radius = radius / decrease_factor;
      decrease_factor *= 2.0;
      reuse_diag = true;
    }
  }
  iter_out = iter;
  term_out = term;
}

__global__ __launch_bounds__(64) void upnp_kernel(const double* __restrict__ pts2d, const double* __restrict__ pts3d,
                                                  const double* __restrict__ wgt2d, const double* __restrict__ Kb,
                                                  const double* __restrict__ init_rt, double* __restrict__ result_rt,
                                                  int* __restrict__ info, int pn) {
  const int bi = blockIdx.x;
  const double* p2 = pts2d + (size_t)bi * pn * 2;
  const double* p3 = pts3d + (size_t)bi * pn * 3;
  const double* wg = wgt2d + (size_t)bi * pn * 3;
  const double* K = Kb + 9 * (size_t)bi;
  double x0[6], x[6];
  for (int k = 0; k < 6; ++k) x0[k] = init_rt[6 * (size_t)bi + k];
  int iter, term;
  lm_minimise<double>(x0, x, p2, p3, wg, K[0], K[4], K[2], K[5], pn, iter, term);
  if ((threadIdx.x & 63) == 0) {
    for (int k = 0; k < 6; ++k) result_rt[6 * (size_t)bi + k] = x[k];
    if (info) { info[2 * bi] = iter; info[2 * bi + 1] = term; }
  }
}

// ---- net-initialised iterative PnP on the decoded correspondences (gdrn_evaluator.py:241-371, pnp_type "iter") ----
// cv2.Rodrigues restated (matrix -> rotation vector) in fp64
__device__ void rodrigues_log(const double* R, double* r) {
  double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
  const double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
  double c = (R[0] + R[4] + R[8] - 1) * 0.5;
  c = c > 1. ? 1. : (c < -1. ? -1. : c);
  const double theta = acos(c);
  if (s < 1e-5) {
    if (c > 0) { r[0] = r[1] = r[2] = 0.0; return; }
    double t = (R[0] + 1) * 0.5;
    rx = sqrt(fmax(t, 0.));
    t = (R[4] + 1) * 0.5;
    ry = sqrt(fmax(t, 0.)) * (R[1] < 0 ? -1. : 1.);
    t = (R[8] + 1) * 0.5;
    rz = sqrt(fmax(t, 0.)) * (R[2] < 0 ? -1. : 1.);
    if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && (R[5] > 0) != (ry * rz > 0)) rz = -rz;
    const double sc = theta / sqrt(rx * rx + ry * ry + rz * rz);
    r[0] = rx * sc; r[1] = ry * sc; r[2] = rz * sc;
    return;
  }
  const double vth = 1 / (2 * s) * theta;
  r[0] = rx * vth; r[1] = ry * vth; r[2] = rz * vth;
}

__global__ __launch_bounds__(64) void pnp_iter_kernel(const float* __restrict__ img_pts, const float* __restrict__ mdl_pts,
                                                      const int* __restrict__ count, int stride,
                                                      const float* __restrict__ Kb, const float* __restrict__ R_net,
                                                      const float* __restrict__ t_net, float* __restrict__ R_out,
                                                      float* __restrict__ t_out, int* __restrict__ info) {
  const int bi = blockIdx.x;
  const int pn = count[bi];
  const float* p2 = img_pts + (size_t)bi * stride * 2;
  const float* p3 = mdl_pts + (size_t)bi * stride * 3;
  const float* K = Kb + 9 * (size_t)bi;
  double Rn[9], x0[6], x[6];
  for (int k = 0; k < 9; ++k) Rn[k] = (double)R_net[9 * (size_t)bi + k];
  for (int k = 0; k < 3; ++k) x0[3 + k] = (double)t_net[3 * (size_t)bi + k];
  rodrigues_log(Rn, x0);
  int iter = 0, term = -1;
  bool use_pnp = pn >= 4;  // fewer than 4 correspondences: keep the network pose (gdrn_evaluator.py:355-358)
  if (use_pnp) lm_minimise<float>(x0, x, p2, p3, (const float*)nullptr, (double)K[0], (double)K[4], (double)K[2], (double)K[5], pn, iter, term);
  if ((threadIdx.x & 63) == 0) {
    float* Ro = R_out + 9 * (size_t)bi;
    float* to = t_out + 3 * (size_t)bi;
    if (!use_pnp) {
      for (int k = 0; k < 9; ++k) Ro[k] = R_net[9 * (size_t)bi + k];
      for (int k = 0; k < 3; ++k) to[k] = t_net[3 * (size_t)bi + k];
    } else {
      double R[9], dR[3][9];
      rotation_and_derivs(x, R, dR);
      for (int k = 0; k < 9; ++k) Ro[k] = (float)R[k];
      // te(t_est, t_net) > 1 m -> fall back to the network translation (gdrn_evaluator.py:347-351)
      const double dx = x[3] - x0[3], dy = x[4] - x0[4], dz = x[5] - x0[5];
      const bool far = sqrt(dx * dx + dy * dy + dz * dz) > 1.0;
      for (int k = 0; k < 3; ++k) to[k] = far ? t_net[3 * (size_t)bi + k] : (float)x[3 + k];
      if (far) term = 100 + term;
    }
    if (info) { info[2 * bi] = iter; info[2 * bi + 1] = term; }
  }
}

}  // namespace

extern "C" {

int gdrnpp_uncertainty_pnp_batched(const double* pts2d, const double* pts3d, const double* wgt2d, const double* K,
                                   const double* init_rt, double* result_rt, int* info, int b, int pn,
                                   void* stream) {
  GDRNPP_REQUIRE(pts2d && pts3d && wgt2d && K && init_rt && result_rt, GDRNPP_EINVAL,
                 "gdrnpp_uncertainty_pnp_batched: null pointer");
  GDRNPP_REQUIRE(b > 0 && pn > 0, GDRNPP_EINVAL, "gdrnpp_uncertainty_pnp_batched: b=%d pn=%d", b, pn);
  hipLaunchKernelGGL(upnp_kernel, dim3(b), dim3(64), 0, (hipStream_t)stream, pts2d, pts3d, wgt2d, K, init_rt,
                     result_rt, info, pn);
  return gdrnpp::check_launch("gdrnpp_uncertainty_pnp_batched");
}

int gdrnpp_pnp_iter_from_correspondences(const float* img_pts, const float* mdl_pts, const int* count, int stride,
                                         const float* K, const float* R_net, const float* t_net, float* R_out,
                                         float* t_out, int* info, int b, void* stream) {
  GDRNPP_REQUIRE(img_pts && mdl_pts && count && K && R_net && t_net && R_out && t_out, GDRNPP_EINVAL,
                 "gdrnpp_pnp_iter_from_correspondences: null pointer");
  GDRNPP_REQUIRE(b > 0 && stride > 0, GDRNPP_EINVAL, "gdrnpp_pnp_iter_from_correspondences: b=%d stride=%d", b, stride);
  hipLaunchKernelGGL(pnp_iter_kernel, dim3(b), dim3(64), 0, (hipStream_t)stream, img_pts, mdl_pts, count, stride, K,
                     R_net, t_net, R_out, t_out, info);
  return gdrnpp::check_launch("gdrnpp_pnp_iter_from_correspondences");
}

void uncertainty_pnp(double* pts2d, double* pts3d, double* wgt2d, double* K, double* init_rt, double* result_rt,
                     int pn) {
  // host-pointer drop-in for core/csrc/uncertainty_pnp/src/ext.h:1-9; the reference has no
  // error channel, so failures are reported on stderr and result_rt is filled with NaN.
  auto fail = [&](const char* what) {
    fprintf(stderr, "[gdrnpp_hip] uncertainty_pnp: %s: %s\n", what, gdrnpp_last_error());
    for (int k = 0; k < 6; ++k) result_rt[k] = __builtin_nan("");
  };
  if (!pts2d || !pts3d || !wgt2d || !K || !init_rt || !result_rt || pn <= 0) {
    gdrnpp::set_error("bad arguments pn=%d", pn);
    if (result_rt) fail("arguments");
    return;
  }
  const size_t n2 = sizeof(double) * 2 * (size_t)pn, n3 = sizeof(double) * 3 * (size_t)pn;
  const size_t total = n2 + 2 * n3 + sizeof(double) * (9 + 6 + 6);
  char* d = (char*)gdrnpp::shim_scratch(total);   // kept per host thread: no hipMalloc / hipFree per call
  if (!d) return fail("scratch");
  double* d2 = (double*)d;
  double* d3 = (double*)(d + n2);
  double* dw = (double*)(d + n2 + n3);
  double* dK = (double*)(d + n2 + 2 * n3);
  double* di = dK + 9;
  double* dr = di + 6;
  hipError_t e = hipSuccess;
  auto up = [&](void* dst, const void* src, size_t n) { if (e == hipSuccess) e = hipMemcpy(dst, src, n, hipMemcpyHostToDevice); };
  up(d2, pts2d, n2); up(d3, pts3d, n3); up(dw, wgt2d, n3); up(dK, K, sizeof(double) * 9); up(di, init_rt, sizeof(double) * 6);
  if (e != hipSuccess) { gdrnpp::set_error("%s", hipGetErrorString(e)); return fail("hipMemcpy H2D"); }
  int rc = gdrnpp_uncertainty_pnp_batched(d2, d3, dw, dK, di, dr, nullptr, 1, pn, nullptr);
  if (rc == 0) {
    e = hipMemcpy(result_rt, dr, sizeof(double) * 6, hipMemcpyDeviceToHost);
    if (e != hipSuccess) { gdrnpp::set_error("%s", hipGetErrorString(e)); fail("hipMemcpy D2H"); }
  } else {
    fail("launch");
  }
}

}  // extern "C"
