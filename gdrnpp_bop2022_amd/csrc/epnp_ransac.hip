// EPnP + RANSAC on the device — SURVEY.md §8 row a7: the reference's cv2.solvePnPRansac(flags=SOLVEPNP_EPNP) call sites
// (lib/pysixd/misc.py:153-208, core/gdrn_modeling/engine/gdrn_evaluator.py:313-330,373-459) and the EPnP initialiser of
// uncertainty-PnP (core/csrc/uncertainty_pnp/un_pnp_utils.py:27-44), all ROIs of a batch at once, fp64.
//
// The arithmetic restates OpenCV's published algorithms (calib3d epnp.cpp / ptsetreg.cpp / solvepnp.cpp; OpenCV is a
// third-party dependency outside the reference tree): EPnP = PCA control points, barycentric coordinates, the 12x12
// normal matrix MtM of the 2n x 12 projection system, its four smallest eigenvectors, three beta approximations each
// polished by five Gauss-Newton steps on the six control-point distances, Horn/Arun absolute orientation, smallest mean
// reprojection error wins; RANSAC = 5-point minimal sets from cv::RNG(2^64-1) (or injected 32-bit words), float32
// squared reprojection error against reprojErr^2, strictly-more-inliers update, adaptive iteration count, final EPnP
// on the inliers of the best hypothesis; exactly 5 correspondences = one EPnP over all of them, exactly 4 = one P3P solve
// (p3p_4points below), as solvePnPRansac switches its kernel.
//
// Work decomposition (nothing here is bandwidth-relevant: ~100 hypotheses x <= 4096 points per ROI):
//   subsets_kernel     one thread per ROI      the RNG stream is sequential by definition: draws every minimal set
//   hypotheses_kernel  one thread per (ROI, hypothesis)   serial EPnP on 5 points (12x12 Jacobi in scratch)
//   count_kernel       one workgroup per (ROI, hypothesis) inlier count, wave ballot + LDS sum
//   final_kernel       one wave per ROI        replays the sequential best-model / iteration-count logic over the counts,
//                      rebuilds the winner's inlier mask and runs EPnP over the inliers with wave-shuffle reductions
#include "common.hpp"

namespace {

constexpr int kModelPts = 5;

struct Cam { double fu, fv, uc, vc; };

__device__ __forceinline__ bool is_finite(double v) { return fabs(v) <= 1.7976931348623157e308; }  // false for NaN / inf

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// cyclic Jacobi on a symmetric N x N matrix (row-major a), eigenvectors in the COLUMNS of v, eigenvalues ascending
template <int N>
__device__ void jacobi_eigh(double* a, double* v, double* w) {
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) v[i * N + j] = i == j ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 40; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (int i = 0; i < N; ++i) {
      diag += a[i * N + i] * a[i * N + i];
      for (int j = i + 1; j < N; ++j) off += a[i * N + j] * a[i * N + j];
    }
    if (off <= 1e-30 * diag || off == 0.0) break;
    for (int p = 0; p < N - 1; ++p)
      for (int q = p + 1; q < N; ++q) {
        const double apq = a[p * N + q];
        if (apq == 0.0) continue;
        const double theta = (a[q * N + q] - a[p * N + p]) / (2.0 * apq);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < N; ++k) {
          const double akp = a[k * N + p], akq = a[k * N + q];
          a[k * N + p] = c * akp - s * akq;
          a[k * N + q] = s * akp + c * akq;
        }
        for (int k = 0; k < N; ++k) {
          const double apk = a[p * N + k], aqk = a[q * N + k];
          a[p * N + k] = c * apk - s * aqk;
          a[q * N + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < N; ++k) {
          const double vkp = v[k * N + p], vkq = v[k * N + q];
          v[k * N + p] = c * vkp - s * vkq;
          v[k * N + q] = s * vkp + c * vkq;
        }
      }
  }
  for (int i = 0; i < N; ++i) w[i] = a[i * N + i];
  for (int i = 0; i < N - 1; ++i) {  // selection sort, ascending, columns follow
    int m = i;
    for (int j = i + 1; j < N; ++j)
      if (w[j] < w[m]) m = j;
    if (m != i) {
      const double tw = w[i]; w[i] = w[m]; w[m] = tw;
      for (int k = 0; k < N; ++k) { const double tv = v[k * N + i]; v[k * N + i] = v[k * N + m]; v[k * N + m] = tv; }
    }
  }
}

// one-sided Jacobi (Hestenes) SVD of a 3x3 matrix: a = u * diag(s) * vt, singular values descending; rank-deficient
// columns of u are completed to an orthonormal basis
__device__ void svd3(const double* a, double* u, double* s, double* vt) {
  double g[9], v[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  for (int i = 0; i < 9; ++i) g[i] = a[i];
  for (int sweep = 0; sweep < 30; ++sweep) {
    double rot = 0.0;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        double al = 0, be = 0, ga = 0;
        for (int k = 0; k < 3; ++k) { al += g[k * 3 + p] * g[k * 3 + p]; be += g[k * 3 + q] * g[k * 3 + q]; ga += g[k * 3 + p] * g[k * 3 + q]; }
        if (ga == 0.0 || fabs(ga) <= 1e-18 * sqrt(al * be)) continue;
        rot += fabs(ga);
        const double zeta = (be - al) / (2.0 * ga);
        const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
        for (int k = 0; k < 3; ++k) {
          const double gp = g[k * 3 + p], gq = g[k * 3 + q];
          g[k * 3 + p] = c * gp - sn * gq; g[k * 3 + q] = sn * gp + c * gq;
          const double vp = v[k * 3 + p], vq = v[k * 3 + q];
          v[k * 3 + p] = c * vp - sn * vq; v[k * 3 + q] = sn * vp + c * vq;
        }
      }
    if (rot == 0.0) break;
  }
  int ord[3] = {0, 1, 2};
  double nrm[3];
  for (int j = 0; j < 3; ++j) nrm[j] = sqrt(g[j] * g[j] + g[3 + j] * g[3 + j] + g[6 + j] * g[6 + j]);
  for (int i = 0; i < 2; ++i)
    for (int j = i + 1; j < 3; ++j)
      if (nrm[ord[j]] > nrm[ord[i]]) { const int t = ord[i]; ord[i] = ord[j]; ord[j] = t; }
  for (int j = 0; j < 3; ++j) {
    const int c = ord[j];
    s[j] = nrm[c];
    for (int k = 0; k < 3; ++k) { vt[j * 3 + k] = v[k * 3 + c]; u[k * 3 + j] = nrm[c] > 0.0 ? g[k * 3 + c] / nrm[c] : 0.0; }
  }
  if (!(s[1] > 1e-14 * s[0])) {  // rank <= 1: any unit vector orthogonal to u0
    const double x = fabs(u[0]), y = fabs(u[3]), z = fabs(u[6]);
    double e[3] = {x <= y && x <= z ? 1.0 : 0.0, (y < x && y <= z) ? 1.0 : 0.0, (z < x && z < y) ? 1.0 : 0.0};
    const double d = e[0] * u[0] + e[1] * u[3] + e[2] * u[6];
    double w[3] = {e[0] - d * u[0], e[1] - d * u[3], e[2] - d * u[6]};
    const double n = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    u[1] = w[0] / n; u[4] = w[1] / n; u[7] = w[2] / n;
  }
  if (!(s[2] > 1e-14 * s[0])) {  // rank <= 2: u2 = u0 x u1
    u[2] = u[3] * u[7] - u[6] * u[4];
    u[5] = u[6] * u[1] - u[0] * u[7];
    u[8] = u[0] * u[4] - u[3] * u[1];
  }
}

// least squares min |A x - b| for A m x n (row-major, m = 6, n <= 5) by Householder QR (epnp.cpp qr_solve)
template <int M, int NMAX>
__device__ bool qr_solve(double* A, double* b, int n, double* x) {
  for (int k = 0; k < n; ++k) {
    double eta = 0.0;
    for (int i = k; i < M; ++i) eta = fmax(eta, fabs(A[i * NMAX + k]));
    if (eta == 0.0) return false;
    double sum = 0.0;
    for (int i = k; i < M; ++i) { A[i * NMAX + k] /= eta; sum += A[i * NMAX + k] * A[i * NMAX + k]; }
    double sigma = sqrt(sum);
    if (A[k * NMAX + k] < 0.0) sigma = -sigma;
    A[k * NMAX + k] += sigma;
    const double a1 = sigma * A[k * NMAX + k], a2 = -eta * sigma;
    for (int j = k + 1; j < n; ++j) {
      double s2 = 0.0;
      for (int i = k; i < M; ++i) s2 += A[i * NMAX + k] * A[i * NMAX + j];
      const double tau = s2 / a1;
      for (int i = k; i < M; ++i) A[i * NMAX + j] -= tau * A[i * NMAX + k];
    }
    double s2 = 0.0;
    for (int i = k; i < M; ++i) s2 += A[i * NMAX + k] * b[i];
    const double tau = s2 / a1;
    for (int i = k; i < M; ++i) b[i] -= tau * A[i * NMAX + k];
    A[k * NMAX + k] = a2;  // diagonal of R
  }
  for (int i = n - 1; i >= 0; --i) {
    double s2 = b[i];
    for (int j = i + 1; j < n; ++j) s2 -= A[i * NMAX + j] * x[j];
    if (A[i * NMAX + i] == 0.0) return false;
    x[i] = s2 / A[i * NMAX + i];
  }
  return true;
}

__device__ __forceinline__ bool is_inlier(const float* uv, const float* pw, const double* P, const Cam cam, float thr2);

// The point set an EPnP runs on: an explicit index list (RANSAC minimal set), every point of the ROI, or the inliers of a
// hypothesis (the predicate is re-evaluated on the fly — nothing is read back from the mask the kernel writes).
struct PointSet {
  const float* uv;    // [count,2]
  const float* pw;    // [count,3]
  const int* idx;     // nullable: n entries
  const double* P;    // nullable: hypothesis pose (R row-major, t) whose inliers form the set
  Cam cam;
  float thr2;
  int count;          // points of the ROI
  int n;              // size of the set when idx is given
};

template <bool WAVE>
struct Loop {
  int lane;
  __device__ int begin() const { return WAVE ? lane : 0; }
  __device__ int step() const { return WAVE ? 64 : 1; }
  __device__ double sum(double v) const {
    if constexpr (WAVE) return wave_sum(v);
    else return v;
  }
};

template <bool WAVE>
__device__ __forceinline__ bool fetch(const PointSet& ps, int i, double* pw, double* uv) {
  int j = i;
  if (ps.idx) j = ps.idx[i];
  else if (ps.P && !is_inlier(ps.uv + 2 * i, ps.pw + 3 * i, ps.P, ps.cam, ps.thr2)) return false;
  pw[0] = ps.pw[j * 3 + 0]; pw[1] = ps.pw[j * 3 + 1]; pw[2] = ps.pw[j * 3 + 2];
  uv[0] = ps.uv[j * 2 + 0]; uv[1] = ps.uv[j * 2 + 1];
  return true;
}

// EPnP over a point set.  WAVE: the 64 lanes of a wave stride over the set and reduce by shuffles (all lanes end with the
// same pose); otherwise one thread walks the whole set.  Returns false for fewer than 4 points or a degenerate system.
template <bool WAVE>
__device__ bool epnp_solve(const PointSet& ps, const Cam cam, int lane, double* R, double* t) {
  const Loop<WAVE> lp{lane};
  const int n_iter = ps.idx ? ps.n : ps.count;
  double pw[3], uv[2];
  // ---- centroid + PCA of the model points -> control points (epnp.cpp choose_control_points)
  double acc[4] = {0, 0, 0, 0};
  for (int i = lp.begin(); i < n_iter; i += lp.step())
    if (fetch<WAVE>(ps, i, pw, uv)) { acc[0] += pw[0]; acc[1] += pw[1]; acc[2] += pw[2]; acc[3] += 1.0; }
  for (int k = 0; k < 4; ++k) acc[k] = lp.sum(acc[k]);
  const double n = acc[3];
  if (n < 4.0) return false;
  double cws[4][3];
  for (int k = 0; k < 3; ++k) cws[0][k] = acc[k] / n;
  double c6[6] = {0, 0, 0, 0, 0, 0};
  for (int i = lp.begin(); i < n_iter; i += lp.step())
    if (fetch<WAVE>(ps, i, pw, uv)) {
      const double d0 = pw[0] - cws[0][0], d1 = pw[1] - cws[0][1], d2 = pw[2] - cws[0][2];
      c6[0] += d0 * d0; c6[1] += d0 * d1; c6[2] += d0 * d2; c6[3] += d1 * d1; c6[4] += d1 * d2; c6[5] += d2 * d2;
    }
  for (int k = 0; k < 6; ++k) c6[k] = lp.sum(c6[k]);
  {
    double a3[9] = {c6[0], c6[1], c6[2], c6[1], c6[3], c6[4], c6[2], c6[4], c6[5]}, v3[9], w3[3];
    jacobi_eigh<3>(a3, v3, w3);
    for (int i = 1; i < 4; ++i) {  // largest eigenvalue first, like the SVD order of the original
      const int c = 3 - i;
      double k = sqrt(fmax(w3[c], 0.0) / n);
      // the sign of a principal axis is the eigen-solver's choice and would mirror the control point: fixed — the
      // largest-magnitude component of the axis is positive
      int big = 0;
      for (int j = 1; j < 3; ++j)
        if (fabs(v3[j * 3 + c]) > fabs(v3[big * 3 + c])) big = j;
      if (v3[big * 3 + c] < 0.0) k = -k;
      for (int j = 0; j < 3; ++j) cws[i][j] = cws[0][j] + k * v3[j * 3 + c];
    }
  }
  // ---- barycentric coordinates: alpha(1..3) = CC^-1 (pw - c0), CC columns = c_j - c_0
  double cci[9];
  {
    double cc[9];
    for (int r = 0; r < 3; ++r)
      for (int j = 0; j < 3; ++j) cc[r * 3 + j] = cws[j + 1][r] - cws[0][r];
    const double det = cc[0] * (cc[4] * cc[8] - cc[5] * cc[7]) - cc[1] * (cc[3] * cc[8] - cc[5] * cc[6]) + cc[2] * (cc[3] * cc[7] - cc[4] * cc[6]);
    if (!(fabs(det) > 0.0)) return false;
    const double id = 1.0 / det;
    cci[0] = (cc[4] * cc[8] - cc[5] * cc[7]) * id; cci[1] = (cc[2] * cc[7] - cc[1] * cc[8]) * id; cci[2] = (cc[1] * cc[5] - cc[2] * cc[4]) * id;
    cci[3] = (cc[5] * cc[6] - cc[3] * cc[8]) * id; cci[4] = (cc[0] * cc[8] - cc[2] * cc[6]) * id; cci[5] = (cc[2] * cc[3] - cc[0] * cc[5]) * id;
    cci[6] = (cc[3] * cc[7] - cc[4] * cc[6]) * id; cci[7] = (cc[1] * cc[6] - cc[0] * cc[7]) * id; cci[8] = (cc[0] * cc[4] - cc[1] * cc[3]) * id;
  }
  auto alphas = [&](const double* p, double* a) {
    const double d0 = p[0] - cws[0][0], d1 = p[1] - cws[0][1], d2 = p[2] - cws[0][2];
    a[1] = cci[0] * d0 + cci[1] * d1 + cci[2] * d2;
    a[2] = cci[3] * d0 + cci[4] * d1 + cci[5] * d2;
    a[3] = cci[6] * d0 + cci[7] * d1 + cci[8] * d2;
    a[0] = 1.0 - a[1] - a[2] - a[3];
  };
  // ---- MtM (12 x 12, upper triangle accumulated per lane, then reduced)
  double mtm[144];
  for (int k = 0; k < 144; ++k) mtm[k] = 0.0;
  for (int i = lp.begin(); i < n_iter; i += lp.step())
    if (fetch<WAVE>(ps, i, pw, uv)) {
      double a[4], r1[12], r2[12];
      alphas(pw, a);
      for (int j = 0; j < 4; ++j) {
        r1[3 * j] = a[j] * cam.fu; r1[3 * j + 1] = 0.0; r1[3 * j + 2] = a[j] * (cam.uc - uv[0]);
        r2[3 * j] = 0.0; r2[3 * j + 1] = a[j] * cam.fv; r2[3 * j + 2] = a[j] * (cam.vc - uv[1]);
      }
      for (int r = 0; r < 12; ++r)
        for (int c = r; c < 12; ++c) mtm[r * 12 + c] += r1[r] * r1[c] + r2[r] * r2[c];
    }
  for (int r = 0; r < 12; ++r)
    for (int c = r; c < 12; ++c) {
      const double s = lp.sum(mtm[r * 12 + c]);
      mtm[r * 12 + c] = s; mtm[c * 12 + r] = s;
    }
  double evec[144], eval[12];
  jacobi_eigh<12>(mtm, evec, eval);
  {  // canonical basis of an exactly degenerate null space (4 or 5 points): project e_0, e_1, ... onto it and
     // orthonormalise in that order — any basis is a valid set of "smallest eigenvectors", this one is unique
    const double tol = 1e-9 * eval[11];
    int d4 = 0, d = 0;
    for (int i = 0; i < 4; ++i) d4 += eval[i] <= tol ? 1 : 0;
    for (int i = 0; i < 12; ++i) d += eval[i] <= tol ? 1 : 0;
    if (d4 >= 2) {
      double* basis = mtm;  // MtM is dead: d x 12 scratch
      int nb = 0;
      for (int k = 0; k < 12 && nb < d; ++k) {
        double c[12];
        for (int r = 0; r < 12; ++r) {
          double s = 0.0;
          for (int j = 0; j < d; ++j) s += evec[r * 12 + j] * evec[k * 12 + j];
          c[r] = s;
        }
        for (int j = 0; j < nb; ++j) {
          double dot = 0.0;
          for (int r = 0; r < 12; ++r) dot += basis[j * 12 + r] * c[r];
          for (int r = 0; r < 12; ++r) c[r] -= dot * basis[j * 12 + r];
        }
        double nrm = 0.0;
        for (int r = 0; r < 12; ++r) nrm += c[r] * c[r];
        nrm = sqrt(nrm);
        if (nrm > 1e-6) {
          for (int r = 0; r < 12; ++r) basis[nb * 12 + r] = c[r] / nrm;
          ++nb;
        }
      }
      if (nb == d)
        for (int j = 0; j < d; ++j)
          for (int r = 0; r < 12; ++r) evec[r * 12 + j] = basis[j * 12 + r];
    }
  }
  auto V = [&](int i, int k) { return evec[k * 12 + i]; };  // i-th smallest eigenvector (ut[11 - i] of the original), component k
  // ---- the 6 x 10 distance system
  double L[6][10], rho[6];
  {
    const int pa[6] = {0, 0, 0, 1, 1, 2}, pb[6] = {1, 2, 3, 2, 3, 3};
    for (int r = 0; r < 6; ++r) {
      double dv[4][3];
      for (int i = 0; i < 4; ++i)
        for (int k = 0; k < 3; ++k) dv[i][k] = V(i, 3 * pa[r] + k) - V(i, 3 * pb[r] + k);
      auto dot = [&](int i, int j) { return dv[i][0] * dv[j][0] + dv[i][1] * dv[j][1] + dv[i][2] * dv[j][2]; };
      L[r][0] = dot(0, 0); L[r][1] = 2 * dot(0, 1); L[r][2] = dot(1, 1); L[r][3] = 2 * dot(0, 2); L[r][4] = 2 * dot(1, 2);
      L[r][5] = dot(2, 2); L[r][6] = 2 * dot(0, 3); L[r][7] = 2 * dot(1, 3); L[r][8] = 2 * dot(2, 3); L[r][9] = dot(3, 3);
      double d = 0.0;
      for (int k = 0; k < 3; ++k) d += (cws[pa[r]][k] - cws[pb[r]][k]) * (cws[pa[r]][k] - cws[pb[r]][k]);
      rho[r] = d;
    }
  }
  double pw0[3] = {cws[0][0], cws[0][1], cws[0][2]};
  double best_err = 1e300;
  bool have = false;
  for (int cand = 0; cand < 3; ++cand) {
    double betas[4] = {0, 0, 0, 0};
    {  // find_betas_approx_{1,2,3}
      const int ncol = cand == 0 ? 4 : (cand == 1 ? 3 : 5);
      const int cols1[4] = {0, 1, 3, 6}, cols2[3] = {0, 1, 2}, cols3[5] = {0, 1, 2, 3, 4};
      const int* cols = cand == 0 ? cols1 : (cand == 1 ? cols2 : cols3);
      double A[6 * 5], bb[6], x[5] = {0, 0, 0, 0, 0};
      for (int r = 0; r < 6; ++r) {
        for (int c = 0; c < ncol; ++c) A[r * 5 + c] = L[r][cols[c]];
        bb[r] = rho[r];
      }
      if (!qr_solve<6, 5>(A, bb, ncol, x)) continue;
      if (cand == 0) {
        if (x[0] < 0) { betas[0] = sqrt(-x[0]); betas[1] = -x[1] / betas[0]; betas[2] = -x[2] / betas[0]; betas[3] = -x[3] / betas[0]; }
        else { betas[0] = sqrt(x[0]); betas[1] = x[1] / betas[0]; betas[2] = x[2] / betas[0]; betas[3] = x[3] / betas[0]; }
      } else {
        if (x[0] < 0) { betas[0] = sqrt(-x[0]); betas[1] = x[2] < 0 ? sqrt(-x[2]) : 0.0; }
        else { betas[0] = sqrt(x[0]); betas[1] = x[2] > 0 ? sqrt(x[2]) : 0.0; }
        if (x[1] < 0) betas[0] = -betas[0];
        betas[2] = cand == 2 ? x[3] / betas[0] : 0.0;
        betas[3] = 0.0;
      }
      if (!(is_finite(betas[0]) && is_finite(betas[1]) && is_finite(betas[2]) && is_finite(betas[3]))) continue;
    }
    bool ok = true;
    for (int it = 0; it < 5 && ok; ++it) {  // gauss_newton
      double A[6 * 4], bb[6], x[4];
      const double b0 = betas[0], b1 = betas[1], b2 = betas[2], b3 = betas[3];
      for (int r = 0; r < 6; ++r) {
        const double* l = L[r];
        A[r * 4 + 0] = 2 * l[0] * b0 + l[1] * b1 + l[3] * b2 + l[6] * b3;
        A[r * 4 + 1] = l[1] * b0 + 2 * l[2] * b1 + l[4] * b2 + l[7] * b3;
        A[r * 4 + 2] = l[3] * b0 + l[4] * b1 + 2 * l[5] * b2 + l[8] * b3;
        A[r * 4 + 3] = l[6] * b0 + l[7] * b1 + l[8] * b2 + 2 * l[9] * b3;
        bb[r] = rho[r] - (l[0] * b0 * b0 + l[1] * b0 * b1 + l[2] * b1 * b1 + l[3] * b0 * b2 + l[4] * b1 * b2 + l[5] * b2 * b2 +
                          l[6] * b0 * b3 + l[7] * b1 * b3 + l[8] * b2 * b3 + l[9] * b3 * b3);
      }
      ok = qr_solve<6, 4>(A, bb, 4, x);
      if (ok)
        for (int k = 0; k < 4; ++k) betas[k] += x[k];
    }
    if (!ok) continue;
    // ---- compute_R_and_t: camera-frame control points, sign, Horn/Arun, mean reprojection error
    double ccs[4][3];
    for (int j = 0; j < 4; ++j)
      for (int k = 0; k < 3; ++k)
        ccs[j][k] = betas[0] * V(0, 3 * j + k) + betas[1] * V(1, 3 * j + k) + betas[2] * V(2, 3 * j + k) + betas[3] * V(3, 3 * j + k);
    // sign from the first point of the set (solve_for_sign: pcs[0].z < 0)
    double zfirst = 0.0;
    {
      double a[4];
      for (int i = 0; i < n_iter; ++i)
        if (fetch<WAVE>(ps, i, pw, uv)) break;
      alphas(pw, a);
      zfirst = a[0] * ccs[0][2] + a[1] * ccs[1][2] + a[2] * ccs[2][2] + a[3] * ccs[3][2];
    }
    if (zfirst < 0.0)
      for (int j = 0; j < 4; ++j)
        for (int k = 0; k < 3; ++k) ccs[j][k] = -ccs[j][k];
    double s3[3] = {0, 0, 0};
    for (int i = lp.begin(); i < n_iter; i += lp.step())
      if (fetch<WAVE>(ps, i, pw, uv)) {
        double a[4];
        alphas(pw, a);
        for (int k = 0; k < 3; ++k) s3[k] += a[0] * ccs[0][k] + a[1] * ccs[1][k] + a[2] * ccs[2][k] + a[3] * ccs[3][k];
      }
    double pc0[3];
    for (int k = 0; k < 3; ++k) pc0[k] = lp.sum(s3[k]) / n;
    double abt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = lp.begin(); i < n_iter; i += lp.step())
      if (fetch<WAVE>(ps, i, pw, uv)) {
        double a[4], pc[3];
        alphas(pw, a);
        for (int k = 0; k < 3; ++k) pc[k] = a[0] * ccs[0][k] + a[1] * ccs[1][k] + a[2] * ccs[2][k] + a[3] * ccs[3][k] - pc0[k];
        for (int r = 0; r < 3; ++r)
          for (int c = 0; c < 3; ++c) abt[r * 3 + c] += pc[r] * (pw[c] - pw0[c]);
      }
    for (int k = 0; k < 9; ++k) abt[k] = lp.sum(abt[k]);
    double u[9], sv[3], vt[9], Rc[9], tc[3];
    svd3(abt, u, sv, vt);
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) Rc[r * 3 + c] = u[r * 3 + 0] * vt[0 * 3 + c] + u[r * 3 + 1] * vt[1 * 3 + c] + u[r * 3 + 2] * vt[2 * 3 + c];
    const double det = Rc[0] * (Rc[4] * Rc[8] - Rc[5] * Rc[7]) - Rc[1] * (Rc[3] * Rc[8] - Rc[5] * Rc[6]) + Rc[2] * (Rc[3] * Rc[7] - Rc[4] * Rc[6]);
    if (det < 0) { Rc[6] = -Rc[6]; Rc[7] = -Rc[7]; Rc[8] = -Rc[8]; }
    for (int k = 0; k < 3; ++k) tc[k] = pc0[k] - (Rc[k * 3] * pw0[0] + Rc[k * 3 + 1] * pw0[1] + Rc[k * 3 + 2] * pw0[2]);
    double es = 0.0;
    for (int i = lp.begin(); i < n_iter; i += lp.step())
      if (fetch<WAVE>(ps, i, pw, uv)) {
        const double xc = Rc[0] * pw[0] + Rc[1] * pw[1] + Rc[2] * pw[2] + tc[0];
        const double yc = Rc[3] * pw[0] + Rc[4] * pw[1] + Rc[5] * pw[2] + tc[1];
        const double iz = 1.0 / (Rc[6] * pw[0] + Rc[7] * pw[1] + Rc[8] * pw[2] + tc[2]);
        const double du = cam.uc + cam.fu * xc * iz - uv[0], dv2 = cam.vc + cam.fv * yc * iz - uv[1];
        es += sqrt(du * du + dv2 * dv2);
      }
    const double err = lp.sum(es) / n;
    if (is_finite(err) && err < best_err) {
      best_err = err;
      have = true;
      for (int k = 0; k < 9; ++k) R[k] = Rc[k];
      for (int k = 0; k < 3; ++k) t[k] = tc[k];
    }
  }
  return have;
}

// ---- exactly four correspondences: cv2.solvePnPRansac solves once with SOLVEPNP_P3P (solvepnp.cpp: npoints == 4 -> model_points
// = 4, kernel P3P, model_points == npoints -> plain solvePnP).  p3p.cpp (Gao et al. 2003) finds the poses consistent with the first
// three points and orders them by the reprojection error of the fourth; here the same contract through Grunert's elimination
// (oracle/epnp.py p3p_4points has the derivation): a quartic in v = s3 / s1, u = s2 / s1 re-derived from the two cosine laws it has
// to satisfy, camera points s_i j_i, absolute orientation of the three pairs, fourth point picks the pose.  One thread, fp64.
__device__ bool p3p_4points(const float* uv, const float* pw, const Cam cam, double* R, double* t) {
  double xn[4][3], j[3][3], P[4][3];
  for (int i = 0; i < 4; ++i) {
    xn[i][0] = ((double)uv[2 * i] - cam.uc) / cam.fu;
    xn[i][1] = ((double)uv[2 * i + 1] - cam.vc) / cam.fv;
    xn[i][2] = 1.0;
    for (int k = 0; k < 3; ++k) P[i][k] = (double)pw[3 * i + k];
  }
  for (int i = 0; i < 3; ++i) {
    const double n = sqrt(xn[i][0] * xn[i][0] + xn[i][1] * xn[i][1] + 1.0);
    for (int k = 0; k < 3; ++k) j[i][k] = xn[i][k] / n;
  }
  auto d2 = [&](int a, int b) { double s = 0; for (int k = 0; k < 3; ++k) s += (P[a][k] - P[b][k]) * (P[a][k] - P[b][k]); return s; };
  auto dot = [&](int a, int b) { return j[a][0] * j[b][0] + j[a][1] * j[b][1] + j[a][2] * j[b][2]; };
  const double a2 = d2(1, 2), b2 = d2(0, 2), c2 = d2(0, 1);
  if (!(a2 > 0.0 && b2 > 0.0 && c2 > 0.0)) return false;
  const double ca = dot(1, 2), cb = dot(0, 2), cg = dot(0, 1);
  const double q = (a2 - c2) / b2, cb2 = c2 / b2;
  // polynomials, lowest power first: N (deg 2), D (deg 1), Q = 1 - (c^2 / b^2)(1 + v^2 - 2 v cos beta) (deg 2)
  const double N[3] = {1.0 + q, -2.0 * q * cb, q - 1.0}, D[2] = {2.0 * cg, -2.0 * ca}, Q[3] = {1.0 - cb2, 2.0 * cb2 * cb, -cb2};
  double c[5] = {0, 0, 0, 0, 0};                      // N^2 - 2 cos(gamma) N D + D^2 Q
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) c[a + b] += N[a] * N[b];
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 2; ++b) c[a + b] -= 2.0 * cg * N[a] * D[b];
  double DD[3] = {D[0] * D[0], 2.0 * D[0] * D[1], D[1] * D[1]};
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) c[a + b] += DD[a] * Q[b];
  for (int k = 0; k < 5; ++k)
    if (!is_finite(c[k])) return false;
  if (!(fabs(c[4]) >= 1e-300)) return false;
  const double m[4] = {c[0] / c[4], c[1] / c[4], c[2] / c[4], c[3] / c[4]};   // monic: z^4 + m3 z^3 + m2 z^2 + m1 z + m0
  // all four roots by Durand-Kerner (Weierstrass) iteration in complex fp64
  double radius = 0.0;
  for (int k = 0; k < 4; ++k) radius = fmax(radius, fabs(m[k]));
  radius = 1.0 + radius;
  double zr[4], zi[4];
  {
    double pr = 1.0, pi = 0.0;                          // powers of 0.4 + 0.9 i
    for (int k = 0; k < 4; ++k) {
      zr[k] = pr * radius * 0.5; zi[k] = pi * radius * 0.5;
      const double nr = pr * 0.4 - pi * 0.9, ni = pr * 0.9 + pi * 0.4;
      pr = nr; pi = ni;
    }
  }
  for (int it = 0; it < 400; ++it) {
    double moved = 0.0;
    for (int k = 0; k < 4; ++k) {
      double vr = zr[k] + m[3], vi = zi[k];             // Horner: ((z + m3) z + m2) z + m1) z + m0
      for (int e = 2; e >= 0; --e) {
        const double nr = vr * zr[k] - vi * zi[k] + m[e], ni = vr * zi[k] + vi * zr[k];
        vr = nr; vi = ni;
      }
      double dr = 1.0, di = 0.0;
      for (int o = 0; o < 4; ++o)
        if (o != k) {
          const double er = zr[k] - zr[o], ei = zi[k] - zi[o];
          const double nr = dr * er - di * ei, ni = dr * ei + di * er;
          dr = nr; di = ni;
        }
      const double den = dr * dr + di * di;
      if (!(den > 0.0)) continue;
      const double sr = (vr * dr + vi * di) / den, si = (vi * dr - vr * di) / den;
      zr[k] -= sr; zi[k] -= si;
      moved = fmax(moved, (fabs(sr) + fabs(si)) / fmax(1.0, fabs(zr[k]) + fabs(zi[k])));
    }
    if (moved < 1e-15) break;
  }
  bool have = false;
  double best = 0.0;
  for (int k = 0; k < 4; ++k) {
    double v = zr[k];
    if (!(fabs(zi[k]) <= 1e-7 * fmax(1.0, fabs(v))) || !(v > 0.0)) continue;
    const double f = 1.0 + v * v - 2.0 * v * cb;
    const double disc = fmax(cg * cg - 1.0 + cb2 * f, 0.0), sq = sqrt(disc);
    const double u0 = cg + sq, u1 = cg - sq, a2b2 = a2 / b2;
    const double r0 = fabs(u0 * u0 - 2.0 * v * ca * u0 + v * v - a2b2 * f), r1 = fabs(u1 * u1 - 2.0 * v * ca * u1 + v * v - a2b2 * f);
    const double u = r0 <= r1 ? u0 : u1;
    const double w = 1.0 + u * u - 2.0 * u * cg;
    if (!(u > 0.0) || !(w > 0.0)) continue;
    const double s1 = sqrt(c2 / w), sc[3] = {s1, u * s1, v * s1};
    double pc[3][3], mw[3] = {0, 0, 0}, mc[3] = {0, 0, 0};
    for (int i = 0; i < 3; ++i)
      for (int e = 0; e < 3; ++e) { pc[i][e] = sc[i] * j[i][e]; mw[e] += P[i][e] / 3.0; mc[e] += pc[i][e] / 3.0; }
    double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 3; ++i)
      for (int r = 0; r < 3; ++r)
        for (int e = 0; e < 3; ++e) H[r * 3 + e] += (pc[i][r] - mc[r]) * (P[i][e] - mw[e]);
    double U[9], sv[3], Vt[9], Rc[9], tc[3];
    svd3(H, U, sv, Vt);
    for (int r = 0; r < 3; ++r)
      for (int e = 0; e < 3; ++e) Rc[r * 3 + e] = U[r * 3] * Vt[e] + U[r * 3 + 1] * Vt[3 + e] + U[r * 3 + 2] * Vt[6 + e];
    const double det = Rc[0] * (Rc[4] * Rc[8] - Rc[5] * Rc[7]) - Rc[1] * (Rc[3] * Rc[8] - Rc[5] * Rc[6]) + Rc[2] * (Rc[3] * Rc[7] - Rc[4] * Rc[6]);
    if (det < 0)                                        // Arun / Umeyama: flip the direction the three points do not span
      for (int r = 0; r < 3; ++r)
        for (int e = 0; e < 3; ++e) Rc[r * 3 + e] -= 2.0 * U[r * 3 + 2] * Vt[6 + e];
    for (int r = 0; r < 3; ++r) tc[r] = mc[r] - (Rc[r * 3] * mw[0] + Rc[r * 3 + 1] * mw[1] + Rc[r * 3 + 2] * mw[2]);
    double p4[3];
    for (int r = 0; r < 3; ++r) p4[r] = Rc[r * 3] * P[3][0] + Rc[r * 3 + 1] * P[3][1] + Rc[r * 3 + 2] * P[3][2] + tc[r];
    const double ex = p4[0] / p4[2] - xn[3][0], ey = p4[1] / p4[2] - xn[3][1], err = ex * ex + ey * ey;
    if (is_finite(err) && (!have || err < best)) {
      have = true;
      best = err;
      for (int e = 0; e < 9; ++e) R[e] = Rc[e];
      for (int e = 0; e < 3; ++e) t[e] = tc[e];
    }
  }
  return have;
}

__device__ __forceinline__ Cam cam_of(const float* K9) { return Cam{(double)K9[0], (double)K9[4], (double)K9[2], (double)K9[5]}; }

// cv::RNG (multiply-with-carry) or an injected stream of 32-bit words
struct WordStream {
  unsigned long long state;
  const unsigned* words;
  int n_words, pos;
  __device__ bool next(unsigned* out) {
    if (words) {
      if (pos >= n_words) return false;
      *out = words[pos++];
      return true;
    }
    state = (unsigned long long)(unsigned)state * 4164903690ull + (unsigned)(state >> 32);
    *out = (unsigned)state;
    return true;
  }
};

// ptsetreg.cpp getSubset for every iteration of every ROI (sequential per ROI: one thread each)
__global__ void subsets_kernel(const int* __restrict__ count, int stride, const unsigned* __restrict__ words, int n_words,
                               int iters, int* __restrict__ idx, int* __restrict__ n_sub, int b) {
  const int bi = blockIdx.x * blockDim.x + threadIdx.x;
  if (bi >= b) return;
  const int cnt = min(count[bi], stride);
  int made = 0;
  if (cnt > kModelPts) {
    WordStream ws{0xFFFFFFFFFFFFFFFFull, words ? words + (size_t)bi * n_words : nullptr, n_words, 0};
    for (int it = 0; it < iters; ++it) {
      int sel[kModelPts], got = 0, guard = 0;
      while (got < kModelPts && guard < 1000) {
        unsigned w;
        if (!ws.next(&w)) break;
        ++guard;
        const int cand = (int)(w % (unsigned)cnt);
        bool dup = false;
        for (int j = 0; j < got; ++j) dup |= sel[j] == cand;
        if (!dup) sel[got++] = cand;
      }
      if (got < kModelPts) break;
      for (int j = 0; j < kModelPts; ++j) idx[((size_t)bi * iters + it) * kModelPts + j] = sel[j];
      ++made;
    }
  }
  n_sub[bi] = made;
}

__global__ void hypotheses_kernel(const float* __restrict__ img_pts, const float* __restrict__ mdl_pts,
                                  const int* __restrict__ count, int stride, const float* __restrict__ K,
                                  const int* __restrict__ idx, const int* __restrict__ n_sub, int iters,
                                  double* __restrict__ pose, int b) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= b * iters) return;
  const int bi = g / iters, it = g - bi * iters;
  double* out = pose + (size_t)g * 13;
  out[12] = 0.0;
  if (it >= n_sub[bi]) return;
  const Cam cam = cam_of(K + bi * 9);
  PointSet ps{img_pts + (size_t)bi * stride * 2, mdl_pts + (size_t)bi * stride * 3, idx + (size_t)g * kModelPts, nullptr, cam,
              0.f, min(count[bi], stride), kModelPts};
  double R[9], t[3];
  if (epnp_solve<false>(ps, cam, 0, R, t)) {
    for (int k = 0; k < 9; ++k) out[k] = R[k];
    for (int k = 0; k < 3; ++k) out[9 + k] = t[k];
    out[12] = 1.0;
  }
}

// PnPRansacCallback::computeError + findInliers: projections rounded to float32, squared distance in float32
__device__ __forceinline__ bool is_inlier(const float* uv, const float* pw, const double* P, const Cam cam, float thr2) {
  const double x = P[0] * pw[0] + P[1] * pw[1] + P[2] * pw[2] + P[9];
  const double y = P[3] * pw[0] + P[4] * pw[1] + P[5] * pw[2] + P[10];
  const double z = P[6] * pw[0] + P[7] * pw[1] + P[8] * pw[2] + P[11];
  const float u = (float)(cam.fu * x / z + cam.uc), v = (float)(cam.fv * y / z + cam.vc);
  const float dx = uv[0] - u, dy = uv[1] - v;
  return dx * dx + dy * dy <= thr2;
}

__global__ __launch_bounds__(256) void count_kernel(const float* __restrict__ img_pts, const float* __restrict__ mdl_pts,
                                                    const int* __restrict__ count, int stride, const float* __restrict__ K,
                                                    const double* __restrict__ pose, int iters, float thr2,
                                                    int* __restrict__ cnt) {
  const int it = blockIdx.x, bi = blockIdx.y;
  const double* P = pose + ((size_t)bi * iters + it) * 13;
  __shared__ int part[4];
  if (P[12] == 0.0) {
    if (threadIdx.x == 0) cnt[bi * iters + it] = -1;
    return;
  }
  const int n = min(count[bi], stride);
  const Cam cam = cam_of(K + bi * 9);
  const float* uv = img_pts + (size_t)bi * stride * 2;
  const float* pw = mdl_pts + (size_t)bi * stride * 3;
  int c = 0;
  for (int i = threadIdx.x; i < n; i += 256) c += is_inlier(uv + 2 * i, pw + 3 * i, P, cam, thr2) ? 1 : 0;
  for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) cnt[bi * iters + it] = part[0] + part[1] + part[2] + part[3];
}

__device__ int update_num_iters(double p, double ep, int model_points, int max_iters) {  // ptsetreg.cpp RANSACUpdateNumIters
  p = fmin(fmax(p, 0.0), 1.0);
  ep = fmin(fmax(ep, 0.0), 1.0);
  double num = fmax(1.0 - p, 2.2250738585072014e-308);
  double denom = 1.0 - pow(1.0 - ep, (double)model_points);
  if (denom < 2.2250738585072014e-308) return 0;
  num = log(num);
  denom = log(denom);
  return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)rint(num / denom);
}

// one wave per ROI.  status: 1 = pose found, 0 = no model (fewer than 4 points, no hypothesis with >= 5 inliers, or a
// degenerate final / P3P system) — the caller applies the reference's fallbacks.
__global__ __launch_bounds__(64) void final_kernel(const float* __restrict__ img_pts, const float* __restrict__ mdl_pts,
                                                   const int* __restrict__ count, int stride, const float* __restrict__ K,
                                                   const double* __restrict__ pose, const int* __restrict__ cnt,
                                                   const int* __restrict__ n_sub, int iters, float thr2, double confidence,
                                                   unsigned char* __restrict__ mask, float* __restrict__ R_out,
                                                   float* __restrict__ t_out, int* __restrict__ n_inl, int* __restrict__ status) {
  const int bi = blockIdx.x, lane = threadIdx.x;
  const int n = min(count[bi], stride);
  const Cam cam = cam_of(K + bi * 9);
  const float* uv = img_pts + (size_t)bi * stride * 2;
  const float* pw = mdl_pts + (size_t)bi * stride * 3;
  unsigned char* m = mask + (size_t)bi * stride;
  int good = 0;
  bool ok = false;
  const double* Pbest = nullptr;
  // solvePnPRansac: model_points == npoints -> plain solve, every point an inlier; with exactly 4 correspondences the kernel is P3P
  bool p3p = false;
  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t[3] = {0, 0, 0};
  if (n == 4) {
    int okp = 0;
    if (lane == 0) okp = p3p_4points(uv, pw, cam, R, t) ? 1 : 0;
    okp = __shfl(okp, 0, 64);
    for (int k = 0; k < 9; ++k) R[k] = __shfl(R[k], 0, 64);
    for (int k = 0; k < 3; ++k) t[k] = __shfl(t[k], 0, 64);
    p3p = true;
    ok = okp != 0;
    good = ok ? 4 : 0;
    for (int i = lane; i < n; i += 64) m[i] = ok ? 1 : 0;
  } else
  if (n == kModelPts) {
    for (int i = lane; i < n; i += 64) m[i] = 1;
    good = n;
    ok = true;
  } else if (n > kModelPts) {
    int best = -1;
    if (lane == 0) {  // the sequential part of RANSACPointSetRegistrator::run over the precomputed counts
      int niters = iters, max_good = 0;
      const int avail = n_sub[bi];
      for (int it = 0; it < niters && it < avail; ++it) {
        const int c = cnt[bi * iters + it];
        if (c > max(max_good, kModelPts - 1)) {
          max_good = c;
          best = it;
          niters = update_num_iters(confidence, (double)(n - c) / n, kModelPts, niters);
        }
      }
    }
    best = __shfl(best, 0, 64);
    if (best >= 0) {
      const double* P = pose + ((size_t)bi * iters + best) * 13;
      Pbest = P;
      for (int i = lane; i < n; i += 64) {
        const bool in = is_inlier(uv + 2 * i, pw + 3 * i, P, cam, thr2);
        m[i] = in ? 1 : 0;
        good += in ? 1 : 0;
      }
      for (int off = 32; off >= 1; off >>= 1) good += __shfl_xor(good, off, 64);
      ok = true;
    }
  }
  for (int i = n + lane; i < stride; i += 64) m[i] = 0;
  if (ok && !p3p) {
    PointSet ps{uv, pw, nullptr, Pbest, cam, thr2, n, n};
    ok = epnp_solve<true>(ps, cam, lane, R, t);
  }
  if (!ok) {
    for (int i = lane; i < n; i += 64) m[i] = 0;
    good = 0;
  }
  if (lane == 0) {
    for (int k = 0; k < 9; ++k) R_out[bi * 9 + k] = (float)(ok ? R[k] : (k % 4 == 0 ? 1.0 : 0.0));
    for (int k = 0; k < 3; ++k) t_out[bi * 3 + k] = (float)(ok ? t[k] : 0.0);
    n_inl[bi] = good;
    status[bi] = ok ? 1 : 0;
  }
}

// plain EPnP on every point of each problem (n the same for all): one wave per problem
__global__ __launch_bounds__(64) void epnp_batched_kernel(const float* __restrict__ img_pts, const float* __restrict__ mdl_pts, int n,
                                                          const float* __restrict__ K, float* __restrict__ R_out,
                                                          float* __restrict__ t_out, int* __restrict__ status) {
  const int bi = blockIdx.x, lane = threadIdx.x;
  const Cam cam = cam_of(K + bi * 9);
  PointSet ps{img_pts + (size_t)bi * n * 2, mdl_pts + (size_t)bi * n * 3, nullptr, nullptr, cam, 0.f, n, n};
  double R[9], t[3];
  const bool ok = epnp_solve<true>(ps, cam, lane, R, t);
  if (lane == 0) {
    for (int k = 0; k < 9; ++k) R_out[bi * 9 + k] = (float)(ok ? R[k] : (k % 4 == 0 ? 1.0 : 0.0));
    for (int k = 0; k < 3; ++k) t_out[bi * 3 + k] = (float)(ok ? t[k] : 0.0);
    status[bi] = ok ? 1 : 0;
  }
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

extern "C" size_t gdrnpp_epnp_ransac_workspace_bytes(int b, int stride, int iters) {
  if (b <= 0 || stride <= 0 || iters <= 0) return 0;
  return align256((size_t)b * iters * kModelPts * sizeof(int)) + align256((size_t)b * sizeof(int)) +
         align256((size_t)b * iters * 13 * sizeof(double)) + align256((size_t)b * iters * sizeof(int));
}

extern "C" int gdrnpp_epnp_ransac(const float* img_pts, const float* mdl_pts, const int* count, int stride, const float* K,
                                  const unsigned* draws, int n_draws, int iters, float reproj_err, double confidence,
                                  float* R_out, float* t_out, int* n_inliers, int* status, unsigned char* inlier_mask,
                                  int b, void* workspace, size_t workspace_bytes, void* stream) {
  if (b == 0) return 0;
  GDRNPP_REQUIRE(img_pts && mdl_pts && count && K && R_out && t_out && n_inliers && status && inlier_mask && workspace,
                 GDRNPP_EINVAL, "gdrnpp_epnp_ransac: null pointer");
  GDRNPP_REQUIRE(b > 0 && stride > 0 && iters > 0 && iters <= 65535 && reproj_err > 0.f, GDRNPP_EINVAL,
                 "gdrnpp_epnp_ransac: b=%d stride=%d iters=%d reproj_err=%g", b, stride, iters, (double)reproj_err);
  GDRNPP_REQUIRE(b <= 65535, GDRNPP_ELIMIT, "gdrnpp_epnp_ransac: b=%d > 65535", b);
  GDRNPP_REQUIRE(!draws || n_draws >= iters * kModelPts, GDRNPP_EINVAL,
                 "gdrnpp_epnp_ransac: %d injected words cannot fill %d minimal sets", n_draws, iters);
  GDRNPP_REQUIRE(workspace_bytes >= gdrnpp_epnp_ransac_workspace_bytes(b, stride, iters), GDRNPP_EINVAL,
                 "gdrnpp_epnp_ransac: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  char* w = (char*)workspace;
  int* idx = (int*)w; w += align256((size_t)b * iters * kModelPts * sizeof(int));
  int* n_sub = (int*)w; w += align256((size_t)b * sizeof(int));
  double* pose = (double*)w; w += align256((size_t)b * iters * 13 * sizeof(double));
  int* cnt = (int*)w;
  const float thr2 = reproj_err * reproj_err;
  hipLaunchKernelGGL(subsets_kernel, dim3((b + 63) / 64), dim3(64), 0, st, count, stride, draws, n_draws, iters, idx, n_sub, b);
  hipLaunchKernelGGL(hypotheses_kernel, dim3((b * iters + 63) / 64), dim3(64), 0, st, img_pts, mdl_pts, count, stride, K, idx,
                     n_sub, iters, pose, b);
  hipLaunchKernelGGL(count_kernel, dim3(iters, b), dim3(256), 0, st, img_pts, mdl_pts, count, stride, K, pose, iters, thr2, cnt);
  hipLaunchKernelGGL(final_kernel, dim3(b), dim3(64), 0, st, img_pts, mdl_pts, count, stride, K, pose, cnt, n_sub, iters, thr2,
                     confidence, inlier_mask, R_out, t_out, n_inliers, status);
  return gdrnpp::check_launch("gdrnpp_epnp_ransac");
}

extern "C" int gdrnpp_epnp_batched(const float* img_pts, const float* mdl_pts, int n, const float* K, float* R_out,
                                   float* t_out, int* status, int b, void* stream) {
  if (b == 0) return 0;
  GDRNPP_REQUIRE(img_pts && mdl_pts && K && R_out && t_out && status, GDRNPP_EINVAL, "gdrnpp_epnp_batched: null pointer");
  GDRNPP_REQUIRE(b > 0 && n >= 4, GDRNPP_EINVAL, "gdrnpp_epnp_batched: b=%d n=%d (EPnP needs at least 4 points)", b, n);
  hipLaunchKernelGGL(epnp_batched_kernel, dim3(b), dim3(64), 0, (hipStream_t)stream, img_pts, mdl_pts, n, K, R_out, t_out, status);
  return gdrnpp::check_launch("gdrnpp_epnp_batched");
}
