// ORACLE build shim: evaluates the uncertainty-PnP cost functor with the reference's VENDORED
// header-only Ceres pieces (core/csrc/uncertainty_pnp/include/ceres/{jet.h,rotation.h,
// tiny_solver.h}) — libceres itself is not in the tree.  The functor body restates
// uncertainty_pnp.cpp:16-34 token for token (it cannot be #included: that file pulls in
// <ceres/ceres.h>, which needs glog and the compiled library).
#include <cmath>
#include <cstring>
#include "ceres/jet.h"
#include "ceres/rotation.h"
#include "ceres/tiny_solver.h"

struct ReprojectionErrorArray {
  double x2d, y2d, x3d, y3d, z3d, fx, fy, px, py, wxx, wxy, wyy;
  template <typename T>
  bool operator()(const T* const pose, T* residuals) const {
    T pts3d[] = {T(x3d), T(y3d), T(z3d)};
    T trans_pts3d[3];
    ceres::AngleAxisRotatePoint(pose, pts3d, trans_pts3d);
    trans_pts3d[0] += pose[3];
    trans_pts3d[1] += pose[4];
    trans_pts3d[2] += pose[5];
    T proj_x = T(fx) * trans_pts3d[0] / trans_pts3d[2] + T(px);
    T proj_y = T(fy) * trans_pts3d[1] / trans_pts3d[2] + T(py);
    T diff_x = proj_x - T(x2d);
    T diff_y = proj_y - T(y2d);
    residuals[0] = T(wxx) * diff_x + T(wxy) * diff_y;
    residuals[1] = T(wxy) * diff_x + T(wyy) * diff_y;
    return true;
  }
};

static ReprojectionErrorArray make(const double* p2, const double* p3, const double* w, const double* K) {
  return ReprojectionErrorArray{p2[0], p2[1], p3[0], p3[1], p3[2], K[0], K[4], K[2], K[5], w[0], w[1], w[2]};
}

// TinySolver problem: all residuals stacked, 6 parameters, Jacobian by ceres::Jet<double,6>
struct StackedProblem {
  typedef double Scalar;
  enum { NUM_RESIDUALS = Eigen::Dynamic, NUM_PARAMETERS = 6 };
  const double *p2, *p3, *w, *K;
  int pn;
  int NumResiduals() const { return 2 * pn; }
  bool operator()(const double* x, double* r, double* J) const {
    typedef ceres::Jet<double, 6> JetT;
    for (int i = 0; i < pn; ++i) {
      ReprojectionErrorArray f = make(p2 + 2 * i, p3 + 3 * i, w + 3 * i, K);
      if (J) {
        JetT xj[6], rj[2];
        for (int k = 0; k < 6; ++k) xj[k] = JetT(x[k], k);
        f(xj, rj);
        for (int q = 0; q < 2; ++q) {
          r[2 * i + q] = rj[q].a;
          // Eigen matrix handed to TinySolver is column-major [2pn x 6]
          for (int k = 0; k < 6; ++k) J[(2 * i + q) + k * (2 * pn)] = rj[q].v[k];
        }
      } else {
        f(x, r + 2 * i);
      }
    }
    return true;
  }
};

extern "C" {
// residual + jacobian (row-major [2][6]) of ONE observation through ceres/jet.h
void ref_upnp_residual(const double* pose, const double* p2, const double* p3, const double* w, const double* K,
                       double* r, double* J) {
  typedef ceres::Jet<double, 6> JetT;
  ReprojectionErrorArray f = make(p2, p3, w, K);
  JetT xj[6], rj[2];
  for (int k = 0; k < 6; ++k) xj[k] = JetT(pose[k], k);
  f(xj, rj);
  for (int q = 0; q < 2; ++q) {
    r[q] = rj[q].a;
    for (int k = 0; k < 6; ++k) J[q * 6 + k] = rj[q].v[k];
  }
}

// LM minimisation with the vendored ceres::TinySolver (NOT ceres::Solve — different schedule,
// same minimum): used to check that the restated minimiser lands on the same optimum.
void ref_upnp_tinysolver(const double* pts2d, const double* pts3d, const double* wgt, const double* K,
                         const double* init, double* result, int pn, int max_iter) {
  StackedProblem prob{pts2d, pts3d, wgt, K, pn};
  ceres::TinySolver<StackedProblem> solver;
  solver.options.max_num_iterations = max_iter;
  solver.options.gradient_tolerance = 1e-14;
  solver.options.parameter_tolerance = 1e-14;
  Eigen::Matrix<double, 6, 1> x;
  for (int k = 0; k < 6; ++k) x[k] = init[k];
  solver.Solve(prob, &x);
  for (int k = 0; k < 6; ++k) result[k] = x[k];
}
}
