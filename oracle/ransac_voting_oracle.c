/*
 * ORACLE — test infrastructure only.
 * Scalar CPU restatement of the ARITHMETIC of the four CUDA kernels in
 * core/csrc/ransac_voting/src/ransac_voting_kernel.cu:
 *   generate_hypothesis_kernel :22-48, voting_for_hypothesis_kernel :100-125,
 *   generate_hypothesis_vanishing_point_kernel :181-228,
 *   voting_for_hypothesis_vanishing_point_kernel :280-309.
 * fp32 evaluation, left-to-right, no FMA (build with -ffp-contract=off);
 * `x<1e-6` compares after promotion to double exactly as the C++ source does.
 * PINNED by the reference's own kernels: the .cu cannot be built as CUDA here (no nvcc) and the reference has no golden
 * vectors, but the four kernel bodies are plain C per thread; oracle/ref_shims/ compiles them verbatim for
 * the host (libransac_ref.so) and tests/golden/ransac_golden.npz holds their outputs — this file reproduces them bit for bit.
 */
#include <math.h>
#include <string.h>

void oracle_generate_hypothesis(const float* direct, const float* coords, const int* idxs, float* hypo_pts, int tn,
                                int vn, int hn) {
  memset(hypo_pts, 0, sizeof(float) * hn * vn * 2); /* at::zeros, kernel.cu:75 */
  for (int hvi = 0; hvi < hn * vn; ++hvi) {
    int hi = hvi / vn, vi = hvi - hi * vn;
    int t0 = idxs[hi * vn * 2 + vi * 2], t1 = idxs[hi * vn * 2 + vi * 2 + 1];
    float nx0 = direct[t0 * vn * 2 + vi * 2 + 1];
    float ny0 = -direct[t0 * vn * 2 + vi * 2];
    float cx0 = coords[t0 * 2], cy0 = coords[t0 * 2 + 1];
    float nx1 = direct[t1 * vn * 2 + vi * 2 + 1];
    float ny1 = -direct[t1 * vn * 2 + vi * 2];
    float cx1 = coords[t1 * 2], cy1 = coords[t1 * 2 + 1];
    if (fabsf(nx1 * ny0 - nx0 * ny1) < 1e-6) continue;
    if (fabsf(ny1 * nx0 - ny0 * nx1) < 1e-6) continue;
    float y = (nx1 * (nx0 * cx0 + ny0 * cy0) - nx0 * (nx1 * cx1 + ny1 * cy1)) / (nx1 * ny0 - nx0 * ny1);
    float x = (ny1 * (nx0 * cx0 + ny0 * cy0) - ny0 * (nx1 * cx1 + ny1 * cy1)) / (ny1 * nx0 - ny0 * nx1);
    hypo_pts[hi * vn * 2 + vi * 2] = x;
    hypo_pts[hi * vn * 2 + vi * 2 + 1] = y;
  }
}

/* inliers must be pre-zeroed by the caller (ransac_voting_gpu.py:58) */
void oracle_voting_for_hypothesis(const float* direct, const float* coords, const float* hypo_pts,
                                  unsigned char* inliers, int tn, int vn, int hn, float inlier_thresh) {
  for (int hi = 0; hi < hn; ++hi)
    for (int vti = 0; vti < vn * tn; ++vti) {
      int vi = vti / tn, ti = vti - vi * tn;
      float cx = coords[ti * 2], cy = coords[ti * 2 + 1];
      float hx = hypo_pts[hi * vn * 2 + vi * 2], hy = hypo_pts[hi * vn * 2 + vi * 2 + 1];
      float nx = direct[ti * vn * 2 + vi * 2], ny = direct[ti * vn * 2 + vi * 2 + 1];
      float dx = hx - cx, dy = hy - cy;
      float norm1 = sqrtf(nx * nx + ny * ny);
      float norm2 = sqrtf(dx * dx + dy * dy);
      if (norm1 < 1e-6 || norm2 < 1e-6) continue;
      float angle_dist = (dx * nx + dy * ny) / (norm1 * norm2);
      if (angle_dist > inlier_thresh) inliers[(size_t)hi * vn * tn + vi * tn + ti] = 1;
    }
}

void oracle_generate_hypothesis_vanishing_point(const float* direct, const float* coords, const int* idxs,
                                                float* hypo_pts, int tn, int vn, int hn) {
  for (int hvi = 0; hvi < hn * vn; ++hvi) {
    int hi = hvi / vn, vi = hvi - hi * vn;
    int id0 = idxs[hi * vn * 2 + vi * 2], id1 = idxs[hi * vn * 2 + vi * 2 + 1];
    float dx0 = direct[id0 * vn * 2 + vi * 2], dy0 = direct[id0 * vn * 2 + vi * 2 + 1];
    float cx0 = coords[id0 * 2], cy0 = coords[id0 * 2 + 1];
    float dx1 = direct[id1 * vn * 2 + vi * 2], dy1 = direct[id1 * vn * 2 + vi * 2 + 1];
    float cx1 = coords[id1 * 2], cy1 = coords[id1 * 2 + 1];
    float lx0 = dy0, ly0 = -dx0, lz0 = cy0 * dx0 - cx0 * dy0;
    float lx1 = dy1, ly1 = -dx1, lz1 = cy1 * dx1 - cx1 * dy1;
    float x = ly0 * lz1 - lz0 * ly1;
    float y = lz0 * lx1 - lx0 * lz1;
    float z = lx0 * ly1 - ly0 * lx1;
    float val_x0 = dx0 * (x - z * cx0);
    float val_x1 = dx1 * (x - z * cx1);
    float val_y0 = dy0 * (y - z * cy0);
    float val_y1 = dy1 * (y - z * cy1);
    if (val_x0 < 0 && val_x1 < 0 && val_y0 < 0 && val_y1 < 0) { z = -z; x = -x; y = -y; }
    if (val_x0 * val_x1 < 0 || val_y0 * val_y1 < 0) { x = 0.f; y = 0.f; z = 0.f; }
    hypo_pts[hi * vn * 3 + vi * 3] = x;
    hypo_pts[hi * vn * 3 + vi * 3 + 1] = y;
    hypo_pts[hi * vn * 3 + vi * 3 + 2] = z;
  }
}

void oracle_voting_for_hypothesis_vanishing_point(const float* direct, const float* coords, const float* hypo_pts,
                                                  unsigned char* inliers, int tn, int vn, int hn,
                                                  float inlier_thresh) {
  for (int hi = 0; hi < hn; ++hi)
    for (int vti = 0; vti < vn * tn; ++vti) {
      int vi = vti / tn, ti = vti - vi * tn;
      float cx = coords[ti * 2], cy = coords[ti * 2 + 1];
      float hx = hypo_pts[hi * vn * 3 + vi * 3], hy = hypo_pts[hi * vn * 3 + vi * 3 + 1];
      float hz = hypo_pts[hi * vn * 3 + vi * 3 + 2];
      float direct_x = direct[ti * vn * 2 + vi * 2], direct_y = direct[ti * vn * 2 + vi * 2 + 1];
      float diff_x = hx - cx * hz, diff_y = hy - cy * hz;
      float norm1 = sqrtf(direct_x * direct_x + direct_y * direct_y);
      float norm2 = sqrtf(diff_x * diff_x + diff_y * diff_y);
      if (norm1 < 1e-6 || norm2 < 1e-6) continue;
      float angle_dist = (direct_x * diff_x + direct_y * diff_y) / (norm1 * norm2);
      float val_x = diff_x * direct_x, val_y = diff_y * direct_y;
      if (val_x < 0 || val_y < 0) continue;
      if (fabsf(angle_dist) > inlier_thresh) inliers[(size_t)hi * vn * tn + vi * tn + ti] = 1;
    }
}
