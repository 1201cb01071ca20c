/*
 * ORACLE — test infrastructure only.
 * Float64 software rasteriser standing in for the vispy/OpenGL depth render used by
 * process_depth_refine:  lib/render_vispy/renderer.py
 *   projective_matrix :461-477, set_cam :126-130 (near 0.1, far 100),
 *   draw_model :363-407 (u_model = diag(1,-1,-1,1) * [R|t]),
 *   finish :155-182 (z-buffer -> metric depth, background -> 0, rows flipped).
 * Derivation: ndc.x = 2u/w - 1, ndc.y = 1 - 2v/h with (u,v) the OpenCV
 * projection under K; GL samples pixel (row j, col i) at (u,v) = (i+0.5, j+0.5); the
 * linearised depth is the camera-space Z of the nearest hit with near <= Z <= far, no culling.
 * Ray/triangle intersection in homogeneous pixel space h = K (R v + t):
 *   w_k = q . (h_{k+1} x h_{k+2}),  Z = det[h0 h1 h2] / (w0 + w1 + w2),  q = (u, v, 1).
 * PARITY UNPINNED against GL itself (no GL/EGL in this container, and GL rasterisation is
 * implementation-defined at sub-pixel level); the analytic checks in tests/test_raster.py
 * (planes, spheres with closed-form depth) pin the geometry.
 * Build with -ffp-contract=off: the HIP kernel evaluates the same expressions in the same
 * order, so depth maps are compared bit-for-bit.
 */
#include <math.h>
#include <stdlib.h>

static void cross3(const double* a, const double* b, double* o) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}

static void project_vertex(const float* v, const double* K, const double* R, const double* t, double* h) {
  double x = v[0], y = v[1], z = v[2];
  double X = ((R[0] * x + R[1] * y) + R[2] * z) + t[0];
  double Y = ((R[3] * x + R[4] * y) + R[5] * z) + t[1];
  double Z = ((R[6] * x + R[7] * y) + R[8] * z) + t[2];
  h[0] = (K[0] * X + K[1] * Y) + K[2] * Z;
  h[1] = (K[3] * X + K[4] * Y) + K[5] * Z;
  h[2] = (K[6] * X + K[7] * Y) + K[8] * Z;
}

static int clampi(double v, int lo, int hi) {
  v = fmax(v, (double)lo);
  v = fmin(v, (double)hi);
  return (int)v;
}

/* depth f32[res_h*res_w] (0 = background); xyz f32[res_h*res_w*3] object-space hit point or NULL;
 * face_id i32 or NULL */
void oracle_render_depth(const float* verts, const int* faces, int nfaces, const float* Kf, const float* Rf,
                         const double* t, int res_w, int res_h, double z_near, double z_far, float* depth,
                         float* xyz, int* face_id) {
  double K[9], R[9];
  for (int k = 0; k < 9; ++k) { K[k] = Kf[k]; R[k] = Rf[k]; }
  int n = res_w * res_h;
  for (int p = 0; p < n; ++p) {
    depth[p] = INFINITY;
    if (face_id) face_id[p] = -1;
    if (xyz) xyz[3 * p] = xyz[3 * p + 1] = xyz[3 * p + 2] = 0.f;
  }
  for (int f = 0; f < nfaces; ++f) {
    double h0[3], h1[3], h2[3], e0[3], e1[3], e2[3];
    const float* v0 = verts + 3 * faces[3 * f];
    const float* v1 = verts + 3 * faces[3 * f + 1];
    const float* v2 = verts + 3 * faces[3 * f + 2];
    project_vertex(v0, K, R, t, h0);
    project_vertex(v1, K, R, t, h1);
    project_vertex(v2, K, R, t, h2);
    cross3(h1, h2, e0);
    cross3(h2, h0, e1);
    cross3(h0, h1, e2);
    double D = (h0[0] * e0[0] + h0[1] * e0[1]) + h0[2] * e0[2];
    if (!(D != 0.0)) continue;
    /* conservative pixel bbox of the part with Z >= z_near (same expressions as csrc/raster.hpp:setup_triangle):
     * whole triangle in front -> bbox of the projected corners; crossing the near plane -> corners in front plus the
     * edge/near-plane intersections; entirely nearer than z_near or beyond z_far -> no fragment possible */
    double zmin = fmin(h0[2], fmin(h1[2], h2[2])), zmax = fmax(h0[2], fmax(h1[2], h2[2]));
    double umin = 1e300, umax = -1e300, vmin = 1e300, vmax = -1e300;
    if (zmin >= z_near) {
      double u0 = h0[0] / h0[2], u1 = h1[0] / h1[2], u2 = h2[0] / h2[2];
      double w0 = h0[1] / h0[2], w1 = h1[1] / h1[2], w2 = h2[1] / h2[2];
      umin = fmin(u0, fmin(u1, u2)); umax = fmax(u0, fmax(u1, u2));
      vmin = fmin(w0, fmin(w1, w2)); vmax = fmax(w0, fmax(w1, w2));
    } else {
      const double* hvv[3] = {h0, h1, h2};
      for (int e = 0; e < 3; ++e) {
        const double* a = hvv[e];
        const double* b = hvv[(e + 1) % 3];
        int ain = a[2] >= z_near, bin = b[2] >= z_near;
        if (ain) {
          double u = a[0] / a[2], v = a[1] / a[2];
          umin = fmin(umin, u); umax = fmax(umax, u); vmin = fmin(vmin, v); vmax = fmax(vmax, v);
        }
        if (ain != bin) {
          double tt = (z_near - a[2]) / (b[2] - a[2]);
          double u = (a[0] + tt * (b[0] - a[0])) / z_near, v = (a[1] + tt * (b[1] - a[1])) / z_near;
          umin = fmin(umin, u); umax = fmax(umax, u); vmin = fmin(vmin, v); vmax = fmax(vmax, v);
        }
      }
    }
    if (zmax < z_near || zmin > z_far || !(umin <= umax)) continue;
    int i_lo = clampi(floor(umin - 0.5), 0, res_w);
    int i_hi = clampi(ceil(umax - 0.5), -1, res_w - 1);
    int j_lo = clampi(floor(vmin - 0.5), 0, res_h);
    int j_hi = clampi(ceil(vmax - 0.5), -1, res_h - 1);
    for (int j = j_lo; j <= j_hi; ++j)
      for (int i = i_lo; i <= i_hi; ++i) {
        double u = (double)i + 0.5, v = (double)j + 0.5;
        double w0 = (e0[0] * u + e0[1] * v) + e0[2];
        double w1 = (e1[0] * u + e1[1] * v) + e1[2];
        double w2 = (e2[0] * u + e2[1] * v) + e2[2];
        int pos = (w0 >= 0.0) && (w1 >= 0.0) && (w2 >= 0.0);
        int neg = (w0 <= 0.0) && (w1 <= 0.0) && (w2 <= 0.0);
        if (!(pos || neg)) continue;
        double sum = (w0 + w1) + w2;
        if (sum == 0.0) continue;
        double Z = D / sum;
        if (!(Z >= z_near && Z <= z_far)) continue;
        float zf = (float)Z;
        int p = j * res_w + i;
        /* nearest wins; equal float depth -> lowest face id (matches the (Zbits<<32|face) key) */
        if (zf < depth[p]) {
          depth[p] = zf;
          if (face_id) face_id[p] = f;
          if (xyz) {
            double l0 = w0 / sum, l1 = w1 / sum, l2 = w2 / sum;
            for (int c = 0; c < 3; ++c)
              xyz[3 * p + c] = (float)((l0 * (double)v0[c] + l1 * (double)v1[c]) + l2 * (double)v2[c]);
          }
        }
      }
  }
  for (int p = 0; p < n; ++p)
    if (isinf(depth[p])) depth[p] = 0.f;
}
