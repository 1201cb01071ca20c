// Depth rasteriser shared by gdrnpp_render_depth and gdrnpp_depth_refine (gfx950).
//
// Replaces the OpenGL pipeline of lib/render_vispy/renderer.py for the 64x64 ROI render:
//   projective_matrix :461-477, draw_model :363-407 (model matrix diag(1,-1,-1,1)*[R|t]),
//   finish :155-182 (z-buffer -> metric depth, background 0, vertical flip).
// Derived geometry (SURVEY.md §8a notes): GL pixel (row j, col i) is sampled at the OpenCV
// image point (u,v) = (i+0.5, j+0.5) under K_crop; the linearised depth equals the camera-space
// Z of the nearest surface hit, kept only for z_near <= Z <= z_far, both faces (no culling).
//
// Formulation (no GL, no per-pixel division by vertex w): with h_i = K * (R v_i + t) the
// homogeneous pixel-space vertices and q = (u, v, 1),
//     w0 = q . (h1 x h2),  w1 = q . (h2 x h0),  w2 = q . (h0 x h1),  D = h0 . (h1 x h2)
// the ray through q hits the triangle iff w0,w1,w2 share a sign, and there Z = D / (w0+w1+w2)
// (exactly the perspective-correct depth GL interpolates).  This also handles triangles that
// cross the camera plane, which GL handles by clipping.
// All arithmetic is fp64 with a fixed evaluation order and no FMA, identical to
// oracle/raster_oracle.c, so coverage and depth are bit-exact against the oracle.
#pragma once
#include <hip/hip_runtime.h>

namespace gdrnpp {

struct TriSetup {
  double e0[3], e1[3], e2[3];  // edge planes (cross products)
  double D;
  int i_lo, i_hi, j_lo, j_hi;  // conservative pixel bbox (inclusive); empty if i_lo > i_hi
};

__device__ __forceinline__ void cross3(const double* a, const double* b, double* o) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}

// K, R row-major (already widened to double); t double
__device__ __forceinline__ void project_vertex(const float* __restrict__ v, const double* K, const double* R,
                                               const double* t, double* h) {
  const double x = v[0], y = v[1], z = v[2];
  const double X = ((R[0] * x + R[1] * y) + R[2] * z) + t[0];
  const double Y = ((R[3] * x + R[4] * y) + R[5] * z) + t[1];
  const double Z = ((R[6] * x + R[7] * y) + R[8] * z) + t[2];
  h[0] = (K[0] * X + K[1] * Y) + K[2] * Z;
  h[1] = (K[3] * X + K[4] * Y) + K[5] * Z;
  h[2] = (K[6] * X + K[7] * Y) + K[8] * Z;
}

__device__ __forceinline__ int clampi(double v, int lo, int hi) {
  v = fmax(v, (double)lo);
  v = fmin(v, (double)hi);
  return (int)v;
}

// Candidate pixels of a triangle.  The conservative pixel bbox of the part of the triangle with Z >= z_near (h[2] is the
// camera-space Z: the last row of K is (0,0,1)): triangles entirely nearer than z_near or farther than z_far cannot produce a
// fragment and are dropped; triangles crossing the near plane are clipped against it (Sutherland-Hodgman on the 3 edges, in
// homogeneous pixel space where interpolation is linear) so that objects straddling the camera plane do not degrade to a
// full-image bbox.  oracle/raster_oracle.c computes the same bbox with the same expressions.
// The bbox rounds OUTWARDS (floor / ceil), so a sub-pixel triangle with no pixel centre inside still names up to 2x2
// candidates.  Coverage is decided by the edge functions alone; a covered centre lies inside [umin, umax] x [vmin, vmax] up
// to the rounding of fp64 (~1e-13 px), so the candidates are intersected with the centres inside that box widened by 1e-6 px:
// the rendered maps are unchanged, and most triangles of a 64x64 render of a 5k-face mesh (0.5 px each) leave before any
// cross product is formed.
// uv0..2: u = h[0] / h[2], v = h[1] / h[2] of the vertices, read only when all three z >= z_near (else h x, y are read).
// Returns false when the triangle has no candidate pixel.
__device__ __forceinline__ bool triangle_bbox(const double* h0, const double* h1, const double* h2, const double* uv0,
                                              const double* uv1, const double* uv2, int res_w, int res_h, double z_near,
                                              double z_far, TriSetup& s) {
  const double zmin = fmin(h0[2], fmin(h1[2], h2[2])), zmax = fmax(h0[2], fmax(h1[2], h2[2]));
  double umin = 1e300, umax = -1e300, vmin = 1e300, vmax = -1e300;
  if (zmin >= z_near) {
    umin = fmin(uv0[0], fmin(uv1[0], uv2[0])); umax = fmax(uv0[0], fmax(uv1[0], uv2[0]));
    vmin = fmin(uv0[1], fmin(uv1[1], uv2[1])); vmax = fmax(uv0[1], fmax(uv1[1], uv2[1]));
  } else {
    const double* hv[3] = {h0, h1, h2};
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      const double* a = hv[e];
      const double* b = hv[(e + 1) % 3];
      const bool ain = a[2] >= z_near, bin = b[2] >= z_near;
      if (ain) {
        const double u = a[0] / a[2], v = a[1] / a[2];
        umin = fmin(umin, u); umax = fmax(umax, u); vmin = fmin(vmin, v); vmax = fmax(vmax, v);
      }
      if (ain != bin) {
        const double t = (z_near - a[2]) / (b[2] - a[2]);
        const double u = (a[0] + t * (b[0] - a[0])) / z_near, v = (a[1] + t * (b[1] - a[1])) / z_near;
        umin = fmin(umin, u); umax = fmax(umax, u); vmin = fmin(vmin, v); vmax = fmax(vmax, v);
      }
    }
  }
  if (zmax < z_near || zmin > z_far || !(umin <= umax)) { s.i_lo = 1; s.i_hi = 0; return false; }
  // pixel centre i+0.5 in [umin,umax]  =>  i in [umin-0.5, umax-0.5].  The oracle rounds this range outwards (floor / ceil);
  // the centres inside the box widened by 1e-6 px are a subset of that (ceil(x - 1e-6) >= floor(x), floor(x + 1e-6) <=
  // ceil(x)) and contain every centre the edge functions can accept: candidates only, see above
  s.i_lo = clampi(ceil(umin - 0.5 - 1e-6), 0, res_w);   // res_w => empty after the hi clamp
  s.i_hi = clampi(floor(umax - 0.5 + 1e-6), -1, res_w - 1);
  s.j_lo = clampi(ceil(vmin - 0.5 - 1e-6), 0, res_h);
  s.j_hi = clampi(floor(vmax - 0.5 + 1e-6), -1, res_h - 1);
  return s.i_lo <= s.i_hi && s.j_lo <= s.j_hi;
}

// Candidate pixels from single-precision projections (uvz = {u, v, z, -} of each vertex, relative error <= 3e-7): the same
// box as triangle_bbox's all-in-front branch, widened by 1e-3 px instead of 1e-6 — still a superset of every centre the fp64
// edge functions can accept, so the rendered maps do not change.  Returns 0 = no candidate, 1 = candidates in s, 2 = the
// triangle is not safely in front of the near plane (or not finite): the caller must take the fp64 path.
__device__ __forceinline__ int triangle_bbox_f32(const float4 a, const float4 b, const float4 c, int res_w, int res_h,
                                                 float z_near, float z_far, TriSetup& s) {
  const float zmin = fminf(a.z, fminf(b.z, c.z)), zmax = fmaxf(a.z, fmaxf(b.z, c.z));
  if (!(zmin >= z_near * 1.00001f)) return 2;          // near-plane clipping (or NaN): exact path
  if (zmin > z_far * 1.00001f) return 0;               // entirely beyond the far plane (zmax >= zmin >= z_near here)
  const float umin = fminf(a.x, fminf(b.x, c.x)), umax = fmaxf(a.x, fmaxf(b.x, c.x));
  const float vmin = fminf(a.y, fminf(b.y, c.y)), vmax = fmaxf(a.y, fmaxf(b.y, c.y));
  if (!(umin <= umax) || !(vmin <= vmax)) return 2;    // NaN / inf projections: exact path decides
  const float lo_u = fmaxf(ceilf(umin - 0.501f), 0.f), hi_u = fminf(floorf(umax - 0.499f), (float)(res_w - 1));
  const float lo_v = fmaxf(ceilf(vmin - 0.501f), 0.f), hi_v = fminf(floorf(vmax - 0.499f), (float)(res_h - 1));
  if (!(lo_u <= hi_u && lo_v <= hi_v)) return 0;
  s.i_lo = (int)lo_u; s.i_hi = (int)hi_u; s.j_lo = (int)lo_v; s.j_hi = (int)hi_v;
  (void)zmax;
  return 1;
}

// edge planes and determinant; false for a degenerate triangle (no fragment)
__device__ __forceinline__ bool triangle_edges(const double* h0, const double* h1, const double* h2, TriSetup& s) {
  cross3(h1, h2, s.e0);
  cross3(h2, h0, s.e1);
  cross3(h0, h1, s.e2);
  s.D = (h0[0] * s.e0[0] + h0[1] * s.e0[1]) + h0[2] * s.e0[2];
  if (!(s.D != 0.0)) { s.i_lo = 1; s.i_hi = 0; return false; }
  return true;
}

__device__ __forceinline__ void setup_triangle(const double* h0, const double* h1, const double* h2, int res_w,
                                               int res_h, double z_near, double z_far, TriSetup& s) {
  double uv0[2] = {0, 0}, uv1[2] = {0, 0}, uv2[2] = {0, 0};
  if (fmin(h0[2], fmin(h1[2], h2[2])) >= z_near) {
    uv0[0] = h0[0] / h0[2]; uv1[0] = h1[0] / h1[2]; uv2[0] = h2[0] / h2[2];
    uv0[1] = h0[1] / h0[2]; uv1[1] = h1[1] / h1[2]; uv2[1] = h2[1] / h2[2];
  }
  s.D = 0.0;
  if (!triangle_bbox(h0, h1, h2, uv0, uv1, uv2, res_w, res_h, z_near, z_far, s)) { s.i_lo = 1; s.i_hi = 0; return; }
  triangle_edges(h0, h1, h2, s);
}

// depth of the triangle at pixel (i,j); returns false when the pixel centre is not covered
__device__ __forceinline__ bool sample_triangle(const TriSetup& s, int i, int j, double z_near, double z_far,
                                                double& Z, double* lam) {
  const double u = (double)i + 0.5, v = (double)j + 0.5;
  const double w0 = (s.e0[0] * u + s.e0[1] * v) + s.e0[2];
  const double w1 = (s.e1[0] * u + s.e1[1] * v) + s.e1[2];
  const double w2 = (s.e2[0] * u + s.e2[1] * v) + s.e2[2];
  const bool pos = (w0 >= 0.0) && (w1 >= 0.0) && (w2 >= 0.0);
  const bool neg = (w0 <= 0.0) && (w1 <= 0.0) && (w2 <= 0.0);
  if (!(pos || neg)) return false;
  const double sum = (w0 + w1) + w2;
  if (sum == 0.0) return false;
  Z = s.D / sum;
  if (!(Z >= z_near && Z <= z_far)) return false;
  if (lam) { lam[0] = w0 / sum; lam[1] = w1 / sum; lam[2] = w2 / sum; }
  return true;
}

// wave-wide broadcast of a triangle set-up from lane `src` (used to rasterise a large triangle with all 64 lanes)
__device__ __forceinline__ TriSetup shfl_setup(const TriSetup& s, int src) {
  TriSetup o;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    o.e0[k] = __shfl(s.e0[k], src, 64);
    o.e1[k] = __shfl(s.e1[k], src, 64);
    o.e2[k] = __shfl(s.e2[k], src, 64);
  }
  o.D = __shfl(s.D, src, 64);
  o.i_lo = __shfl(s.i_lo, src, 64);
  o.i_hi = __shfl(s.i_hi, src, 64);
  o.j_lo = __shfl(s.j_lo, src, 64);
  o.j_hi = __shfl(s.j_hi, src, 64);
  return o;
}

}  // namespace gdrnpp
