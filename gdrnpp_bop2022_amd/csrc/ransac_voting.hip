// PVNet-style RANSAC voting on gfx950 — SURVEY.md §8 row a10.
//
// Behavioural spec (arithmetic only; launch geometry is ours):
//   core/csrc/ransac_voting/src/ransac_voting_kernel.cu
//     generate_hypothesis_kernel                   :11-49
//     voting_for_hypothesis_kernel                 :88-126
//     generate_hypothesis_vanishing_point_kernel   :170-229
//     voting_for_hypothesis_vanishing_point_kernel :268-310
//   torch-extension surface: src/ransac_voting.cpp:30-41,51-65,74-85,95-109
// Evaluation order, fp32 rounding, the float->double promotion in the `<1e-6`
// tests and IEEE sqrt/div are kept, FMA contraction is off, so inlier flags and
// counts are bit-exact against oracle/ransac_voting_oracle.c for identical
// hypotheses.
//
// The reference maps (hi, vi*tn+ti) onto a (1,1024)-thread block
// (cuda_common.h:35-55); here the pixel index ti is the fastest-varying lane
// index so that coords/direct reads and the u8 inlier writes coalesce, and the
// fused vote+count kernel stages one keypoint's pixel data (cx,cy,nx,ny) in LDS
// and never materialises the [hn,vn,tn] flag tensor (4.7 MB per ROI-round).
#include "common.hpp"

namespace {

__global__ void generate_hypothesis_kernel(const float* __restrict__ direct, const float* __restrict__ coords,
                                           const int* __restrict__ idxs, float* __restrict__ hypo_pts, int tn,
                                           int vn, int hn) {
  const int hvi = blockIdx.x * blockDim.x + threadIdx.x;
  if (hvi >= hn * vn) return;
  const int hi = hvi / vn, vi = hvi - hi * vn;
  const int t0 = idxs[hvi * 2], t1 = idxs[hvi * 2 + 1];

  const float nx0 = direct[(t0 * vn + vi) * 2 + 1];
  const float ny0 = -direct[(t0 * vn + vi) * 2];
  const float cx0 = coords[t0 * 2], cy0 = coords[t0 * 2 + 1];
  const float nx1 = direct[(t1 * vn + vi) * 2 + 1];
  const float ny1 = -direct[(t1 * vn + vi) * 2];
  const float cx1 = coords[t1 * 2], cy1 = coords[t1 * 2 + 1];

  float x = 0.f, y = 0.f;  // at::zeros + early return (kernel.cu:42-43,75)
  const float det_y = nx1 * ny0 - nx0 * ny1;
  const float det_x = ny1 * nx0 - ny0 * nx1;
  if (!((double)fabsf(det_y) < 1e-6) && !((double)fabsf(det_x) < 1e-6)) {
    const float a0 = nx0 * cx0 + ny0 * cy0;
    const float a1 = nx1 * cx1 + ny1 * cy1;
    y = (nx1 * a0 - nx0 * a1) / det_y;
    x = (ny1 * a0 - ny0 * a1) / det_x;
  }
  hypo_pts[hvi * 2] = x;
  hypo_pts[hvi * 2 + 1] = y;
  (void)hi;
}

__global__ void generate_hypothesis_vp_kernel(const float* __restrict__ direct, const float* __restrict__ coords,
                                              const int* __restrict__ idxs, float* __restrict__ hypo_pts, int tn,
                                              int vn, int hn) {
  const int hvi = blockIdx.x * blockDim.x + threadIdx.x;
  if (hvi >= hn * vn) return;
  const int hi = hvi / vn, vi = hvi - hi * vn;
  const int id0 = idxs[hvi * 2], id1 = idxs[hvi * 2 + 1];

  const float dx0 = direct[(id0 * vn + vi) * 2], dy0 = direct[(id0 * vn + vi) * 2 + 1];
  const float cx0 = coords[id0 * 2], cy0 = coords[id0 * 2 + 1];
  const float dx1 = direct[(id1 * vn + vi) * 2], dy1 = direct[(id1 * vn + vi) * 2 + 1];
  const float cx1 = coords[id1 * 2], cy1 = coords[id1 * 2 + 1];

  const float lx0 = dy0, ly0 = -dx0, lz0 = cy0 * dx0 - cx0 * dy0;
  const float lx1 = dy1, ly1 = -dx1, lz1 = cy1 * dx1 - cx1 * dy1;

  float x = ly0 * lz1 - lz0 * ly1;
  float y = lz0 * lx1 - lx0 * lz1;
  float z = lx0 * ly1 - ly0 * lx1;

  const float val_x0 = dx0 * (x - z * cx0);
  const float val_x1 = dx1 * (x - z * cx1);
  const float val_y0 = dy0 * (y - z * cy0);
  const float val_y1 = dy1 * (y - z * cy1);

  if (val_x0 < 0 && val_x1 < 0 && val_y0 < 0 && val_y1 < 0) { z = -z; x = -x; y = -y; }
  if (val_x0 * val_x1 < 0 || val_y0 * val_y1 < 0) { x = 0.f; y = 0.f; z = 0.f; }

  hypo_pts[hvi * 3] = x;
  hypo_pts[hvi * 3 + 1] = y;
  hypo_pts[hvi * 3 + 2] = z;
  (void)hi;
}

// one inlier decision; HOMO = vanishing-point (homogeneous hypothesis) variant
template <bool HOMO>
__device__ __forceinline__ bool is_inlier(float cx, float cy, float nx, float ny, float hx, float hy, float hz,
                                          float thresh) {
  float dx, dy;
  if (HOMO) { dx = hx - cx * hz; dy = hy - cy * hz; }
  else      { dx = hx - cx;      dy = hy - cy; }
  const float norm1 = sqrtf(nx * nx + ny * ny);
  const float norm2 = sqrtf(dx * dx + dy * dy);
  if ((double)norm1 < 1e-6 || (double)norm2 < 1e-6) return false;
  if (HOMO) {
    // kernel.cu:303-309 (operand order direct*diff kept)
    const float angle_dist = (nx * dx + ny * dy) / (norm1 * norm2);
    const float val_x = dx * nx, val_y = dy * ny;
    if (val_x < 0 || val_y < 0) return false;
    return fabsf(angle_dist) > thresh;
  } else {
    const float angle_dist = (dx * nx + dy * ny) / (norm1 * norm2);  // kernel.cu:123
    return angle_dist > thresh;
  }
}

// grid: (ceil(tn/256), vn, hn); writes only 1s (inliers pre-zeroed by the caller)
template <bool HOMO>
__global__ void voting_kernel(const float* __restrict__ direct, const float* __restrict__ coords,
                              const float* __restrict__ hypo_pts, unsigned char* __restrict__ inliers, int tn,
                              int vn, int hn, float thresh) {
  const int ti = blockIdx.x * blockDim.x + threadIdx.x;
  const int vi = blockIdx.y, hi = blockIdx.z;
  if (ti >= tn) return;
  const float cx = coords[ti * 2], cy = coords[ti * 2 + 1];
  const float nx = direct[(ti * vn + vi) * 2], ny = direct[(ti * vn + vi) * 2 + 1];
  const int hs = HOMO ? 3 : 2;
  const float* h = hypo_pts + (size_t)(hi * vn + vi) * hs;
  const float hz = HOMO ? h[2] : 1.f;
  if (is_inlier<HOMO>(cx, cy, nx, ny, h[0], h[1], hz, thresh))
    inliers[((size_t)hi * vn + vi) * tn + ti] = 1;
}

constexpr int kPixTile = 4096;  // pixels per LDS tile: 64 KiB as float4 -> 2 workgroups / CU
constexpr int kHypPerWG = 4;    // hypotheses per workgroup (1 per wave): hn*vn/4 workgroups fill the chip

// grid: (vn, ceil(hn/4)); counts[hi,vi] = number of inlier pixels
template <bool HOMO>
__global__ __launch_bounds__(256) void vote_count_kernel(const float* __restrict__ direct,
                                                         const float* __restrict__ coords,
                                                         const float* __restrict__ hypo_pts,
                                                         int* __restrict__ counts, int tn, int vn, int hn,
                                                         float thresh) {
  __shared__ float4 pix[kPixTile];
  const int vi = blockIdx.x;
  const int h0 = blockIdx.y * kHypPerWG;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hs = HOMO ? 3 : 2;

  int cnt[kHypPerWG / 4];
  float hx[kHypPerWG / 4], hy[kHypPerWG / 4], hz[kHypPerWG / 4];
#pragma unroll
  for (int k = 0; k < kHypPerWG / 4; ++k) {
    const int hi = h0 + wave * (kHypPerWG / 4) + k;
    cnt[k] = 0;
    hx[k] = hy[k] = 0.f; hz[k] = 1.f;
    if (hi < hn) {
      const float* h = hypo_pts + (size_t)(hi * vn + vi) * hs;
      hx[k] = h[0]; hy[k] = h[1];
      if (HOMO) hz[k] = h[2];
    }
  }
  for (int t0 = 0; t0 < tn; t0 += kPixTile) {
    const int n = min(kPixTile, tn - t0);
    __syncthreads();
    for (int t = threadIdx.x; t < n; t += 256) {
      const int ti = t0 + t;
      pix[t] = make_float4(coords[ti * 2], coords[ti * 2 + 1], direct[(ti * vn + vi) * 2],
                           direct[(ti * vn + vi) * 2 + 1]);
    }
    __syncthreads();
    for (int t = lane; t < n; t += 64) {
      const float4 p = pix[t];
#pragma unroll
      for (int k = 0; k < kHypPerWG / 4; ++k)
        cnt[k] += is_inlier<HOMO>(p.x, p.y, p.z, p.w, hx[k], hy[k], hz[k], thresh) ? 1 : 0;
    }
  }
#pragma unroll
  for (int k = 0; k < kHypPerWG / 4; ++k) {
    int c = cnt[k];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off, 64);
    const int hi = h0 + wave * (kHypPerWG / 4) + k;
    if (lane == 0 && hi < hn) counts[hi * vn + vi] = c;
  }
}

int check_common(const void* a, const void* b, const void* c, const void* d, int tn, int vn, int hn,
                 const char* who) {
  GDRNPP_REQUIRE(a && b && c && d, GDRNPP_EINVAL, "%s: null pointer", who);
  GDRNPP_REQUIRE(tn > 0 && vn > 0 && hn > 0, GDRNPP_EINVAL, "%s: tn=%d vn=%d hn=%d", who, tn, vn, hn);
  GDRNPP_REQUIRE(vn <= 65535 && hn <= 65535, GDRNPP_ELIMIT, "%s: vn=%d hn=%d above grid limit", who, vn, hn);
  return 0;
}

}  // namespace

extern "C" {

int gdrnpp_generate_hypothesis(const float* direct, const float* coords, const int* idxs, float* hypo_pts, int tn,
                               int vn, int hn, void* stream) {
  if (int rc = check_common(direct, coords, idxs, hypo_pts, tn, vn, hn, "gdrnpp_generate_hypothesis")) return rc;
  hipLaunchKernelGGL(generate_hypothesis_kernel, dim3((hn * vn + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     direct, coords, idxs, hypo_pts, tn, vn, hn);
  return gdrnpp::check_launch("gdrnpp_generate_hypothesis");
}

int gdrnpp_generate_hypothesis_vanishing_point(const float* direct, const float* coords, const int* idxs,
                                               float* hypo_pts, int tn, int vn, int hn, void* stream) {
  if (int rc = check_common(direct, coords, idxs, hypo_pts, tn, vn, hn, "gdrnpp_generate_hypothesis_vanishing_point"))
    return rc;
  hipLaunchKernelGGL(generate_hypothesis_vp_kernel, dim3((hn * vn + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     direct, coords, idxs, hypo_pts, tn, vn, hn);
  return gdrnpp::check_launch("gdrnpp_generate_hypothesis_vanishing_point");
}

int gdrnpp_voting_for_hypothesis(const float* direct, const float* coords, const float* hypo_pts,
                                 unsigned char* inliers, int tn, int vn, int hn, float inlier_thresh, void* stream) {
  if (int rc = check_common(direct, coords, hypo_pts, inliers, tn, vn, hn, "gdrnpp_voting_for_hypothesis")) return rc;
  hipLaunchKernelGGL(voting_kernel<false>, dim3((tn + 255) / 256, vn, hn), dim3(256), 0, (hipStream_t)stream, direct,
                     coords, hypo_pts, inliers, tn, vn, hn, inlier_thresh);
  return gdrnpp::check_launch("gdrnpp_voting_for_hypothesis");
}

int gdrnpp_voting_for_hypothesis_vanishing_point(const float* direct, const float* coords, const float* hypo_pts,
                                                 unsigned char* inliers, int tn, int vn, int hn,
                                                 float inlier_thresh, void* stream) {
  if (int rc = check_common(direct, coords, hypo_pts, inliers, tn, vn, hn,
                            "gdrnpp_voting_for_hypothesis_vanishing_point"))
    return rc;
  hipLaunchKernelGGL(voting_kernel<true>, dim3((tn + 255) / 256, vn, hn), dim3(256), 0, (hipStream_t)stream, direct,
                     coords, hypo_pts, inliers, tn, vn, hn, inlier_thresh);
  return gdrnpp::check_launch("gdrnpp_voting_for_hypothesis_vanishing_point");
}

int gdrnpp_vote_count(const float* direct, const float* coords, const float* hypo_pts, int* counts, int tn, int vn,
                      int hn, float inlier_thresh, int homogeneous, void* stream) {
  if (int rc = check_common(direct, coords, hypo_pts, counts, tn, vn, hn, "gdrnpp_vote_count")) return rc;
  dim3 grid(vn, (hn + kHypPerWG - 1) / kHypPerWG);
  if (homogeneous)
    hipLaunchKernelGGL(vote_count_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, direct, coords, hypo_pts,
                       counts, tn, vn, hn, inlier_thresh);
  else
    hipLaunchKernelGGL(vote_count_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, direct, coords, hypo_pts,
                       counts, tn, vn, hn, inlier_thresh);
  return gdrnpp::check_launch("gdrnpp_vote_count");
}

}  // extern "C"
