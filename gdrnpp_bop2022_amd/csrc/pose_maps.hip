// Map-space kernels of the GDRNPP post-processing — SURVEY.md §8 rows a3.4, a3.5, a4, a6, a8.2, a13.
//
//   gdrnpp_decode_correspondences  engine_utils.py:295-333 (get_out_coor regression branch,
//                                  get_out_mask L1 / BCE) + gdrn_evaluator.py:115-153
//                                  (get_img_model_points_with_coords2d)
//   gdrnpp_pose_from_pred_centroid_z
//                                  rot_reps.py:34-55, pose_from_pred_centroid_z.py:56-154,
//                                  core/utils/utils.py:31-75 (allocentric_to_egocentric) with
//                                  transforms3d.axangles.axangle2mat restated
//   gdrnpp_zoom_K                  camera_geometry.py:6-21 as called at engine_utils.py:260-264
//   gdrnpp_pack_pose_records       fixed 64-byte record replacing the pickled dict list of
//                                  gdrn_evaluator.py:636-665 for the all-gather (§8e)
//
// Selection masks / correspondence order are bit-exact: every compare uses the
// same fp32 operands the NumPy code forms (NEP-50 float32 semantics for the
// `0.0001*extent` and `mask>thr` scalars), IEEE division, no FMA.
#include "common.hpp"
#include <cfloat>

namespace {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;

// ---------------------------------------------------------------------------------------
// decode + compaction: one workgroup per ROI.  Wave w owns the contiguous pixel range
// [w*chunk, (w+1)*chunk); lane l of iteration k handles pixel w*chunk + 64k + l, so every
// wave-level access is one coalesced 256-byte row.  Row-major order of the boolean-mask
// gather is reproduced by ballot + popcount ranks with a 4-entry cross-wave scan.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void decode_corr_kernel(
    const float* __restrict__ coor_x, const float* __restrict__ coor_y, const float* __restrict__ coor_z,
    const float* __restrict__ mask_raw, const float* __restrict__ coord2d, const float* __restrict__ extent,
    const float* __restrict__ imwh, float* __restrict__ out_mask, int* __restrict__ count,
    int* __restrict__ sel_idx, float* __restrict__ img_pts, float* __restrict__ mdl_pts, int hw, int mask_type,
    float mask_thr) {
  __shared__ float s_min[kWaves], s_max[kWaves];
  __shared__ int s_cnt[kWaves];
  const int bi = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* cx = coor_x + (size_t)bi * hw;
  const float* cy = coor_y + (size_t)bi * hw;
  const float* cz = coor_z + (size_t)bi * hw;
  const float* mk = mask_raw + (size_t)bi * hw;
  const float* c2 = coord2d + (size_t)bi * 2 * hw;
  const float e0 = extent[bi * 3], e1 = extent[bi * 3 + 1], e2 = extent[bi * 3 + 2];
  const float imW = imwh[bi * 2], imH = imwh[bi * 2 + 1];

  float mmin = 0.f, mden = 1.f;
  if (mask_type == 0) {
    float lo = FLT_MAX, hi = -FLT_MAX;
    bool has_nan = false;
    for (int i = threadIdx.x; i < hw; i += kThreads) {
      float v = mk[i];
      has_nan |= (v != v);
      lo = fminf(lo, v);
      hi = fmaxf(hi, v);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      lo = fminf(lo, __shfl_xor(lo, off, 64));
      hi = fmaxf(hi, __shfl_xor(hi, off, 64));
    }
    const bool any_nan = __any(has_nan);
    if (lane == 0) { s_min[wave] = any_nan ? NAN : lo; s_max[wave] = any_nan ? NAN : hi; }
    __syncthreads();
    lo = s_min[0]; hi = s_max[0];
    for (int w = 1; w < kWaves; ++w) {  // torch.max/min propagate NaN
      lo = (s_min[w] != s_min[w] || lo != lo) ? NAN : fminf(lo, s_min[w]);
      hi = (s_max[w] != s_max[w] || hi != hi) ? NAN : fmaxf(hi, s_max[w]);
    }
    mmin = lo;
    mden = hi - lo;  // no epsilon (engine_utils.py:325): constant map -> 0/0 = NaN -> nothing selected
  }

  const float t0 = 0.0001f * e0, t1 = 0.0001f * e1, t2 = 0.0001f * e2;
  const int chunk = ((hw + kThreads - 1) / kThreads) * 64;  // pixels per wave, multiple of 64
  const int begin = wave * chunk, end = min(hw, begin + chunk);

  // pass 1: count
  int my_cnt = 0;
  for (int p0 = begin; p0 < end; p0 += 64) {
    const int p = p0 + lane;
    bool sel = false;
    if (p < end) {
      float m = mk[p];
      if (mask_type == 0) m = (m - mmin) / mden;
      else if (mask_type == 1) m = 1.f / (1.f + expf(-m));  // 2: already a probability / label (CE argmax)
      if (out_mask) out_mask[(size_t)bi * hw + p] = m;
      const float x = (cx[p] - 0.5f) * e0, y = (cy[p] - 0.5f) * e1, z = (cz[p] - 0.5f) * e2;
      sel = (m > mask_thr) && (fabsf(x) > t0) && (fabsf(y) > t1) && (fabsf(z) > t2);
    }
    my_cnt += __popcll(__ballot(sel));
  }
  if (lane == 0) s_cnt[wave] = my_cnt;
  __syncthreads();
  int base = 0, total = 0;
  for (int w = 0; w < kWaves; ++w) {
    if (w < wave) base += s_cnt[w];
    total += s_cnt[w];
  }
  if (threadIdx.x == 0) count[bi] = total;

  // pass 2: ranks + scatter (same arithmetic as pass 1; inputs come from L1/L2)
  for (int p0 = begin; p0 < end; p0 += 64) {
    const int p = p0 + lane;
    bool sel = false;
    float x = 0.f, y = 0.f, z = 0.f;
    if (p < end) {
      float m = mk[p];
      if (mask_type == 0) m = (m - mmin) / mden;
      else if (mask_type == 1) m = 1.f / (1.f + expf(-m));  // 2: already a probability / label (CE argmax)
      x = (cx[p] - 0.5f) * e0; y = (cy[p] - 0.5f) * e1; z = (cz[p] - 0.5f) * e2;
      sel = (m > mask_thr) && (fabsf(x) > t0) && (fabsf(y) > t1) && (fabsf(z) > t2);
    }
    const unsigned long long bal = __ballot(sel);
    if (sel) {
      const int r = base + __popcll(bal & ((1ull << lane) - 1ull));
      const size_t o = (size_t)bi * hw + r;
      if (sel_idx) sel_idx[o] = p;
      img_pts[o * 2] = c2[p] * imW;
      img_pts[o * 2 + 1] = c2[hw + p] * imH;
      mdl_pts[o * 3] = x; mdl_pts[o * 3 + 1] = y; mdl_pts[o * 3 + 2] = z;
    }
    base += __popcll(bal);
  }
}

// ---------------------------------------------------------------------------------------
// The same decode for maps of at most 4096 pixels (OUTPUT_RES 64): 16 waves per ROI, every pixel's inputs are loaded
// once, all loads of a lane in flight together, and stay in registers from the mask normalisation to the scatter; two
// barriers instead of three passes over the maps.  Pixel ranges per wave, ballot ranks and arithmetic as above, so the
// selection, its order and the values are identical.
// ---------------------------------------------------------------------------------------
constexpr int kRegThreads = 1024, kRegWaves = kRegThreads / 64, kRegIt = 4, kRegMaxHw = kRegThreads * kRegIt;

__global__ __launch_bounds__(kRegThreads) void decode_corr_reg_kernel(
    const float* __restrict__ coor_x, const float* __restrict__ coor_y, const float* __restrict__ coor_z,
    const float* __restrict__ mask_raw, const float* __restrict__ coord2d, const float* __restrict__ extent,
    const float* __restrict__ imwh, float* __restrict__ out_mask, int* __restrict__ count,
    int* __restrict__ sel_idx, float* __restrict__ img_pts, float* __restrict__ mdl_pts, int hw, int mask_type,
    float mask_thr) {
  __shared__ float s_min[kRegWaves], s_max[kRegWaves];
  __shared__ int s_cnt[kRegWaves];
  const int bi = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* cx = coor_x + (size_t)bi * hw;
  const float* cy = coor_y + (size_t)bi * hw;
  const float* cz = coor_z + (size_t)bi * hw;
  const float* mk = mask_raw + (size_t)bi * hw;
  const float* c2 = coord2d + (size_t)bi * 2 * hw;
  const int chunk = ((hw + kRegThreads - 1) / kRegThreads) * 64;  // pixels per wave, multiple of 64, <= 64 * kRegIt
  const int begin = wave * chunk, end = min(hw, begin + chunk);

  float mv[kRegIt], xv[kRegIt], yv[kRegIt], zv[kRegIt], ux[kRegIt], uy[kRegIt];
  bool in[kRegIt];
#pragma unroll
  for (int j = 0; j < kRegIt; ++j) {
    const int p = begin + 64 * j + lane;
    in[j] = p < end;
    const int q = in[j] ? p : 0;
    mv[j] = mk[q]; xv[j] = cx[q]; yv[j] = cy[q]; zv[j] = cz[q]; ux[j] = c2[q]; uy[j] = c2[hw + q];
  }
  const float e0 = extent[bi * 3], e1 = extent[bi * 3 + 1], e2 = extent[bi * 3 + 2];
  const float imW = imwh[bi * 2], imH = imwh[bi * 2 + 1];

  float mmin = 0.f, mden = 1.f;
  if (mask_type == 0) {
    float lo = FLT_MAX, hi = -FLT_MAX;
    bool has_nan = false;
#pragma unroll
    for (int j = 0; j < kRegIt; ++j)
      if (in[j]) {
        has_nan |= (mv[j] != mv[j]);
        lo = fminf(lo, mv[j]);
        hi = fmaxf(hi, mv[j]);
      }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      lo = fminf(lo, __shfl_xor(lo, off, 64));
      hi = fmaxf(hi, __shfl_xor(hi, off, 64));
    }
    const bool any_nan = __any(has_nan);
    if (lane == 0) { s_min[wave] = any_nan ? NAN : lo; s_max[wave] = any_nan ? NAN : hi; }
    __syncthreads();
    lo = s_min[0]; hi = s_max[0];
    for (int w = 1; w < kRegWaves; ++w) {  // torch.max/min propagate NaN
      lo = (s_min[w] != s_min[w] || lo != lo) ? NAN : fminf(lo, s_min[w]);
      hi = (s_max[w] != s_max[w] || hi != hi) ? NAN : fmaxf(hi, s_max[w]);
    }
    mmin = lo;
    mden = hi - lo;
  }

  const float t0 = 0.0001f * e0, t1 = 0.0001f * e1, t2 = 0.0001f * e2;
  unsigned long long bal[kRegIt];
  int my_cnt = 0;
#pragma unroll
  for (int j = 0; j < kRegIt; ++j) {
    bool sel = false;
    if (in[j]) {
      float m = mv[j];
      if (mask_type == 0) m = (m - mmin) / mden;
      else if (mask_type == 1) m = 1.f / (1.f + expf(-m));
      if (out_mask) out_mask[(size_t)bi * hw + begin + 64 * j + lane] = m;
      xv[j] = (xv[j] - 0.5f) * e0; yv[j] = (yv[j] - 0.5f) * e1; zv[j] = (zv[j] - 0.5f) * e2;
      sel = (m > mask_thr) && (fabsf(xv[j]) > t0) && (fabsf(yv[j]) > t1) && (fabsf(zv[j]) > t2);
    }
    bal[j] = __ballot(sel);
    my_cnt += __popcll(bal[j]);
  }
  if (lane == 0) s_cnt[wave] = my_cnt;
  __syncthreads();
  int base = 0, total = 0;
  for (int w = 0; w < kRegWaves; ++w) {
    if (w < wave) base += s_cnt[w];
    total += s_cnt[w];
  }
  if (threadIdx.x == 0) count[bi] = total;
#pragma unroll
  for (int j = 0; j < kRegIt; ++j) {
    if ((bal[j] >> lane) & 1ull) {
      const int r = base + __popcll(bal[j] & ((1ull << lane) - 1ull));
      const size_t o = (size_t)bi * hw + r;
      if (sel_idx) sel_idx[o] = begin + 64 * j + lane;
      img_pts[o * 2] = ux[j] * imW;
      img_pts[o * 2 + 1] = uy[j] * imH;
      mdl_pts[o * 3] = xv[j]; mdl_pts[o * 3 + 1] = yv[j]; mdl_pts[o * 3 + 2] = zv[j];
    }
    base += __popcll(bal[j]);
  }
}

// ---------------------------------------------------------------------------------------
// rot_mode: 0 = 6-d representation, 1 = quaternion (w,x,y,z), 2 = rotation matrix (row-major), 3 = log-quaternion (3 values:
//           quaternion_lf.qexp then quat2mat_torch), 4 = Lie vector / angle-axis (3 values: lie_algebra.lie_vec_to_rot)
// t_mode:   0 = centroid_z with relative z (SITE), 1 = centroid_z with absolute z, 2 = centroid_z_abs (absolute 2-d centre
//           and z: pose_from_pred_centroid_z_abs.py:44-76), 3 = trans (the head's output IS the translation: pose_from_pred.py:25-27)
// rot_in / t_: the ROI's OWN rows of the head outputs (already offset); the other arrays are indexed by the ROI number i
__device__ void pose_from_pred_one(const float* rot_in, const float* t_,
                                   const float* __restrict__ cams, const float* __restrict__ centers,
                                   const float* __restrict__ whs, const float* __restrict__ resize_ratios,
                                   float* __restrict__ rot, float* __restrict__ trans, int i, int t_mode,
                                   int is_allo, int rot_mode) {
  float Ra[9];
  if (rot_mode == 0) {
    // rot6d_to_mat_batch (rot_reps.py:34-55): x = normalize(a), z = normalize(x X b), y = z X x; columns (x,y,z)
    const float* d6 = rot_in;
    float ax = d6[0], ay = d6[1], az = d6[2], bx = d6[3], by = d6[4], bz = d6[5];
    float na = fmaxf(sqrtf((ax * ax + ay * ay) + az * az), 1e-12f);  // F.normalize eps
    float x0 = ax / na, x1 = ay / na, x2 = az / na;
    float z0 = x1 * bz - x2 * by, z1 = x2 * bx - x0 * bz, z2 = x0 * by - x1 * bx;
    float nz = fmaxf(sqrtf((z0 * z0 + z1 * z1) + z2 * z2), 1e-12f);
    z0 /= nz; z1 /= nz; z2 /= nz;
    float y0 = z1 * x2 - z2 * x1, y1 = z2 * x0 - z0 * x2, y2 = z0 * x1 - z1 * x0;
    Ra[0] = x0; Ra[1] = y0; Ra[2] = z0; Ra[3] = x1; Ra[4] = y1; Ra[5] = z1; Ra[6] = x2; Ra[7] = y2; Ra[8] = z2;
  } else if (rot_mode == 1 || rot_mode == 3) {
    float q[4];
    if (rot_mode == 1) {
      for (int k = 0; k < 4; ++k) q[k] = rot_in[k];
    } else {
      // quaternion_lf.qexp on a 3-vector (core/utils/quaternion_lf.py:294-318): s = 0, theta = |v|, exp(q) = (cos theta,
      // sin theta / max(theta, 1e-8) * v); get_rot_mat feeds it to quat2mat_torch (model_utils.py:350-352)
      const float* v = rot_in;
      const float theta = sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
      const float k = 1.0f / fmaxf(theta, 1e-8f) * sinf(theta);
      q[0] = cosf(theta); q[1] = k * v[0]; q[2] = k * v[1]; q[3] = k * v[2];
    }
    // quat2mat_torch (core/utils/pose_utils.py:349-400, eps = 0): normalise, then the (w,x,y,z) -> matrix polynomial
    const float nq = sqrtf(((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3]);
    const float qw = q[0] / nq, qx = q[1] / nq, qy = q[2] / nq, qz = q[3] / nq;
    const float X = qx * 2.f, Y = qy * 2.f, Z = qz * 2.f;
    const float wX = qw * X, wY = qw * Y, wZ = qw * Z, xX = qx * X, xY = qx * Y, xZ = qx * Z, yY = qy * Y, yZ = qy * Z, zZ = qz * Z;
    Ra[0] = 1.f - (yY + zZ); Ra[1] = xY - wZ; Ra[2] = xZ + wY;
    Ra[3] = xY + wZ; Ra[4] = 1.f - (xX + zZ); Ra[5] = yZ - wX;
    Ra[6] = xZ - wY; Ra[7] = yZ + wX; Ra[8] = 1.f - (xX + yY);
  } else if (rot_mode == 4) {
    // lie_algebra.lie_vec_to_rot (core/utils/lie_algebra.py:7-77, kornia's angle_axis_to_rotation_matrix): Rodrigues with the
    // axis divided by (theta + 1e-6); theta^2 <= 1e-6 takes the first-order form I + [r]x
    const float* v = rot_in;
    const float rx = v[0], ry = v[1], rz = v[2];
    const float theta2 = (rx * rx + ry * ry) + rz * rz;
    if (theta2 > 1e-6f) {
      const float theta = sqrtf(theta2);
      const float wx = rx / (theta + 1e-6f), wy = ry / (theta + 1e-6f), wz = rz / (theta + 1e-6f);
      const float c = cosf(theta), sn = sinf(theta), C = 1.0f - c;
      Ra[0] = c + wx * wx * C;       Ra[1] = wx * wy * C - wz * sn; Ra[2] = wy * sn + wx * wz * C;
      Ra[3] = wz * sn + wx * wy * C; Ra[4] = c + wy * wy * C;       Ra[5] = -wx * sn + wy * wz * C;
      Ra[6] = -wy * sn + wx * wz * C; Ra[7] = wx * sn + wy * wz * C; Ra[8] = c + wz * wz * C;
    } else {
      Ra[0] = 1.f; Ra[1] = -rz; Ra[2] = ry; Ra[3] = rz; Ra[4] = 1.f; Ra[5] = -rx; Ra[6] = -ry; Ra[7] = rx; Ra[8] = 1.f;
    }
  } else {
    for (int k = 0; k < 9; ++k) Ra[k] = rot_in[k];
  }

  // pose_from_predictions_test (pose_from_pred_centroid_z.py:73-110 and its _abs / plain-translation siblings), fp32 like torch
  const float* K = cams + 9 * (size_t)i;
  float tx, ty, z;
  if (t_mode == 3) {
    tx = t_[0]; ty = t_[1]; z = t_[2];
  } else {
    const float cxp = t_mode == 2 ? t_[0] : t_[0] * whs[2 * i] + centers[2 * i];
    const float cyp = t_mode == 2 ? t_[1] : t_[1] * whs[2 * i + 1] + centers[2 * i + 1];
    z = (t_mode == 0) ? t_[2] * resize_ratios[i] : t_[2];
    tx = z * (cxp - K[2]) / K[0];
    ty = z * (cyp - K[5]) / K[4];
  }
  trans[3 * i] = tx; trans[3 * i + 1] = ty; trans[3 * i + 2] = z;

  float* Ro = rot + 9 * (size_t)i;
  if (!is_allo) {
    for (int k = 0; k < 9; ++k) Ro[k] = Ra[k];
    return;
  }
  // allocentric_to_egocentric (utils.py:48-62): float32 obj_ray, float64 angle/axis/matrix, fp32 store
  const float nt = sqrtf((tx * tx + ty * ty) + z * z);
  const float ox = tx / nt, oy = ty / nt, oz = z / nt;
  const double angle = acos((double)oz);
  if (angle > 0) {
    double ux = -(double)oy, uy = (double)ox, uz = 0.0;  // cross((0,0,1), obj_ray)
    const double n = sqrt(ux * ux + uy * uy + uz * uz);
    ux /= n; uy /= n; uz /= n;
    const double c = cos(angle), s = sin(angle), C = 1 - c;
    const double xs = ux * s, ys = uy * s, zs = uz * s;
    const double xC = ux * C, yC = uy * C, zC = uz * C;
    const double xyC = ux * yC, yzC = uy * zC, zxC = uz * xC;
    const double M[9] = {ux * xC + c, xyC - zs, zxC + ys, xyC + zs, uy * yC + c, yzC - xs, zxC - ys, yzC + xs, uz * zC + c};
    for (int r = 0; r < 3; ++r)
      for (int cc = 0; cc < 3; ++cc) {
        double acc = 0.0;
        for (int k = 0; k < 3; ++k) acc += M[r * 3 + k] * (double)Ra[k * 3 + cc];
        Ro[r * 3 + cc] = (float)acc;
      }
  } else {
    for (int k = 0; k < 9; ++k) Ro[k] = Ra[k];
  }
}

__global__ void pose_from_pred_kernel(const float* __restrict__ rot_in, const float* __restrict__ t_,
                                      const float* __restrict__ cams, const float* __restrict__ centers,
                                      const float* __restrict__ whs, const float* __restrict__ resize_ratios,
                                      float* __restrict__ rot, float* __restrict__ trans, int b, int t_mode,
                                      int is_allo, int rot_mode) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= b) return;
  const int rdim = rot_mode == 0 ? 6 : (rot_mode == 1 ? 4 : (rot_mode == 2 ? 9 : 3));
  pose_from_pred_one(rot_in + (size_t)rdim * i, t_ + 3 * (size_t)i, cams, centers, whs, resize_ratios, rot, trans, i, t_mode, is_allo, rot_mode);
}

// Patch-PnP's output layers fc_r | fc_t (conv_pnp_net.py:99-101,178-182) for one ROI per wavefront: lane l multiplies the
// elements k = l, l + 64, ... of the ROI's feature row with each of the rot_dim + 3 weight rows, the partial sums meet in a
// fixed-order butterfly (deterministic).  K <= 1024, rot_dim <= 9.  PoseArgs.rot != NULL: lane 0 goes on to the pose of its ROI
// (pose_from_pred_one on the values it has just written) — the network's last launch.
struct PoseArgs { const float* cams; const float* centers; const float* whs; const float* resize_ratios; float* rot; float* trans; int t_mode, is_allo, rot_mode; };
__global__ __launch_bounds__(64) void pnp_fc_heads_kernel(const float* __restrict__ x, const float* __restrict__ w_r,
                                                          const float* __restrict__ b_r, const float* __restrict__ w_t,
                                                          const float* __restrict__ b_t, float* __restrict__ rot_,
                                                          float* __restrict__ t_, int K, int rot_dim, PoseArgs pa) {
  const int i = blockIdx.x, lane = threadIdx.x;
  float xv[16];
  const int nk = (K + 63) / 64;
#pragma unroll
  for (int j = 0; j < 16; ++j) xv[j] = (j < nk && lane + 64 * j < K) ? x[(size_t)i * K + lane + 64 * j] : 0.f;
  float outv[12];
  for (int r = 0; r < rot_dim + 3; ++r) {
    const float* w = r < rot_dim ? w_r + (size_t)r * K : w_t + (size_t)(r - rot_dim) * K;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (j < nk && lane + 64 * j < K) s = fmaf(xv[j], w[lane + 64 * j], s);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    s += r < rot_dim ? (b_r ? b_r[r] : 0.f) : (b_t ? b_t[r - rot_dim] : 0.f);
    outv[r] = s;
    if (lane == 0) {
      if (r < rot_dim) rot_[(size_t)i * rot_dim + r] = s;
      else t_[(size_t)i * 3 + (r - rot_dim)] = s;
    }
  }
  if (pa.rot && lane == 0)     // the pose from the values this lane holds in registers
    pose_from_pred_one(outv, outv + rot_dim, pa.cams, pa.centers, pa.whs, pa.resize_ratios, pa.rot, pa.trans, i, pa.t_mode, pa.is_allo, pa.rot_mode);
}

__global__ void zoom_K_kernel(const float* __restrict__ K, const float* __restrict__ centers,
                              const float* __restrict__ scales, float* __restrict__ Kc, int b, float out_res) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= b) return;
  const float* k = K + 9 * (size_t)i;
  float* o = Kc + 9 * (size_t)i;
  const float s = scales[i];
  const float x0 = centers[2 * i] - s / 2, y0 = centers[2 * i + 1] - s / 2;
  const float r = out_res / s;
  o[0] = k[0] * r; o[1] = k[1] * r; o[2] = (k[2] - x0) * r;
  o[3] = k[3] * r; o[4] = k[4] * r; o[5] = (k[5] - y0) * r;
  o[6] = k[6]; o[7] = k[7]; o[8] = k[8];
}

__global__ void pack_records_kernel(const float* __restrict__ R, const double* __restrict__ t_ref,
                                    const float* __restrict__ t_net, const float* __restrict__ score,
                                    const int* __restrict__ obj_id, const int* __restrict__ roi_id,
                                    float* __restrict__ rec, int b) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= b) return;
  float* o = rec + 16 * (size_t)i;
  for (int k = 0; k < 9; ++k) o[k] = R[9 * (size_t)i + k];
  for (int k = 0; k < 3; ++k) o[9 + k] = t_ref ? (float)t_ref[3 * (size_t)i + k] : t_net[3 * (size_t)i + k];
  o[12] = score ? score[i] : 1.f;
  o[13] = obj_id ? (float)obj_id[i] : -1.f;
  o[14] = roi_id ? (float)roi_id[i] : (float)i;
  o[15] = 1.f;
}

}  // namespace

extern "C" {

int gdrnpp_decode_correspondences(const float* coor_x, const float* coor_y, const float* coor_z,
                                  const float* mask_raw, const float* coord2d, const float* extent,
                                  const float* imwh, float* out_mask, int* count, int* sel_idx, float* img_pts,
                                  float* mdl_pts, int b, int hw, int mask_type, float mask_thr, void* stream) {
  if (b == 0) return 0;  // an image / a rank without ROIs: nothing to launch
  GDRNPP_REQUIRE(coor_x && coor_y && coor_z && mask_raw && coord2d && extent && imwh && count && img_pts && mdl_pts,
                 GDRNPP_EINVAL, "gdrnpp_decode_correspondences: null pointer");
  GDRNPP_REQUIRE(b > 0 && hw > 0, GDRNPP_EINVAL, "gdrnpp_decode_correspondences: b=%d hw=%d", b, hw);
  GDRNPP_REQUIRE(mask_type >= 0 && mask_type <= 2, GDRNPP_EINVAL, "gdrnpp_decode_correspondences: mask_type=%d",
                 mask_type);
  if (hw <= kRegMaxHw)
    hipLaunchKernelGGL(decode_corr_reg_kernel, dim3(b), dim3(kRegThreads), 0, (hipStream_t)stream, coor_x, coor_y, coor_z,
                       mask_raw, coord2d, extent, imwh, out_mask, count, sel_idx, img_pts, mdl_pts, hw, mask_type,
                       mask_thr);
  else
    hipLaunchKernelGGL(decode_corr_kernel, dim3(b), dim3(kThreads), 0, (hipStream_t)stream, coor_x, coor_y, coor_z,
                       mask_raw, coord2d, extent, imwh, out_mask, count, sel_idx, img_pts, mdl_pts, hw, mask_type,
                       mask_thr);
  return gdrnpp::check_launch("gdrnpp_decode_correspondences");
}

int gdrnpp_pose_from_pred_centroid_z(const float* rot6d, const float* t_, const float* cams, const float* centers,
                                     const float* whs, const float* resize_ratios, float* rot, float* trans, int b,
                                     int z_type, int is_allo, void* stream) {
  GDRNPP_REQUIRE(rot6d && t_ && cams && centers && whs && resize_ratios && rot && trans, GDRNPP_EINVAL,
                 "gdrnpp_pose_from_pred_centroid_z: null pointer");
  GDRNPP_REQUIRE(b > 0 && (z_type == 0 || z_type == 1), GDRNPP_EINVAL,
                 "gdrnpp_pose_from_pred_centroid_z: b=%d z_type=%d", b, z_type);
  hipLaunchKernelGGL(pose_from_pred_kernel, dim3((b + 63) / 64), dim3(64), 0, (hipStream_t)stream, rot6d, t_, cams,
                     centers, whs, resize_ratios, rot, trans, b, z_type, is_allo, 0);
  return gdrnpp::check_launch("gdrnpp_pose_from_pred_centroid_z");
}

int gdrnpp_pose_from_pred(const float* rot_in, int rot_mode, const float* t_, int t_mode, const float* cams,
                          const float* centers, const float* whs, const float* resize_ratios, float* rot, float* trans,
                          int b, int is_allo, void* stream) {
  if (b == 0) return 0;
  GDRNPP_REQUIRE(rot_in && t_ && cams && rot && trans, GDRNPP_EINVAL, "gdrnpp_pose_from_pred: null pointer");
  GDRNPP_REQUIRE(b > 0 && rot_mode >= 0 && rot_mode <= 4 && t_mode >= 0 && t_mode <= 3, GDRNPP_EINVAL,
                 "gdrnpp_pose_from_pred: b=%d rot_mode=%d t_mode=%d", b, rot_mode, t_mode);
  GDRNPP_REQUIRE(t_mode >= 2 || (centers && whs && (t_mode == 1 || resize_ratios)), GDRNPP_EINVAL,
                 "gdrnpp_pose_from_pred: centroid_z needs centers, whs (and resize_ratios for relative z)");
  hipLaunchKernelGGL(pose_from_pred_kernel, dim3((b + 63) / 64), dim3(64), 0, (hipStream_t)stream, rot_in, t_, cams,
                     centers, whs, resize_ratios, rot, trans, b, t_mode, is_allo, rot_mode);
  return gdrnpp::check_launch("gdrnpp_pose_from_pred");
}

int gdrnpp_pnp_fc_heads(const float* x, const float* w_r, const float* b_r, const float* w_t, const float* b_t, float* rot_,
                        float* t_, int b, int K, int rot_dim, void* stream) {
  if (b == 0) return 0;
  GDRNPP_REQUIRE(x && w_r && w_t && rot_ && t_, GDRNPP_EINVAL, "gdrnpp_pnp_fc_heads: null pointer");
  GDRNPP_REQUIRE(b > 0 && K > 0 && K <= 1024 && rot_dim > 0 && rot_dim <= 9, GDRNPP_ELIMIT,
                 "gdrnpp_pnp_fc_heads: b=%d K=%d (<= 1024) rot_dim=%d (<= 9)", b, K, rot_dim);
  hipLaunchKernelGGL(pnp_fc_heads_kernel, dim3(b), dim3(64), 0, (hipStream_t)stream, x, w_r, b_r, w_t, b_t, rot_, t_, K, rot_dim,
                     PoseArgs{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0});
  return gdrnpp::check_launch("gdrnpp_pnp_fc_heads");
}

int gdrnpp_pnp_fc_heads_pose(const float* x, const float* w_r, const float* b_r, const float* w_t, const float* b_t, float* rot_,
                             float* t_, int b, int K, int rot_mode, int t_mode, const float* cams, const float* centers,
                             const float* whs, const float* resize_ratios, float* rot, float* trans, int is_allo, void* stream) {
  if (b == 0) return 0;
  GDRNPP_REQUIRE(x && w_r && w_t && rot_ && t_ && cams && rot && trans, GDRNPP_EINVAL, "gdrnpp_pnp_fc_heads_pose: null pointer");
  GDRNPP_REQUIRE(b > 0 && K > 0 && K <= 1024 && rot_mode >= 0 && rot_mode <= 4 && t_mode >= 0 && t_mode <= 3, GDRNPP_EINVAL,
                 "gdrnpp_pnp_fc_heads_pose: b=%d K=%d rot_mode=%d t_mode=%d", b, K, rot_mode, t_mode);
  GDRNPP_REQUIRE(t_mode >= 2 || (centers && whs && (t_mode == 1 || resize_ratios)), GDRNPP_EINVAL,
                 "gdrnpp_pnp_fc_heads_pose: centroid_z needs centers, whs (and resize_ratios for relative z)");
  const int rot_dim = rot_mode == 0 ? 6 : (rot_mode == 1 ? 4 : (rot_mode == 2 ? 9 : 3));
  hipLaunchKernelGGL(pnp_fc_heads_kernel, dim3(b), dim3(64), 0, (hipStream_t)stream, x, w_r, b_r, w_t, b_t, rot_, t_, K, rot_dim,
                     PoseArgs{cams, centers, whs, resize_ratios, rot, trans, t_mode, is_allo, rot_mode});
  return gdrnpp::check_launch("gdrnpp_pnp_fc_heads_pose");
}

int gdrnpp_zoom_K(const float* K, const float* centers, const float* scales, float* K_crop, int b, float out_res,
                  void* stream) {
  if (b == 0) return 0;
  GDRNPP_REQUIRE(K && centers && scales && K_crop && b > 0, GDRNPP_EINVAL, "gdrnpp_zoom_K: bad arguments");
  hipLaunchKernelGGL(zoom_K_kernel, dim3((b + 63) / 64), dim3(64), 0, (hipStream_t)stream, K, centers, scales,
                     K_crop, b, out_res);
  return gdrnpp::check_launch("gdrnpp_zoom_K");
}

int gdrnpp_pack_pose_records(const float* R, const double* t_refined, const float* t_net, const float* score,
                             const int* obj_id, const int* roi_id, float* rec, int b, void* stream) {
  if (b == 0) return 0;
  GDRNPP_REQUIRE(R && rec && (t_refined || t_net) && b > 0, GDRNPP_EINVAL, "gdrnpp_pack_pose_records: bad arguments");
  hipLaunchKernelGGL(pack_records_kernel, dim3((b + 63) / 64), dim3(64), 0, (hipStream_t)stream, R, t_refined, t_net,
                     score, obj_id, roi_id, rec, b);
  return gdrnpp::check_launch("gdrnpp_pack_pose_records");
}

}  // extern "C"
