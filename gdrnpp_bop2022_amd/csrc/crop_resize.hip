// ROI crop-resize on gfx950 — SURVEY.md §8 row a1 (the "next" row §8f-1: the CPU stage that feeds the network).
//
// Behavioural spec: read_data_test, core/gdrn_modeling/datasets/data_loader.py:754-797:
//   roi_img    = normalize(cv2.warpAffine(image u8 BGR, M, (256,256), INTER_LINEAR).transpose(2,0,1))   :773-778
//   roi_depth  = cv2.warpAffine(depth f32, M, (256,256), INTER_NEAREST)                                    :781-789
//   roi_coord  = cv2.warpAffine(coord_2d f32x2, M64, (64,64), INTER_LINEAR).transpose(2,0,1)               :792-797
// with M = get_affine_transform(bbox_center, scale, 0, out) (core/utils/data_utils.py:136-184) and
// normalize = (x - PIXEL_MEAN)/PIXEL_STD in float64 -> float32 (core/base_data_loader.py:128-135).
// OpenCV's arithmetic (third-party, not in the tree) is restated exactly as in oracle/warp_oracle.c: double LU for
// getAffineTransform, double inverse, cvRound to 10-bit fixed point, 5 fractional bits, 15-bit integer weights for
// u8 (incl. the {32767,0,0,1} entry 0), float weights for f32, border value 0.  Integer paths are bit-exact by
// construction; the float bilinear keeps the left-to-right fp32 sum (no FMA).
//
// Design: grid (pixel tiles, ROI); thread = one output pixel (all channels), x fastest, so the three CHW output
// planes are written as coalesced 256-byte rows per wave — the op is write-bound (1.08 MB/ROI written vs <= 0.16 MB
// of source pixels under the ROI, read through L2).  The coord_2d source map is never materialised: its value at
// integer (x, y) is the analytic float32(x * (1/W)) of get_2d_coord_np (data_utils.py:304-323).
#include "common.hpp"

namespace {

constexpr int kPixPerThread = 16;  // 4096 pixels per workgroup: the per-ROI LU prologue is amortised over 16 rows
constexpr int AB_BITS = 10, AB_SCALE = 1 << AB_BITS, INTER_BITS = 5, INTER_TAB_SIZE = 1 << INTER_BITS;

__device__ __forceinline__ int cv_round(double v) { return __double2int_rn(v); }
__device__ __forceinline__ int sat_short(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

// cv::LUImpl, 6x6, one right-hand side (getAffineTransform).  Every index is a compile-time constant after unrolling (the
// pivot row is applied as a chain of predicated swaps), so the system lives in registers: with a run-time row index the
// arrays went to scratch memory and the solve took 20 us on the one lane that does it.  Same operations in the same order
// as the loop form (first maximal pivot, `>` comparison).
__device__ __forceinline__ bool lu_solve6(double (&A)[36], double (&b)[6]) {
  constexpr int m = 6;
#pragma unroll
  for (int i = 0; i < m; i++) {
    int k = i;
    double best = fabs(A[i * m + i]);
#pragma unroll
    for (int j = i + 1; j < m; j++) {
      const double v = fabs(A[j * m + i]);
      if (v > best) { best = v; k = j; }
    }
    if (best < 2.220446049250313e-16 * 100) return false;
#pragma unroll
    for (int r = i + 1; r < m; r++) {
      const bool sw = k == r;
#pragma unroll
      for (int j = i; j < m; j++) {
        const double ai = A[i * m + j], ar = A[r * m + j];
        A[i * m + j] = sw ? ar : ai;
        A[r * m + j] = sw ? ai : ar;
      }
      const double bi = b[i], br = b[r];
      b[i] = sw ? br : bi;
      b[r] = sw ? bi : br;
    }
    const double d = -1 / A[i * m + i];
#pragma unroll
    for (int j = i + 1; j < m; j++) {
      const double alpha = A[j * m + i] * d;
#pragma unroll
      for (int kk = i + 1; kk < m; kk++) A[j * m + kk] += alpha * A[i * m + kk];
      b[j] += alpha * b[i];
    }
  }
#pragma unroll
  for (int i = m - 1; i >= 0; i--) {
    double s = b[i];
#pragma unroll
    for (int k = i + 1; k < m; k++) s -= A[i * m + k] * b[k];
    b[i] = s / A[i * m + i];
  }
  return true;
}

// get_affine_transform(center, scale, rot=0, (out,out)) -> inverse map M (dst -> src) as warpAffine forms it
__device__ void inverse_affine(double cx, double cy, double scale, int out, double* M) {
  float src[3][2], dst[3][2];
  const double src_dir1 = 0.0 * 0.0 + (scale * -0.5) * 1.0;
  const double src_dir0 = 0.0 * 1.0 - (scale * -0.5) * 0.0;
  src[0][0] = (float)(cx + scale * 0.0); src[0][1] = (float)(cy + scale * 0.0);
  src[1][0] = (float)(cx + src_dir0 + scale * 0.0); src[1][1] = (float)(cy + src_dir1 + scale * 0.0);
  dst[0][0] = (float)(out * 0.5); dst[0][1] = (float)(out * 0.5);
  dst[1][0] = (float)(out * 0.5) + 0.f; dst[1][1] = (float)(out * 0.5) + (float)(out * -0.5);
  float dx = src[0][0] - src[1][0], dy = src[0][1] - src[1][1];
  src[2][0] = src[1][0] + (-dy); src[2][1] = src[1][1] + dx;
  dx = dst[0][0] - dst[1][0]; dy = dst[0][1] - dst[1][1];
  dst[2][0] = dst[1][0] + (-dy); dst[2][1] = dst[1][1] + dx;
  double a[36], b[6];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const int j = i * 12, k = i * 12 + 6;
    a[j] = a[k + 3] = src[i][0];
    a[j + 1] = a[k + 4] = src[i][1];
    a[j + 2] = a[k + 5] = 1;
    a[j + 3] = a[j + 4] = a[j + 5] = 0;
    a[k] = a[k + 1] = a[k + 2] = 0;
    b[i * 2] = dst[i][0];
    b[i * 2 + 1] = dst[i][1];
  }
  if (!lu_solve6(a, b)) {
#pragma unroll
    for (int i = 0; i < 6; ++i) b[i] = 0.0;
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) M[i] = b[i];
  double D = M[0] * M[4] - M[1] * M[3];
  D = D != 0 ? 1. / D : 0;
  const double A11 = M[4] * D, A22 = M[0] * D;
  M[0] = A11; M[1] *= -D;
  M[3] *= -D; M[4] = A22;
  const double b1 = -M[0] * M[2] - M[1] * M[5];
  const double b2 = -M[3] * M[2] - M[4] * M[5];
  M[2] = b1; M[5] = b2;
}

__device__ __forceinline__ void src_coord(const double* M, int x, int y, bool nearest, int& sx, int& sy, int& alpha) {
  const int adelta = cv_round(M[0] * x * AB_SCALE), bdelta = cv_round(M[3] * x * AB_SCALE);
  const int round_delta = nearest ? AB_SCALE / 2 : AB_SCALE / INTER_TAB_SIZE / 2;
  const int X0 = cv_round((M[1] * y + M[2]) * AB_SCALE) + round_delta;
  const int Y0 = cv_round((M[4] * y + M[5]) * AB_SCALE) + round_delta;
  if (nearest) {
    sx = sat_short((X0 + adelta) >> AB_BITS);
    sy = sat_short((Y0 + bdelta) >> AB_BITS);
    alpha = 0;
  } else {
    const int X = (X0 + adelta) >> (AB_BITS - INTER_BITS), Y = (Y0 + bdelta) >> (AB_BITS - INTER_BITS);
    sx = sat_short(X >> INTER_BITS);
    sy = sat_short(Y >> INTER_BITS);
    alpha = (Y & (INTER_TAB_SIZE - 1)) * INTER_TAB_SIZE + (X & (INTER_TAB_SIZE - 1));
  }
}

struct Norm3 {
  double mean[3], stdv[3];
};

// roi_img (bilinear u8 -> normalised f32, CHW) + roi_depth (nearest f32)
__global__ __launch_bounds__(256) void crop_img_depth_kernel(const unsigned char* __restrict__ images,
                                                             const float* __restrict__ depths, int H, int W,
                                                             const int* __restrict__ im_idx,
                                                             const double* __restrict__ centers,
                                                             const double* __restrict__ scales,
                                                             float* __restrict__ roi_img, float* __restrict__ roi_depth,
                                                             int out, Norm3 nrm) {
  __shared__ double sM[6];
  const int bi = blockIdx.y;
  if (threadIdx.x == 0) inverse_affine(centers[2 * bi], centers[2 * bi + 1], scales[bi], out, sM);
  __syncthreads();
  double M[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) M[k] = sM[k];
  const size_t im = (size_t)(im_idx ? im_idx[bi] : 0);
  for (int it = 0; it < kPixPerThread; ++it) {
  const int p = (blockIdx.x * kPixPerThread + it) * blockDim.x + threadIdx.x;
  if (p >= out * out) break;
  const int y = p / out, x = p - y * out;
  if (roi_img) {
    const unsigned char* src = images + im * H * W * 3;
    int sx, sy, alpha;
    src_coord(M, x, y, false, sx, sy, alpha);
    const int fy = alpha >> INTER_BITS, fx = alpha & (INTER_TAB_SIZE - 1);
    int w0 = (32 - fy) * (32 - fx) * 32, w1 = (32 - fy) * fx * 32, w2 = fy * (32 - fx) * 32, w3 = fy * fx * 32;
    if (alpha == 0) { w0 = 32767; w3 = 1; }  // saturate_cast<short>(32768) + the table's sum fix-up
    const bool x0 = sx >= 0 && sx < W, x1 = sx + 1 >= 0 && sx + 1 < W, y0 = sy >= 0 && sy < H,
               y1 = sy + 1 >= 0 && sy + 1 < H;
    const unsigned char* r0 = src + ((size_t)sy * W + sx) * 3;
    const unsigned char* r1 = r0 + (size_t)W * 3;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int v00 = (x0 && y0) ? r0[k] : 0, v01 = (x1 && y0) ? r0[3 + k] : 0;
      const int v10 = (x0 && y1) ? r1[k] : 0, v11 = (x1 && y1) ? r1[3 + k] : 0;
      int s = (v00 * w0 + v01 * w1 + v10 * w2 + v11 * w3 + (1 << 14)) >> 15;
      s = s < 0 ? 0 : (s > 255 ? 255 : s);
      roi_img[(((size_t)bi * 3 + k) * out + y) * out + x] = (float)(((double)s - nrm.mean[k]) / nrm.stdv[k]);
    }
  }
  if (roi_depth && depths) {
    const float* src = depths + im * H * W;
    int sx, sy, alpha;
    src_coord(M, x, y, true, sx, sy, alpha);
    roi_depth[((size_t)bi * out + y) * out + x] = (sx >= 0 && sx < W && sy >= 0 && sy < H) ? src[(size_t)sy * W + sx] : 0.f;
  }
  }
}

// one output pixel of roi_coord_2d: bilinear (float weights) of the analytic coord_2d map, CHW output
__device__ __forceinline__ void coord2d_pixel(const double* M, int H, int W, int p, int out, int bi,
                                              float* __restrict__ roi_coord2d) {
  const int y = p / out, x = p - y * out;
  int sx, sy, alpha;
  src_coord(M, x, y, false, sx, sy, alpha);
  const int fy = alpha >> INTER_BITS, fx = alpha & (INTER_TAB_SIZE - 1);
  const float sc = 1.f / INTER_TAB_SIZE;
  const float vy0 = 1.f - fy * sc, vy1 = fy * sc, vx0 = 1.f - fx * sc, vx1 = fx * sc;
  const float w0 = vy0 * vx0, w1 = vy0 * vx1, w2 = vy1 * vx0, w3 = vy1 * vx1;
  const bool x0 = sx >= 0 && sx < W, x1 = sx + 1 >= 0 && sx + 1 < W, y0 = sy >= 0 && sy < H, y1 = sy + 1 >= 0 && sy + 1 < H;
  // np.linspace(0, 1, n, endpoint=False, dtype=float32): float32(i * (1.0/n) + 0.0)
  const double stepx = 1.0 / (double)W, stepy = 1.0 / (double)H;
  const float cx0 = (float)((double)sx * stepx + 0.0), cx1 = (float)((double)(sx + 1) * stepx + 0.0);
  const float cy0 = (float)((double)sy * stepy + 0.0), cy1 = (float)((double)(sy + 1) * stepy + 0.0);
  // channel 0 = x coordinate, channel 1 = y coordinate (meshgrid(x, y), HWC)
  const float a00 = (x0 && y0) ? cx0 : 0.f, a01 = (x1 && y0) ? cx1 : 0.f, a10 = (x0 && y1) ? cx0 : 0.f,
              a11 = (x1 && y1) ? cx1 : 0.f;
  const float b00 = (x0 && y0) ? cy0 : 0.f, b01 = (x1 && y0) ? cy0 : 0.f, b10 = (x0 && y1) ? cy1 : 0.f,
              b11 = (x1 && y1) ? cy1 : 0.f;
  roi_coord2d[(((size_t)bi * 2 + 0) * out + y) * out + x] = ((a00 * w0 + a01 * w1) + a10 * w2) + a11 * w3;
  roi_coord2d[(((size_t)bi * 2 + 1) * out + y) * out + x] = ((b00 * w0 + b01 * w1) + b10 * w2) + b11 * w3;
}

// The same two maps for out == 256 (the configured INPUT_RES): thread = output column, workgroup = 16 output rows, so
// the column terms of the fixed-point source coordinate are per-thread constants and the row terms are uniform; the
// float64 normalisation ((s - mean) / std -> float32) has only 256 possible inputs per channel and comes from a table in
// LDS (768 double divisions per workgroup instead of 12 288); the 2 x 2 x 3 source bytes of an interior pixel are two
// unaligned 8-byte loads instead of twelve byte loads.  Same integer arithmetic and the same table values as the generic
// kernel above: bit-identical output.
typedef unsigned long long __attribute__((aligned(1))) u64_unaligned;

__global__ __launch_bounds__(256) void crop_img_depth_256_kernel(const unsigned char* __restrict__ images,
                                                                 const float* __restrict__ depths, int H, int W,
                                                                 const int* __restrict__ im_idx,
                                                                 const double* __restrict__ centers,
                                                                 const double* __restrict__ scales,
                                                                 float* __restrict__ roi_img, float* __restrict__ roi_depth,
                                                                 float* __restrict__ roi_coord2d, int out_small, Norm3 nrm) {
  constexpr int out = 256;
  __shared__ double sM[6];
  __shared__ float lut[3][256];
  const int bi = blockIdx.y, x = threadIdx.x;
  if (blockIdx.x >= out / kPixPerThread) {  // the extra workgroup of the ROI: its roi_coord_2d map (own affine map)
    if (threadIdx.x == 0) inverse_affine(centers[2 * bi], centers[2 * bi + 1], scales[bi], out_small, sM);
    __syncthreads();
    double Ms[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) Ms[k] = sM[k];
    for (int p = threadIdx.x; p < out_small * out_small; p += 256) coord2d_pixel(Ms, H, W, p, out_small, bi, roi_coord2d);
    return;
  }
  if (threadIdx.x == 0) inverse_affine(centers[2 * bi], centers[2 * bi + 1], scales[bi], out, sM);
  if (roi_img) {
#pragma unroll
    for (int k = 0; k < 3; ++k) lut[k][x] = (float)(((double)x - nrm.mean[k]) / nrm.stdv[k]);
  }
  __syncthreads();
  double M[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) M[k] = sM[k];
  const size_t im = (size_t)(im_idx ? im_idx[bi] : 0);
  const int adelta = cv_round(M[0] * x * AB_SCALE), bdelta = cv_round(M[3] * x * AB_SCALE);
  const unsigned char* const src = images + im * H * W * 3;
  const float* const dsrc = depths ? depths + im * H * W : nullptr;
  const bool want_depth = roi_depth && depths;
#pragma unroll 4
  for (int it = 0; it < kPixPerThread; ++it) {
    const int y = blockIdx.x * kPixPerThread + it;
    const int X00 = cv_round((M[1] * y + M[2]) * AB_SCALE), Y00 = cv_round((M[4] * y + M[5]) * AB_SCALE);
    if (roi_img) {
      const int X0 = X00 + AB_SCALE / INTER_TAB_SIZE / 2, Y0 = Y00 + AB_SCALE / INTER_TAB_SIZE / 2;
      const int X = (X0 + adelta) >> (AB_BITS - INTER_BITS), Y = (Y0 + bdelta) >> (AB_BITS - INTER_BITS);
      const int sx = sat_short(X >> INTER_BITS), sy = sat_short(Y >> INTER_BITS);
      const int fy = Y & (INTER_TAB_SIZE - 1), fx = X & (INTER_TAB_SIZE - 1);
      int w0 = (32 - fy) * (32 - fx) * 32, w1 = (32 - fy) * fx * 32, w2 = fy * (32 - fx) * 32, w3 = fy * fx * 32;
      if ((fy | fx) == 0) { w0 = 32767; w3 = 1; }  // saturate_cast<short>(32768) + the table's sum fix-up
      int v00[3], v01[3], v10[3], v11[3];
      if (sx >= 0 && sx + 2 < W && sy >= 0 && sy + 1 < H) {  // interior: 8 bytes per source row hold both pixels
        const unsigned char* r0 = src + ((size_t)sy * W + sx) * 3;
        const unsigned long long a = *reinterpret_cast<const u64_unaligned*>(r0);
        const unsigned long long b = *reinterpret_cast<const u64_unaligned*>(r0 + (size_t)W * 3);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          v00[k] = (int)((a >> (8 * k)) & 0xff); v01[k] = (int)((a >> (8 * (k + 3))) & 0xff);
          v10[k] = (int)((b >> (8 * k)) & 0xff); v11[k] = (int)((b >> (8 * (k + 3))) & 0xff);
        }
      } else {
        const bool x0 = sx >= 0 && sx < W, x1 = sx + 1 >= 0 && sx + 1 < W, y0 = sy >= 0 && sy < H,
                   y1 = sy + 1 >= 0 && sy + 1 < H;
        const long base = ((long)sy * W + sx) * 3;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          v00[k] = (x0 && y0) ? src[base + k] : 0; v01[k] = (x1 && y0) ? src[base + 3 + k] : 0;
          v10[k] = (x0 && y1) ? src[base + (long)W * 3 + k] : 0; v11[k] = (x1 && y1) ? src[base + (long)W * 3 + 3 + k] : 0;
        }
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        int sv = (v00[k] * w0 + v01[k] * w1 + v10[k] * w2 + v11[k] * w3 + (1 << 14)) >> 15;
        sv = sv < 0 ? 0 : (sv > 255 ? 255 : sv);
        __builtin_nontemporal_store(lut[k][sv], roi_img + (((size_t)bi * 3 + k) * out + y) * out + x);
      }
    }
    if (want_depth) {
      const int sx = sat_short((X00 + AB_SCALE / 2 + adelta) >> AB_BITS), sy = sat_short((Y00 + AB_SCALE / 2 + bdelta) >> AB_BITS);
      __builtin_nontemporal_store((sx >= 0 && sx < W && sy >= 0 && sy < H) ? dsrc[(size_t)sy * W + sx] : 0.f,
                                  roi_depth + ((size_t)bi * out + y) * out + x);
    }
  }
}

// roi_coord_2d: bilinear (float weights) of the analytic coord_2d map, CHW output
__global__ __launch_bounds__(256) void crop_coord2d_kernel(int H, int W, const double* __restrict__ centers,
                                                           const double* __restrict__ scales,
                                                           float* __restrict__ roi_coord2d, int out) {
  __shared__ double sM[6];
  const int bi = blockIdx.y;
  if (threadIdx.x == 0) inverse_affine(centers[2 * bi], centers[2 * bi + 1], scales[bi], out, sM);
  __syncthreads();
  double M[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) M[k] = sM[k];
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= out * out) return;
  coord2d_pixel(M, H, W, p, out, bi, roi_coord2d);
}

}  // namespace

extern "C" {

int gdrnpp_crop_resize_roi(const unsigned char* images, const float* depths, int n_im, int H, int W,
                           const int* im_idx, const double* centers, const double* scales, float* roi_img,
                           float* roi_depth, float* roi_coord2d, int b, int out_res, int out_res_small,
                           const double* h_mean3, const double* h_std3, void* stream) {
  if (b == 0) return 0;  // an image / a rank without detections: nothing to crop
  GDRNPP_REQUIRE(centers && scales && b > 0 && H > 0 && W > 0 && n_im > 0, GDRNPP_EINVAL,
                 "gdrnpp_crop_resize_roi: bad arguments b=%d H=%d W=%d n_im=%d", b, H, W, n_im);
  GDRNPP_REQUIRE(b <= 65535, GDRNPP_ELIMIT, "gdrnpp_crop_resize_roi: b=%d > 65535", b);
  GDRNPP_REQUIRE(!roi_img || (images && h_mean3 && h_std3 && out_res > 0), GDRNPP_EINVAL,
                 "gdrnpp_crop_resize_roi: roi_img requested without images / mean / std");
  GDRNPP_REQUIRE(W < 32767 && H < 32767, GDRNPP_ELIMIT, "gdrnpp_crop_resize_roi: image larger than SHRT_MAX");
  hipStream_t st = (hipStream_t)stream;
  bool coord_done = false;
  if ((roi_img || (roi_depth && depths)) && out_res > 0) {
    Norm3 nrm;
    for (int k = 0; k < 3; ++k) { nrm.mean[k] = h_mean3 ? h_mean3[k] : 0.0; nrm.stdv[k] = h_std3 ? h_std3[k] : 1.0; }
    dim3 grid((out_res * out_res + 256 * kPixPerThread - 1) / (256 * kPixPerThread), b);
    if (out_res == 256) {
      const bool with_coord = roi_coord2d && out_res_small > 0;   // one more workgroup per ROI writes roi_coord_2d
      if (with_coord) { grid.x += 1; coord_done = true; }
      hipLaunchKernelGGL(crop_img_depth_256_kernel, grid, dim3(256), 0, st, images, depths, H, W, im_idx, centers, scales,
                         roi_img, roi_depth, with_coord ? roi_coord2d : nullptr, out_res_small, nrm);
    } else {
      hipLaunchKernelGGL(crop_img_depth_kernel, grid, dim3(256), 0, st, images, depths, H, W, im_idx, centers, scales,
                         roi_img, roi_depth, out_res, nrm);
    }
  }
  if (roi_coord2d && out_res_small > 0 && !coord_done) {
    dim3 grid((out_res_small * out_res_small + 255) / 256, b);
    hipLaunchKernelGGL(crop_coord2d_kernel, grid, dim3(256), 0, st, H, W, centers, scales, roi_coord2d, out_res_small);
  }
  return gdrnpp::check_launch("gdrnpp_crop_resize_roi");
}

}  // extern "C"
