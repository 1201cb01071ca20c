// ROIAlign crop-resize on gfx950 — SURVEY.md §8 row a1b (the "roi_align" flavour of crop-resize).
//
// Behavioural spec: the reference calls detectron2's ROIAlign(output_size, spatial_scale=1.0, sampling_ratio=0,
// aligned=True) from crop_resize_by_d2_roialign (core/utils/data_utils.py:65-112) and batch_crop_resize
// (core/utils/zoom_utils.py:80-96).  detectron2 is a third-party dependency outside the tree; its published
// forward algorithm (detectron2/layers/csrc/ROIAlignRotated-less ROIAlign, identical to torchvision.ops.roi_align)
// is restated: half-pixel shift when aligned, bin grid = ceil(roi_size / pooled_size) when sampling_ratio == 0,
// bilinear_interpolate with the (-1, size) validity window and clamping at the far border, mean over the grid.
// fp32, accumulation order iy-major / ix-minor, val = w1*v1 + w2*v2 + w3*v3 + w4*v4 left to right, no FMA.
//
// pw fastest (coalesced stores), grid (plane tiles, channel, ROI); taps are gathers inside one [H,W] channel plane
// (L2-resident for image-sized inputs).  Write-bound: 4 B per output element.
#include "common.hpp"
#include <cfloat>

namespace {

struct __attribute__((aligned(4))) float2_u { float x, y; };

// One axis of detectron2's bilinear_interpolate: validity window (-1, size), clamp at 0, far-border clamp.
struct AxisTap {
  bool valid;
  int low, high;
  float l, h;  // weight of `high`, weight of `low`
};
__device__ __forceinline__ AxisTap axis_tap(float v, int size) {
  AxisTap t;
  t.valid = !(v < -1.0f || v > (float)size);
  if (v <= 0) v = 0;
  t.low = (int)v;
  if (t.low >= size - 1) { t.high = t.low = size - 1; v = (float)t.low; } else { t.high = t.low + 1; }
  if (!t.valid) t.low = t.high = 0;  // any in-range address: the value is discarded
  t.l = v - t.low;
  t.h = 1.f - t.l;
  return t;
}

// grid (tiles of row groups x PW, channel, ROI).  A thread owns kRows consecutive output rows of one column: the x taps of
// a sample column are shared by its rows, and the 2 * kRows row loads of a sample are issued together (a workgroup lives
// for a few dependent memory round trips, so the kernel's time is rounds x latency: more loads in flight per thread and
// kRows times fewer workgroups is what it needs).  Per output the samples are still added iy-major, ix-minor, each as
// w1*v1 + w2*v2 + w3*v3 + w4*v4 left to right.
constexpr int kRows = 8;

template <bool kAligned>
__global__ __launch_bounds__(256) void roi_align_kernel(const float* __restrict__ x, const float* __restrict__ rois,
                                                        float* __restrict__ out, int C, int H, int W, int PH, int PW,
                                                        float spatial_scale, int sampling_ratio) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  const int groups = (PH + kRows - 1) / kRows;
  if (p >= groups * PW) return;
  const int g = p / PW, pw = p - g * PW, ph0 = g * kRows;
  const int c = blockIdx.y, n = blockIdx.z;
  const float* r = rois + 5 * (size_t)n;
  const int bi = (int)r[0];
  const float offset = kAligned ? 0.5f : 0.f;
  const float sw = r[1] * spatial_scale - offset, sh = r[2] * spatial_scale - offset;
  const float ew = r[3] * spatial_scale - offset, eh = r[4] * spatial_scale - offset;
  float rw = ew - sw, rh = eh - sh;
  if (!kAligned) { rw = fmaxf(rw, 1.f); rh = fmaxf(rh, 1.f); }
  const float bin_h = rh / (float)PH, bin_w = rw / (float)PW;
  const int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)PH);
  const int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)PW);
  const float count = fmaxf((float)(gh * gw), 1.f);
  const float* data = x + ((size_t)bi * C + c) * H * W;
  const float x0 = sw + pw * bin_w;
  float acc[kRows];
#pragma unroll
  for (int k = 0; k < kRows; ++k) acc[k] = 0.f;
  for (int iy = 0; iy < gh; ++iy) {
    const float dy = (iy + .5f) * bin_h / (float)gh;
    AxisTap ty[kRows];
#pragma unroll
    for (int k = 0; k < kRows; ++k) ty[k] = axis_tap(sh + (ph0 + k) * bin_h + dy, H);
    for (int ix = 0; ix < gw; ++ix) {
      const AxisTap tx = axis_tap(x0 + (ix + .5f) * bin_w / (float)gw, W);
      // high is low + 1 or (clamped at the right border) low: both taps of a row come from one 8-byte load at
      // min(low, W - 2) (4-byte alignment is enough for global_load_dwordx2); W == 1 has a single column
      const int xb = W >= 2 ? min(tx.low, W - 2) : 0;
      float2_u r0[kRows], r1[kRows];
#pragma unroll
      for (int k = 0; k < kRows; ++k) {
        if (W >= 2) {
          r0[k] = *reinterpret_cast<const float2_u*>(data + ty[k].low * W + xb);
          r1[k] = *reinterpret_cast<const float2_u*>(data + ty[k].high * W + xb);
        } else {
          r0[k].x = r0[k].y = data[ty[k].low * W];
          r1[k].x = r1[k].y = data[ty[k].high * W];
        }
      }
#pragma unroll
      for (int k = 0; k < kRows; ++k) {
        const float v1 = tx.low == xb ? r0[k].x : r0[k].y, v2 = tx.high == xb ? r0[k].x : r0[k].y;
        const float v3 = tx.low == xb ? r1[k].x : r1[k].y, v4 = tx.high == xb ? r1[k].x : r1[k].y;
        const float w1 = ty[k].h * tx.h, w2 = ty[k].h * tx.l, w3 = ty[k].l * tx.h, w4 = ty[k].l * tx.l;
        const float val = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
        acc[k] += (ty[k].valid && tx.valid) ? val : 0.f;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < kRows; ++k)
    if (ph0 + k < PH) __builtin_nontemporal_store(acc[k] / count, out + (((size_t)n * C + c) * PH + ph0 + k) * PW + pw);
}

// torchvision.ops.RoIPool forward (the "nearest" flavour of batch_crop_resize, core/utils/zoom_utils.py:92-93): ROI corners
// rounded half away from zero to pixels, width / height = max(end - start + 1, 1), bin [floor(p * bin), ceil((p + 1) * bin))
// shifted by the ROI start and clipped to the image; output = max over the bin, 0 for an empty bin.  One thread per output
// element, the bin's row segments are contiguous reads; a max is order-independent, so results equal the CPU form bit for bit.
__global__ __launch_bounds__(256) void roi_pool_kernel(const float* __restrict__ x, const float* __restrict__ rois,
                                                       float* __restrict__ out, int C, int H, int W, int PH, int PW,
                                                       float spatial_scale, long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int pw = (int)(i % PW), ph = (int)((i / PW) % PH), c = (int)((i / ((long)PW * PH)) % C);
    const int n = (int)(i / ((long)PW * PH * C));
    const float* r = rois + 5 * (size_t)n;
    const int bi = (int)r[0];
    const int sw = (int)roundf(r[1] * spatial_scale), sh = (int)roundf(r[2] * spatial_scale);
    const int ew = (int)roundf(r[3] * spatial_scale), eh = (int)roundf(r[4] * spatial_scale);
    const int rw = max(ew - sw + 1, 1), rh = max(eh - sh + 1, 1);
    const float bh = (float)rh / (float)PH, bw = (float)rw / (float)PW;
    int h0 = (int)floorf((float)ph * bh), w0 = (int)floorf((float)pw * bw);
    int h1 = (int)ceilf((float)(ph + 1) * bh), w1 = (int)ceilf((float)(pw + 1) * bw);
    h0 = min(max(h0 + sh, 0), H); h1 = min(max(h1 + sh, 0), H);
    w0 = min(max(w0 + sw, 0), W); w1 = min(max(w1 + sw, 0), W);
    const bool empty = h1 <= h0 || w1 <= w0;
    float m = empty ? 0.f : -FLT_MAX;
    const float* img = x + ((size_t)bi * C + c) * H * W;
    for (int hh = h0; hh < h1; ++hh)
      for (int ww = w0; ww < w1; ++ww) {
        const float v = img[(size_t)hh * W + ww];
        if (v > m) m = v;
      }
    out[i] = m;
  }
}

}  // namespace

extern "C" int gdrnpp_roi_pool(const float* x, const float* rois, float* out, int n_rois, int C, int H, int W, int pooled_h,
                               int pooled_w, float spatial_scale, void* stream) {
  if (n_rois == 0) return 0;
  GDRNPP_REQUIRE(x && rois && out, GDRNPP_EINVAL, "gdrnpp_roi_pool: null pointer");
  GDRNPP_REQUIRE(n_rois > 0 && C > 0 && H > 0 && W > 0 && pooled_h > 0 && pooled_w > 0, GDRNPP_EINVAL, "gdrnpp_roi_pool: bad sizes");
  const long total = (long)n_rois * C * pooled_h * pooled_w;
  const long blocks = (total + 255) / 256;
  hipLaunchKernelGGL(roi_pool_kernel, dim3((unsigned)(blocks < 65536 * 4 ? blocks : 65536 * 4)), dim3(256), 0, (hipStream_t)stream,
                     x, rois, out, C, H, W, pooled_h, pooled_w, spatial_scale, total);
  return gdrnpp::check_launch("gdrnpp_roi_pool");
}

extern "C" int gdrnpp_roi_align(const float* x, const float* rois, float* out, int n_rois, int C, int H, int W,
                                int pooled_h, int pooled_w, float spatial_scale, int sampling_ratio, int aligned,
                                void* stream) {
  if (n_rois == 0) return 0;
  GDRNPP_REQUIRE(x && rois && out, GDRNPP_EINVAL, "gdrnpp_roi_align: null pointer");
  GDRNPP_REQUIRE(n_rois > 0 && C > 0 && H > 0 && W > 0 && pooled_h > 0 && pooled_w > 0 && sampling_ratio >= 0,
                 GDRNPP_EINVAL, "gdrnpp_roi_align: bad sizes");
  GDRNPP_REQUIRE(n_rois <= 65535 && C <= 65535 && (long)pooled_h * pooled_w < (1l << 30), GDRNPP_ELIMIT,
                 "gdrnpp_roi_align: n_rois=%d / C=%d above 65535 or output plane too large", n_rois, C);
  const dim3 grid((unsigned)((((pooled_h + kRows - 1) / kRows) * pooled_w + 255) / 256), (unsigned)C, (unsigned)n_rois);
  if (aligned)
    hipLaunchKernelGGL(roi_align_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, x, rois, out, C, H, W, pooled_h,
                       pooled_w, spatial_scale, sampling_ratio);
  else
    hipLaunchKernelGGL(roi_align_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, x, rois, out, C, H, W, pooled_h,
                       pooled_w, spatial_scale, sampling_ratio);
  return gdrnpp::check_launch("gdrnpp_roi_align");
}
