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
// One thread per output element, pw fastest (coalesced stores); taps are gathers inside one [H,W] channel plane
// (L2-resident for image-sized inputs).  Write-bound: 4 B per output element.
#include "common.hpp"

namespace {

__device__ __forceinline__ float bilinear_interpolate(const float* __restrict__ data, int height, int width, float y,
                                                      float x) {
  if (y < -1.0f || y > (float)height || x < -1.0f || x > (float)width) return 0.f;
  if (y <= 0) y = 0;
  if (x <= 0) x = 0;
  int y_low = (int)y, x_low = (int)x, y_high, x_high;
  if (y_low >= height - 1) { y_high = y_low = height - 1; y = (float)y_low; } else { y_high = y_low + 1; }
  if (x_low >= width - 1) { x_high = x_low = width - 1; x = (float)x_low; } else { x_high = x_low + 1; }
  const float ly = y - y_low, lx = x - x_low, hy = 1.f - ly, hx = 1.f - lx;
  const float v1 = data[y_low * width + x_low], v2 = data[y_low * width + x_high];
  const float v3 = data[y_high * width + x_low], v4 = data[y_high * width + x_high];
  const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
  return w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
}

__global__ void roi_align_kernel(const float* __restrict__ x, const float* __restrict__ rois, float* __restrict__ out,
                                 long total, int C, int H, int W, int PH, int PW, float spatial_scale,
                                 int sampling_ratio, int aligned) {
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int pw = (int)(idx % PW);
    const int ph = (int)((idx / PW) % PH);
    const int c = (int)((idx / ((long)PW * PH)) % C);
    const int n = (int)(idx / ((long)PW * PH * C));
    const float* r = rois + 5 * (size_t)n;
    const int bi = (int)r[0];
    const float offset = aligned ? 0.5f : 0.f;
    const float sw = r[1] * spatial_scale - offset, sh = r[2] * spatial_scale - offset;
    const float ew = r[3] * spatial_scale - offset, eh = r[4] * spatial_scale - offset;
    float rw = ew - sw, rh = eh - sh;
    if (!aligned) { rw = fmaxf(rw, 1.f); rh = fmaxf(rh, 1.f); }
    const float bin_h = rh / (float)PH, bin_w = rw / (float)PW;
    const int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)PH);
    const int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)PW);
    const float count = fmaxf((float)(gh * gw), 1.f);
    const float* data = x + ((size_t)bi * C + c) * H * W;
    float acc = 0.f;
    for (int iy = 0; iy < gh; ++iy) {
      const float y = sh + ph * bin_h + (iy + .5f) * bin_h / (float)gh;
      for (int ix = 0; ix < gw; ++ix) {
        const float xx = sw + pw * bin_w + (ix + .5f) * bin_w / (float)gw;
        acc += bilinear_interpolate(data, H, W, y, xx);
      }
    }
    out[idx] = acc / count;
  }
}

}  // namespace

extern "C" int gdrnpp_roi_align(const float* x, const float* rois, float* out, int n_rois, int C, int H, int W,
                                int pooled_h, int pooled_w, float spatial_scale, int sampling_ratio, int aligned,
                                void* stream) {
  GDRNPP_REQUIRE(x && rois && out, GDRNPP_EINVAL, "gdrnpp_roi_align: null pointer");
  GDRNPP_REQUIRE(n_rois > 0 && C > 0 && H > 0 && W > 0 && pooled_h > 0 && pooled_w > 0 && sampling_ratio >= 0,
                 GDRNPP_EINVAL, "gdrnpp_roi_align: bad sizes");
  const long total = (long)n_rois * C * pooled_h * pooled_w;
  long blocks = (total + 255) / 256;
  if (blocks > 256 * 64) blocks = 256 * 64;
  hipLaunchKernelGGL(roi_align_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, rois, out, total, C,
                     H, W, pooled_h, pooled_w, spatial_scale, sampling_ratio, aligned);
  return gdrnpp::check_launch("gdrnpp_roi_align");
}
