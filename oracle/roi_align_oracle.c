/*
 * ORACLE — test infrastructure only.
 * CPU restatement of detectron2's ROIAlign forward as the reference calls it
 * (ROIAlign(output_size, 1.0, 0, aligned=True): core/utils/data_utils.py:88, core/utils/zoom_utils.py:92).
 * detectron2 (third-party, source install, unpinned) is not in the tree; this follows its published
 * algorithm (detectron2/layers/csrc/ROIAlign: roi_align_forward + bilinear_interpolate, the same code as
 * torchvision.ops.roi_align).  fp32, left-to-right, no FMA (build with -ffp-contract=off).
 * PARITY UNPINNED by the reference (no tests, detectron2/torchvision not importable here).
 */
#include <math.h>

static float bilinear_interpolate(const float* data, int height, int width, float y, float x) {
  if (y < -1.0f || y > (float)height || x < -1.0f || x > (float)width) return 0.f;
  if (y <= 0) y = 0;
  if (x <= 0) x = 0;
  int y_low = (int)y, x_low = (int)x, y_high, x_high;
  if (y_low >= height - 1) { y_high = y_low = height - 1; y = (float)y_low; } else { y_high = y_low + 1; }
  if (x_low >= width - 1) { x_high = x_low = width - 1; x = (float)x_low; } else { x_high = x_low + 1; }
  float ly = y - y_low, lx = x - x_low, hy = 1.f - ly, hx = 1.f - lx;
  float v1 = data[y_low * width + x_low], v2 = data[y_low * width + x_high];
  float v3 = data[y_high * width + x_low], v4 = data[y_high * width + x_high];
  float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
  return w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
}

void oracle_roi_align(const float* x, const float* rois, float* out, int n_rois, int C, int H, int W, int PH, int PW,
                      float spatial_scale, int sampling_ratio, int aligned) {
  for (int n = 0; n < n_rois; ++n) {
    const float* r = rois + 5 * n;
    int bi = (int)r[0];
    float offset = aligned ? 0.5f : 0.f;
    float sw = r[1] * spatial_scale - offset, sh = r[2] * spatial_scale - offset;
    float ew = r[3] * spatial_scale - offset, eh = r[4] * spatial_scale - offset;
    float rw = ew - sw, rh = eh - sh;
    if (!aligned) { rw = fmaxf(rw, 1.f); rh = fmaxf(rh, 1.f); }
    float bin_h = rh / (float)PH, bin_w = rw / (float)PW;
    int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)PH);
    int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)PW);
    float count = fmaxf((float)(gh * gw), 1.f);
    for (int c = 0; c < C; ++c) {
      const float* data = x + ((long)bi * C + c) * H * W;
      for (int ph = 0; ph < PH; ++ph)
        for (int pw = 0; pw < PW; ++pw) {
          float acc = 0.f;
          for (int iy = 0; iy < gh; ++iy) {
            float y = sh + ph * bin_h + (iy + .5f) * bin_h / (float)gh;
            for (int ix = 0; ix < gw; ++ix) {
              float xx = sw + pw * bin_w + (ix + .5f) * bin_w / (float)gw;
              acc += bilinear_interpolate(data, H, W, y, xx);
            }
          }
          out[(((long)n * C + c) * PH + ph) * PW + pw] = acc / count;
        }
    }
  }
}

/* torchvision.ops.RoIPool forward (core/utils/zoom_utils.py:92-93, interpolation = "nearest") restated from its published
 * CPU kernel (torchvision/csrc/ops/cpu/roi_pool_kernel.cpp): corners rounded with round() (half away from zero), width /
 * height = max(end - start + 1, 1), bin = [floor(p * bin_size), ceil((p + 1) * bin_size)) + start, clipped to the image;
 * max over the bin, 0 when empty.  PARITY UNPINNED (torchvision not importable here). */
void oracle_roi_pool(const float* x, const float* rois, float* out, int n_rois, int C, int H, int W, int PH, int PW,
                     float spatial_scale) {
  for (int n = 0; n < n_rois; ++n) {
    const float* r = rois + 5 * n;
    int bi = (int)r[0];
    int sw = (int)roundf(r[1] * spatial_scale), sh = (int)roundf(r[2] * spatial_scale);
    int ew = (int)roundf(r[3] * spatial_scale), eh = (int)roundf(r[4] * spatial_scale);
    int rw = ew - sw + 1 > 1 ? ew - sw + 1 : 1, rh = eh - sh + 1 > 1 ? eh - sh + 1 : 1;
    float bh = (float)rh / (float)PH, bw = (float)rw / (float)PW;
    for (int ph = 0; ph < PH; ++ph)
      for (int pw = 0; pw < PW; ++pw) {
        int h0 = (int)floorf((float)ph * bh), w0 = (int)floorf((float)pw * bw);
        int h1 = (int)ceilf((float)(ph + 1) * bh), w1 = (int)ceilf((float)(pw + 1) * bw);
        h0 += sh; h1 += sh; w0 += sw; w1 += sw;
        h0 = h0 < 0 ? 0 : (h0 > H ? H : h0); h1 = h1 < 0 ? 0 : (h1 > H ? H : h1);
        w0 = w0 < 0 ? 0 : (w0 > W ? W : w0); w1 = w1 < 0 ? 0 : (w1 > W ? W : w1);
        int empty = h1 <= h0 || w1 <= w0;
        for (int c = 0; c < C; ++c) {
          const float* img = x + ((long)bi * C + c) * H * W;
          float m = empty ? 0.f : -3.402823466e+38f;
          for (int hh = h0; hh < h1; ++hh)
            for (int ww = w0; ww < w1; ++ww)
              if (img[hh * W + ww] > m) m = img[hh * W + ww];
          out[(((long)n * C + c) * PH + ph) * PW + pw] = m;
        }
      }
  }
}
