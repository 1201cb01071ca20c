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
// Write-bound: 4 B per output element; the source pixels under a ROI are read through L2.
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

// Round 6 (profiles/r06_roi_align.md).  grid (row groups x column tiles, channel chunks, ROI); a workgroup = (256 / cols) row lanes x
// cols output columns (cols = 64 / 128 / 256, so narrow outputs keep every lane busy and a wave never straddles two rows); a thread
// = ONE OUTPUT COLUMN for ROWS consecutive output rows and up to 4 channel planes:
//   * everything that depends on the column only — the x taps of the column's sample columns (kept in registers for up to 4
//     per bin), their validity, which half of the 8-byte row load is the low / high tap — is computed ONCE per thread;
//   * everything that depends on the row only — the y taps of a sample row — is uniform per wave: the scalar unit does it;
//   * the four bilinear weights of a sample are formed once and applied to the channel planes;
//   * a wave's stores are 256-byte row segments of each plane.
// What bounds it (rocprofv3 counters, 128 ROIs -> 3 x 256 x 256): not HBM (50 MB fetched, 100 MB written in ~75 us) but the number
// of vector instructions per output — 36 M VALU + 2.4 M gather loads per launch, each 8-byte gather load split into ~16 L1
// accesses — with the waves parked on memory two thirds of the time.  Measured and NOT kept: 8 / 16 rows per thread (fewer
// waves: slower, 85 / 100 us), all 2 x ROWS x channels loads of a sample column issued before the first use (100 - 150 us: the
// registers cost the occupancy), the tile's source box staged in LDS and gathered from there (L1 accesses / 10, but 47 M VALU
// instructions for box, copy and clamps: 126 us), non-temporal stores (+3 %).  This form: 4 rows per thread, plain stores, 75 us
// = 0.17 of HBM against the round-5 kernel's 85 us (0.15).
// Per output the samples are still added iy-major, ix-minor, each as w1*v1 + w2*v2 + w3*v3 + w4*v4 left to right, then divided by
// the sample count (multiplied by its reciprocal when that is a power of two: the same bits): bit for bit roi_align_oracle.c.


struct RoiGeom {
  float sw, sh, bin_h, bin_w, count, rcount;
  int gh, gw;
  bool pow2;
};

// rows ph0 .. ph0 + kRows of output column pw, channels c0 .. c0 + nch.  GW = sample columns per bin when <= 4 (their taps live
// in registers for all rows), 0 = any number (recomputed per sample row).
template <int GW, int ROWS, bool NT, int CH, bool FULL>
__device__ __forceinline__ void roi_align_rows(const float* __restrict__ data, float* __restrict__ o, const RoiGeom& g, int nch,
                                               size_t plane, int H, int W, int PH, int PW, int ph0, int pw) {
  const float x0 = g.sw + pw * g.bin_w;
  // The y taps are uniform per wave, but gfx950's scalar unit has no float arithmetic: computed per sample row they cost every
  // wave ~35 vector instructions (an IEEE division among them) per (row, sample row) — 40 % of the kernel.  So the wave computes
  // all its ROWS x gh taps ONCE, one per lane, and the loops below fetch them with v_readlane (same expression: same bits).
  const int lane = (int)threadIdx.x & 63;
  const bool lane_taps = ROWS * g.gh <= 64;                            // uniform
  AxisTap tl = AxisTap{false, 0, 0, 0.f, 0.f};
  if (lane_taps) {
    const int k = lane / g.gh, iy = lane - k * g.gh;
    tl = axis_tap(g.sh + (ph0 + k) * g.bin_h + (iy + .5f) * g.bin_h / (float)g.gh, H);
  }
  AxisTap txs[GW > 0 ? GW : 1];
  int xbs[GW > 0 ? GW : 1];
  if (GW > 0) {
#pragma unroll
    for (int ix = 0; ix < GW; ++ix) {
      txs[ix] = axis_tap(x0 + (ix + .5f) * g.bin_w / (float)g.gw, W);
      xbs[ix] = min(txs[ix].low, W - 2);
    }
  }
#pragma unroll 2
  for (int k = 0; k < ROWS; ++k) {
    const int ph = ph0 + k;                                           // uniform
    if (ph >= PH) break;
    float acc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = 0.f;
    for (int iy = 0; iy < g.gh; ++iy) {
      AxisTap ty;
      if (lane_taps) {
        const int src_lane = k * g.gh + iy;                            // uniform
        ty.valid = __builtin_amdgcn_readlane((int)tl.valid, src_lane) != 0;
        ty.low = __builtin_amdgcn_readlane(tl.low, src_lane);
        ty.high = __builtin_amdgcn_readlane(tl.high, src_lane);
        ty.l = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tl.l), src_lane));
        ty.h = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tl.h), src_lane));
      } else {
        ty = axis_tap(g.sh + ph * g.bin_h + (iy + .5f) * g.bin_h / (float)g.gh, H);
      }
      const float* row_lo = data + (size_t)ty.low * W;
      const float* row_hi = data + (size_t)ty.high * W;
      auto sample = [&](const AxisTap& tx, int xb) {
        // high is low + 1 or (clamped at the right border) low: both taps of a row come from one 8-byte load at
        // min(low, W - 2) (4-byte alignment is enough for global_load_dwordx2); W == 1 has a single column
        const bool lo_first = tx.low == xb, hi_first = tx.high == xb;
        const float w1 = ty.h * tx.h, w2 = ty.h * tx.l, w3 = ty.l * tx.h, w4 = ty.l * tx.l;
        const bool ok = ty.valid && tx.valid;
        float2_u a[CH], b[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          if (FULL || c < nch) {
            a[c] = *reinterpret_cast<const float2_u*>(row_lo + c * plane + xb);      // (W >= 2: the launcher sends W == 1 elsewhere)
            b[c] = *reinterpret_cast<const float2_u*>(row_hi + c * plane + xb);
          }
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          if (FULL || c < nch) {
            const float v1 = lo_first ? a[c].x : a[c].y, v2 = hi_first ? a[c].x : a[c].y;
            const float v3 = lo_first ? b[c].x : b[c].y, v4 = hi_first ? b[c].x : b[c].y;
            const float val = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
            acc[c] += ok ? val : 0.f;
          }
        }
      };
      if (GW > 0) {
#pragma unroll
        for (int ix = 0; ix < GW; ++ix) sample(txs[ix], xbs[ix]);
      } else {
        for (int ix = 0; ix < g.gw; ++ix) {
          const AxisTap tx = axis_tap(x0 + (ix + .5f) * g.bin_w / (float)g.gw, W);
          sample(tx, min(tx.low, W - 2));
        }
      }
    }
#pragma unroll
    for (int c = 0; c < CH; ++c)
      if (FULL || c < nch) {
        const float v = g.pow2 ? acc[c] * g.rcount : acc[c] / g.count;
        if (NT) __builtin_nontemporal_store(v, o + ((size_t)c * PH + ph) * PW);
        else o[((size_t)c * PH + ph) * PW] = v;
      }
  }
}

template <bool kAligned, int ROWS, bool NT, int CH, bool FULL>
__global__ __launch_bounds__(256) void roi_align_kernel(const float* __restrict__ x, const float* __restrict__ rois,
                                                        float* __restrict__ out, int C, int H, int W, int PH, int PW,
                                                        float spatial_scale, int sampling_ratio, int col_tiles, int cols) {
  const int row_group = blockIdx.x / col_tiles, col_tile = blockIdx.x - row_group * col_tiles;
  // row lane (uniform per wave: cols % 64 == 0 — readfirstlane tells the compiler, so the y taps stay on the scalar unit), column
  const int ry = __builtin_amdgcn_readfirstlane((int)threadIdx.x / cols), cx = (int)threadIdx.x - ry * cols;
  const int pw = col_tile * cols + cx;
  const int ph0 = (row_group * (256 / cols) + ry) * ROWS;
  const int c0 = blockIdx.y * CH, n = blockIdx.z;
  const float* r = rois + 5 * (size_t)n;
  const int bi = (int)r[0];
  const float offset = kAligned ? 0.5f : 0.f;
  RoiGeom g;
  g.sw = r[1] * spatial_scale - offset;
  g.sh = r[2] * spatial_scale - offset;
  const float ew = r[3] * spatial_scale - offset, eh = r[4] * spatial_scale - offset;
  float rw = ew - g.sw, rh = eh - g.sh;
  if (!kAligned) { rw = fmaxf(rw, 1.f); rh = fmaxf(rh, 1.f); }
  g.bin_h = rh / (float)PH;
  g.bin_w = rw / (float)PW;
  g.gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)PH);
  g.gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)PW);
  g.count = fmaxf((float)(g.gh * g.gw), 1.f);
  const int icount = g.gh * g.gw;
  g.pow2 = icount >= 1 && (icount & (icount - 1)) == 0;     // x / 2^k == x * 2^-k exactly, also into the subnormals
  g.rcount = 1.f / g.count;
  if (pw >= PW || ph0 >= PH) return;
  const size_t plane = (size_t)H * W;
  const float* data = x + ((size_t)bi * C + c0) * plane;
  const int nch = FULL ? CH : min(CH, C - c0);              // uniform
  float* o = out + (((size_t)n * C + c0) * PH) * PW + pw;
  // the adaptive sampling grid (sampling_ratio = 0, the reference's setting) is per ROI, i.e. uniform in the workgroup
  switch (g.gw) {
    case 1: roi_align_rows<1, ROWS, NT, CH, FULL>(data, o, g, nch, plane, H, W, PH, PW, ph0, pw); break;
    case 2: roi_align_rows<2, ROWS, NT, CH, FULL>(data, o, g, nch, plane, H, W, PH, PW, ph0, pw); break;
    case 3: roi_align_rows<3, ROWS, NT, CH, FULL>(data, o, g, nch, plane, H, W, PH, PW, ph0, pw); break;
    case 4: roi_align_rows<4, ROWS, NT, CH, FULL>(data, o, g, nch, plane, H, W, PH, PW, ph0, pw); break;
    default: roi_align_rows<0, ROWS, NT, CH, FULL>(data, o, g, nch, plane, H, W, PH, PW, ph0, pw);
  }
}

// W == 1 (a feature map one pixel wide: no 8-byte row load exists): one thread per output element, four taps loaded one by one.
// Same sample order and arithmetic as above.
template <bool kAligned>
__global__ __launch_bounds__(256) void roi_align_generic_kernel(const float* __restrict__ x, const float* __restrict__ rois,
                                                                float* __restrict__ out, int C, int H, int W, int PH, int PW,
                                                                float spatial_scale, int sampling_ratio, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int pw = (int)(i % PW), ph = (int)((i / PW) % PH), c = (int)((i / ((long)PW * PH)) % C);
  const int n = (int)(i / ((long)PW * PH * C));
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
  float acc = 0.f;
  for (int iy = 0; iy < gh; ++iy) {
    const AxisTap ty = axis_tap(sh + ph * bin_h + (iy + .5f) * bin_h / (float)gh, H);
    for (int ix = 0; ix < gw; ++ix) {
      const AxisTap tx = axis_tap(sw + pw * bin_w + (ix + .5f) * bin_w / (float)gw, W);
      const float v1 = data[(size_t)ty.low * W + tx.low], v2 = data[(size_t)ty.low * W + tx.high];
      const float v3 = data[(size_t)ty.high * W + tx.low], v4 = data[(size_t)ty.high * W + tx.high];
      const float w1 = ty.h * tx.h, w2 = ty.h * tx.l, w3 = ty.l * tx.h, w4 = ty.l * tx.l;
      const float val = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
      acc += (ty.valid && tx.valid) ? val : 0.f;
    }
  }
  out[i] = acc / count;
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
  if (W == 1) {
    const long total = (long)n_rois * C * pooled_h * pooled_w;
    GDRNPP_REQUIRE((total + 255) / 256 < (1l << 31), GDRNPP_ELIMIT, "gdrnpp_roi_align: output too large");
    if (aligned)
      hipLaunchKernelGGL(roi_align_generic_kernel<true>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, rois, out, C, H, W,
                         pooled_h, pooled_w, spatial_scale, sampling_ratio, total);
    else
      hipLaunchKernelGGL(roi_align_generic_kernel<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, rois, out, C, H, W,
                         pooled_h, pooled_w, spatial_scale, sampling_ratio, total);
    return gdrnpp::check_launch("gdrnpp_roi_align");
  }
  const int cols = pooled_w > 128 ? 256 : (pooled_w > 64 ? 128 : 64);      // output columns per workgroup; 256 / cols row lanes
  const int col_tiles = (pooled_w + cols - 1) / cols;
  constexpr int rows = 4;                                     // output rows per thread (8 / 16 measured slower: fewer waves in flight)
  const long rows_per_wg = (long)rows * (256 / cols);
  const int ch = C >= 4 ? 4 : C;                              // channel planes per thread; full = no partial last chunk
  const bool full = C % ch == 0;
  GDRNPP_REQUIRE(((pooled_h + rows_per_wg - 1) / rows_per_wg) * col_tiles < (1l << 31), GDRNPP_ELIMIT, "gdrnpp_roi_align: output plane too large");
  const dim3 grid((unsigned)(((pooled_h + rows_per_wg - 1) / rows_per_wg) * col_tiles), (unsigned)((C + ch - 1) / ch), (unsigned)n_rois);
#define GDRNPP_RA(AL, CHV, FU) hipLaunchKernelGGL((roi_align_kernel<AL, rows, true, CHV, FU>), grid, dim3(256), 0, (hipStream_t)stream, x, rois, out, C, H, W, \
                                                 pooled_h, pooled_w, spatial_scale, sampling_ratio, col_tiles, cols)
#define GDRNPP_RA_C(AL) do { if (ch == 1) GDRNPP_RA(AL, 1, true); else if (ch == 2) GDRNPP_RA(AL, 2, true); \
    else if (ch == 3) GDRNPP_RA(AL, 3, true); else if (full) GDRNPP_RA(AL, 4, true); else GDRNPP_RA(AL, 4, false); } while (0)
  if (aligned) GDRNPP_RA_C(true); else GDRNPP_RA_C(false);
#undef GDRNPP_RA_C
#undef GDRNPP_RA
  return gdrnpp::check_launch("gdrnpp_roi_align");
}
