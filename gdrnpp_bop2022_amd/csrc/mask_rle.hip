// Instance masks for SAVE_RESULTS_ONLY — SURVEY.md §8(f) rank 4: gdrn_evaluator.py:914-945 pastes every 64x64 mask
// probability map into the full image (detectron2 paste_masks_in_image: bilinear grid_sample inside the ROI box,
// >= threshold), copies B full-size masks to the host and run-length encodes them with pycocotools, one by one.
// Here the full-size mask is never materialised: one workgroup per instance evaluates the pasted mask on the fly in
// COCO's column-major order and emits the run lengths directly (two passes: count transitions per column, prefix sum,
// write) — a few hundred bytes per instance leave the device instead of H*W.
// Sampling arithmetic is detectron2's / ATen's (see oracle/mask_rle_oracle.c) in float without FMA: the binary mask,
// hence every run length, equals the oracle's exactly.
#include "common.hpp"

namespace {

__device__ __forceinline__ int paste_bit(const float* __restrict__ m, int hm, int wm, float x0, float y0, float x1, float y1,
                                         int x, int y, float threshold) {
  const float gx = ((float)x + 0.5f - x0) / (x1 - x0) * 2 - 1;
  const float gy = ((float)y + 0.5f - y0) / (y1 - y0) * 2 - 1;
  const float ix = ((gx + 1) * wm - 1) / 2;
  const float iy = ((gy + 1) * hm - 1) / 2;
  const float fx = floorf(ix), fy = floorf(iy);
  const int ix_nw = (int)fx, iy_nw = (int)fy, ix_se = ix_nw + 1, iy_se = iy_nw + 1;
  const float nw = ((float)ix_se - ix) * ((float)iy_se - iy), ne = (ix - (float)ix_nw) * ((float)iy_se - iy);
  const float sw = ((float)ix_se - ix) * (iy - (float)iy_nw), se = (ix - (float)ix_nw) * (iy - (float)iy_nw);
  float out = 0.f;
  if (ix_nw >= 0 && ix_nw < wm && iy_nw >= 0 && iy_nw < hm) out += m[iy_nw * wm + ix_nw] * nw;
  if (ix_se >= 0 && ix_se < wm && iy_nw >= 0 && iy_nw < hm) out += m[iy_nw * wm + ix_se] * ne;
  if (ix_nw >= 0 && ix_nw < wm && iy_se >= 0 && iy_se < hm) out += m[iy_se * wm + ix_nw] * sw;
  if (ix_se >= 0 && ix_se < wm && iy_se >= 0 && iy_se < hm) out += m[iy_se * wm + ix_se] * se;
  return out >= threshold;
}

// one 1024-thread workgroup per instance; thread t owns columns t, t + 1024, ...
__global__ __launch_bounds__(1024) void paste_rle_kernel(const float* __restrict__ masks, const float* __restrict__ boxes, int hm,
                                                         int wm, int H, int W, float threshold, unsigned* __restrict__ counts,
                                                         int* __restrict__ n_runs, int max_runs) {
  extern __shared__ float s_mask[];      // [hm * wm]
  __shared__ int s_wave[17];
  __shared__ int s_base;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* m = masks + (size_t)b * hm * wm;
  for (int i = tid; i < hm * wm; i += blockDim.x) s_mask[i] = m[i];
  const float x0 = boxes[4 * b], y0 = boxes[4 * b + 1], x1 = boxes[4 * b + 2], y1 = boxes[4 * b + 3];
  if (tid == 0) s_base = 0;
  __syncthreads();
  unsigned* out = counts + (size_t)b * max_runs;
  // positions of value changes in the linear column-major index i = x*H + y (value before index 0 is 0) are written
  // to out[1..] first, turned into run lengths afterwards
  for (int xb = 0; xb < W; xb += blockDim.x) {
    const int x = xb + tid;
    int cnt = 0;
    int prev = 0;
    if (x < W) {
      prev = x > 0 ? paste_bit(s_mask, hm, wm, x0, y0, x1, y1, x - 1, H - 1, threshold) : 0;
      int p = prev;
      for (int y = 0; y < H; ++y) {
        const int v = paste_bit(s_mask, hm, wm, x0, y0, x1, y1, x, y, threshold);
        cnt += v != p;
        p = v;
      }
    }
    // exclusive prefix sum of cnt over the 1024 columns of this chunk
    int inc = cnt;
    for (int off = 1; off < 64; off <<= 1) {
      const int t = __shfl_up(inc, off, 64);
      if (lane >= off) inc += t;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    if (tid == 0) {
      int acc = 0;
      for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { const int t = s_wave[w]; s_wave[w] = acc; acc += t; }
      s_wave[16] = acc;
    }
    __syncthreads();
    int k = s_base + s_wave[wave] + inc - cnt;   // index of this column's first transition
    if (x < W && cnt > 0) {
      int p = prev;
      for (int y = 0; y < H; ++y) {
        const int v = paste_bit(s_mask, hm, wm, x0, y0, x1, y1, x, y, threshold);
        if (v != p) {
          if (k + 1 < max_runs) out[k + 1] = (unsigned)(x * H + y);
          ++k;
        }
        p = v;
      }
    }
    __syncthreads();
    if (tid == 0) s_base += s_wave[16];
    __syncthreads();
  }
  const int T = s_base;                  // number of transitions; runs = T + 1
  // run k = pos[k+1] - pos[k] with pos[0] = 0 and pos[T+1] = H*W.  In place, back to front is not possible in
  // parallel, so every thread first reads its neighbours, then all write after a barrier.
  const int runs = T + 1;
  const unsigned total = (unsigned)H * (unsigned)W;
  for (int base = 0; base < runs && base < max_runs; base += blockDim.x) {
    const int k = base + tid;
    unsigned len = 0;
    const bool ok = k < runs && k < max_runs;
    if (ok) {
      const unsigned lo = k == 0 ? 0u : out[k];
      const unsigned hi = (k == T) ? total : ((k + 1 < max_runs) ? out[k + 1] : total);
      len = hi - lo;
    }
    __syncthreads();
    if (ok) out[k] = len;
    __syncthreads();
  }
  if (tid == 0) n_runs[b] = runs;
}

}  // namespace

extern "C" int gdrnpp_paste_masks_rle(const float* mask_probs, const float* boxes_xyxy, int B, int mask_h, int mask_w,
                                      int im_H, int im_W, float threshold, unsigned* counts, int* n_runs, int max_runs,
                                      void* stream) {
  GDRNPP_REQUIRE(mask_probs && boxes_xyxy && counts && n_runs, GDRNPP_EINVAL, "gdrnpp_paste_masks_rle: null pointer");
  GDRNPP_REQUIRE(B > 0 && mask_h > 0 && mask_w > 0 && im_H > 0 && im_W > 0 && max_runs > 1, GDRNPP_EINVAL,
                 "gdrnpp_paste_masks_rle: bad shape");
  GDRNPP_REQUIRE((long)im_H * im_W < (1l << 31) && mask_h * mask_w * 4 <= 128 * 1024, GDRNPP_ELIMIT,
                 "gdrnpp_paste_masks_rle: image %dx%d or mask %dx%d too large", im_H, im_W, mask_h, mask_w);
  const int lds = mask_h * mask_w * (int)sizeof(float);
  if (int rc = gdrnpp::ensure_dynamic_lds((const void*)paste_rle_kernel, lds)) return rc;
  hipLaunchKernelGGL(paste_rle_kernel, dim3(B), dim3(1024), lds, (hipStream_t)stream, mask_probs, boxes_xyxy, mask_h, mask_w,
                     im_H, im_W, threshold, counts, n_runs, max_runs);
  return gdrnpp::check_launch("gdrnpp_paste_masks_rle");
}
