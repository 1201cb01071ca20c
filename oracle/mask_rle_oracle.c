/* ORACLE (test infrastructure, not shipped): CPU restatement of the SAVE_RESULTS_ONLY mask path of the reference —
 * gdrn_evaluator.py:914-945: detectron2 `paste_masks_in_image(mask_probs, boxes, (H, W), threshold)` followed by the
 * uncompressed COCO run-length encoding of lib/utils/mask_utils.py:96-109 (column-major, first run counts zeros).
 *
 * parity: the ENCODER is pinned — run lengths equal the reference's own `binary_mask_to_rle(compressed=False)` executed from source
 * (tests/golden/make_golden_pyref.py `rle_case`, tests/test_postproc_oracle.py); the PASTE is unpinned: detectron2 / pycocotools are not
 * installed.  The paste is detectron2's `_do_paste_mask`
 * (layers/mask_ops.py): grid x = ((x + 0.5 - x0) / (x1 - x0)) * 2 - 1, bilinear `grid_sample(align_corners=False,
 * padding_mode="zeros")` as ATen computes it (ix = ((g + 1) * W_in - 1) / 2; corners nw, ne, sw, se accumulated in that
 * order, out-of-range corners contribute 0), then `>= threshold`.  tests/ pin the sampling against torch's own
 * F.grid_sample on the CPU.  float arithmetic, left to right, no FMA (-ffp-contract=off). */
#include <math.h>

static float paste_value(const float* m, int hm, int wm, float x0, float y0, float x1, float y1, int x, int y) {
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
  return out;
}

/* one instance: mask f32[hm,wm], box (x0,y0,x1,y1) -> binary u8[H,W] (row-major, optional) and COCO run lengths.
 * returns the number of runs written to counts (capacity max_runs; returns -needed if it does not fit). */
int oracle_paste_mask_rle(const float* mask, int hm, int wm, const float* box, int H, int W, float threshold,
                          unsigned char* binary_out, unsigned* counts, int max_runs) {
  int n = 0, prev = 0;
  unsigned run = 0;
  for (int x = 0; x < W; ++x)
    for (int y = 0; y < H; ++y) {
      const int v = paste_value(mask, hm, wm, box[0], box[1], box[2], box[3], x, y) >= threshold;
      if (binary_out) binary_out[y * W + x] = (unsigned char)v;
      if (v != prev) {
        if (n < max_runs) counts[n] = run;
        ++n; run = 0; prev = v;
      }
      ++run;
    }
  if (n < max_runs) counts[n] = run;
  ++n;
  return n <= max_runs ? n : -n;
}
