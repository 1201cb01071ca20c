/* ORACLE (test infrastructure, not shipped): CPU restatement of the YOLOX detection post-processing that feeds the
 * GDRNPP ROI path — det/yolox/utils/boxes.py:34-74 (`postprocess`) with torchvision.ops.nms / batched_nms restated.
 *
 * parity unpinned: torchvision is not installed here and its version is not pinned by the reference; the restatement
 * follows torchvision's CPU kernel (ops/cpu/nms_kernel.cpp: areas up front, candidates by descending score, box j is
 * suppressed when inter / (area_i + area_j - inter) > thr) and `_batched_nms_coordinate_trick` (boxes + class *
 * (max_coordinate + 1)).  Equal scores are ordered by ascending anchor index (torch.sort leaves ties unspecified).
 *
 * det [A, 5+C] = (cx, cy, w, h, obj, class scores...) of ONE image -> out [<=A, 7] = (x1, y1, x2, y2, obj, class_conf,
 * class) in keep order; returns the count.  float arithmetic, left to right, no FMA (-ffp-contract=off). */
#include <stdlib.h>
#include <string.h>

typedef struct { float score; int idx; } cand_t;
static int cmp_cand(const void* a, const void* b) {
  const cand_t *x = (const cand_t*)a, *y = (const cand_t*)b;
  if (x->score != y->score) return x->score > y->score ? -1 : 1;
  return x->idx - y->idx;
}

int oracle_yolox_postprocess(const float* det, int A, int C, float conf_thre, float nms_thre, int class_agnostic,
                             float* out) {
  const int S = 5 + C;
  float* rec = (float*)malloc(sizeof(float) * 7 * (size_t)A);
  cand_t* cand = (cand_t*)malloc(sizeof(cand_t) * (size_t)A);
  int n = 0;
  for (int a = 0; a < A; ++a) {
    const float* p = det + (size_t)a * S;
    float best = p[5];
    int arg = 0;
    for (int c = 1; c < C; ++c)
      if (p[5 + c] > best) { best = p[5 + c]; arg = c; }   /* first maximum wins */
    const float score = p[4] * best;
    if (!(score >= conf_thre)) continue;
    float* r = rec + 7 * (size_t)a;
    r[0] = p[0] - p[2] / 2; r[1] = p[1] - p[3] / 2; r[2] = p[0] + p[2] / 2; r[3] = p[1] + p[3] / 2;
    r[4] = p[4]; r[5] = best; r[6] = (float)arg;
    cand[n].score = score; cand[n].idx = a; ++n;
  }
  qsort(cand, (size_t)n, sizeof(cand_t), cmp_cand);
  float max_coord = 0.f;
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < 4; ++k) {
      const float v = rec[7 * (size_t)cand[i].idx + k];
      if (i == 0 && k == 0) max_coord = v; else if (v > max_coord) max_coord = v;
    }
  float* bx = (float*)malloc(sizeof(float) * 4 * (size_t)(n > 0 ? n : 1));
  float* area = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
  for (int i = 0; i < n; ++i) {
    const float* r = rec + 7 * (size_t)cand[i].idx;
    const float off = class_agnostic ? 0.f : r[6] * (max_coord + 1);
    for (int k = 0; k < 4; ++k) bx[4 * i + k] = class_agnostic ? r[k] : r[k] + off;
    area[i] = (bx[4 * i + 2] - bx[4 * i]) * (bx[4 * i + 3] - bx[4 * i + 1]);
  }
  unsigned char* sup = (unsigned char*)calloc((size_t)(n > 0 ? n : 1), 1);
  int kept = 0;
  for (int i = 0; i < n; ++i) {
    if (sup[i]) continue;
    memcpy(out + 7 * (size_t)kept, rec + 7 * (size_t)cand[i].idx, sizeof(float) * 7);
    ++kept;
    for (int j = i + 1; j < n; ++j) {
      if (sup[j]) continue;
      const float xx1 = bx[4 * i] > bx[4 * j] ? bx[4 * i] : bx[4 * j];
      const float yy1 = bx[4 * i + 1] > bx[4 * j + 1] ? bx[4 * i + 1] : bx[4 * j + 1];
      const float xx2 = bx[4 * i + 2] < bx[4 * j + 2] ? bx[4 * i + 2] : bx[4 * j + 2];
      const float yy2 = bx[4 * i + 3] < bx[4 * j + 3] ? bx[4 * i + 3] : bx[4 * j + 3];
      const float w = xx2 - xx1 > 0.f ? xx2 - xx1 : 0.f;
      const float h = yy2 - yy1 > 0.f ? yy2 - yy1 : 0.f;
      const float inter = w * h;
      const float ovr = inter / (area[i] + area[j] - inter);
      if (ovr > nms_thre) sup[j] = 1;
    }
  }
  free(sup); free(area); free(bx); free(cand); free(rec);
  return kept;
}
