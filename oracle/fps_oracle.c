/*
 * ORACLE — test infrastructure only (never imported by the product path).
 * CPU restatement of farthest point sampling,
 * core/csrc/fps/src/farthest_point_sampling.cpp:
 *   update_min_dist :40-54, find_max_dist_idx :56-73,
 *   sample_farthest_points :76-105 (start index made an argument in place of
 *   srand(time(0)); rand()%pn at :93-94), ..._init_center :122-160.
 * Pinned against the reference's own source compiled unmodified
 * (oracle/_ref/libfps_ref.so, tests/test_fps.py::test_oracle_vs_reference).
 * Build with -O2 -ffp-contract=off (no FMA, like the reference's x86-64 -O2).
 */
#include <float.h>
#include <stdlib.h>

static float sqn(const float* a, const float* b) {
  float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
  return dx * dx + dy * dy + dz * dz; /* ((dx*dx)+(dy*dy))+(dz*dz), cpp:23 */
}

static int find_max(const float* md, const unsigned char* mask, int pn) {
  int max_idx = 0;
  float max_d = 0.f;
  for (int i = 0; i < pn; ++i) {
    if (mask[i]) continue;
    if (md[i] > max_d) { max_idx = i; max_d = md[i]; }
  }
  return max_idx;
}

static void loop(const float* pts, int* idxs, int pn, int sn, float* md, unsigned char* mask, int cur) {
  for (int s = 0; s < sn; ++s) {
    mask[cur] = 1;
    idxs[s] = cur;
    if (s < sn - 1) {
      for (int i = 0; i < pn; ++i) {
        if (mask[i]) continue;
        float d = sqn(pts + 3 * i, pts + 3 * cur);
        if (d < md[i]) md[i] = d;
      }
      cur = find_max(md, mask, pn);
    }
  }
}

/* mode 0: start index given; mode 1: init-center */
void oracle_fps(const float* pts, int* idxs, int pn, int sn, int mode, int start) {
  float* md = (float*)malloc(sizeof(float) * pn);
  unsigned char* mask = (unsigned char*)calloc(pn, 1);
  for (int i = 0; i < pn; ++i) md[i] = FLT_MAX;
  int cur = start;
  if (mode == 1) {
    float mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX}, mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
    for (int i = 0; i < pn; ++i)
      for (int c = 0; c < 3; ++c) {
        float v = pts[3 * i + c];
        mx[c] = mx[c] > v ? mx[c] : v; /* std::max(a,b) = (a<b)?b:a */
        mn[c] = v < mn[c] ? v : mn[c]; /* std::min(a,b) = (b<a)?b:a */
      }
    float inv = 1.f / 2.f; /* operator/ multiplies by the reciprocal, cpp:21 */
    float ctr[3] = {(mx[0] + mn[0]) * inv, (mx[1] + mn[1]) * inv, (mx[2] + mn[2]) * inv};
    for (int i = 0; i < pn; ++i) {
      float d = sqn(pts + 3 * i, ctr);
      md[i] = (md[i] < d) ? md[i] : d; /* std::min(d, md) */
    }
    cur = find_max(md, mask, pn);
  }
  loop(pts, idxs, pn, sn, md, mask, cur);
  free(md);
  free(mask);
}
