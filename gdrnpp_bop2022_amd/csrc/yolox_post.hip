// YOLOX detection post-processing on the device — SURVEY.md §8(f) rank 3: the decode + class argmax + confidence filter +
// torchvision batched_nms of det/yolox/utils/boxes.py:34-74, whose kept boxes are the ROIs the GDRNPP path starts from
// (core/gdrn_modeling/demo/predictor_yolo.py:84-165).  torchvision is not installed on the target, so NMS needs a kernel.
//
// Three launches for a whole batch of images, no host round trip:
//   1. yolox_decode_sort_kernel   one 1024-thread workgroup per image: every anchor's (cx,cy,w,h) -> corners, class
//      max / first argmax, score = obj * class_conf, filter; survivors get a 64-bit key (~score bits, anchor) and the
//      keys are bitonic-sorted in LDS (16384 keys = 128 KiB) -> candidates by descending score, ties by anchor index;
//      the sorted candidate records and max_coordinate (torchvision's batched-NMS offset) go to the workspace;
//   2. nms_mask_kernel            the classic 64x64-tile suppression bitmask (row i, 64 columns per u64);
//   3. nms_scan_kernel            one wave per image walks the candidates in order, OR-ing the rows of kept boxes.
// IoU arithmetic is torchvision's (areas, max/min corners, inter / (a_i + a_j - inter) > thr) in float without FMA, so
// the kept set equals the oracle's bit for bit.  Integer / index work, HBM-bound: A*(5+C)*4 B read per image.
#include "common.hpp"

namespace {

constexpr int kSortCap = 16384;  // keys sorted in LDS (YOLOX at 640x640 has 8400 anchors)

struct Cand { float x1, y1, x2, y2, obj, cls_conf, cls, score; };  // 32 B

__device__ __forceinline__ unsigned long long make_key(float score, int idx) {
  return ((unsigned long long)(~__float_as_uint(score)) << 32) | (unsigned)idx;  // score >= 0: bits are monotonic
}

__global__ __launch_bounds__(1024) void yolox_decode_sort_kernel(const float* __restrict__ det, int A, int C, float conf_thre,
                                                                Cand* __restrict__ cands, int* __restrict__ n_cand,
                                                                float* __restrict__ max_coord) {
  extern __shared__ unsigned long long keys[];  // [npad]
  __shared__ int s_count;
  __shared__ float s_max[16];
  const int b = blockIdx.x, tid = threadIdx.x, S = 5 + C;
  const float* d = det + (size_t)b * A * S;
  int npad = 64;
  while (npad < A) npad <<= 1;
  if (tid == 0) s_count = 0;
  __syncthreads();
  for (int a = tid; a < npad; a += blockDim.x) {
    unsigned long long key = ~0ull;
    if (a < A) {
      const float* p = d + (size_t)a * S;
      float best = p[5];
      for (int c = 1; c < C; ++c) best = p[5 + c] > best ? p[5 + c] : best;
      const float score = p[4] * best;
      if (score >= conf_thre) {
        key = make_key(score, a);
        atomicAdd(&s_count, 1);
      }
    }
    keys[a] = key;
  }
  __syncthreads();
  const int n = s_count;
  // bitonic sort, ascending: survivors first (descending score, ascending anchor), ~0 padding last
  for (int k = 2; k <= npad; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < npad; i += blockDim.x) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long x = keys[i], y = keys[ixj];
          const bool up = (i & k) == 0;
          if ((x > y) == up) { keys[i] = y; keys[ixj] = x; }
        }
      }
      __syncthreads();
    }
  // candidate records in sorted order + max coordinate over them
  float m = -3.402823466e38f;
  Cand* out = cands + (size_t)b * A;
  for (int i = tid; i < n; i += blockDim.x) {
    const int a = (int)(keys[i] & 0xffffffffu);
    const float* p = d + (size_t)a * S;
    float best = p[5];
    int arg = 0;
    for (int c = 1; c < C; ++c)
      if (p[5 + c] > best) { best = p[5 + c]; arg = c; }
    Cand r;
    r.x1 = p[0] - p[2] / 2; r.y1 = p[1] - p[3] / 2; r.x2 = p[0] + p[2] / 2; r.y2 = p[1] + p[3] / 2;
    r.obj = p[4]; r.cls_conf = best; r.cls = (float)arg; r.score = p[4] * best;
    out[i] = r;
    m = fmaxf(fmaxf(m, fmaxf(r.x1, r.y1)), fmaxf(r.x2, r.y2));
  }
  for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if ((tid & 63) == 0) s_max[tid >> 6] = m;
  __syncthreads();
  if (tid == 0) {
    float mm = s_max[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) mm = fmaxf(mm, s_max[w]);
    max_coord[b] = n > 0 ? mm : 0.f;
    n_cand[b] = n;
  }
}

__device__ __forceinline__ bool suppresses(const float4 a, float area_a, const float4 bx, float nms_thre) {
  const float xx1 = a.x > bx.x ? a.x : bx.x, yy1 = a.y > bx.y ? a.y : bx.y;
  const float xx2 = a.z < bx.z ? a.z : bx.z, yy2 = a.w < bx.w ? a.w : bx.w;
  const float w = xx2 - xx1 > 0.f ? xx2 - xx1 : 0.f, h = yy2 - yy1 > 0.f ? yy2 - yy1 : 0.f;
  const float inter = w * h;
  const float area_b = (bx.z - bx.x) * (bx.w - bx.y);
  return inter / (area_a + area_b - inter) > nms_thre;
}

__device__ __forceinline__ float4 nms_box(const Cand& c, float off_unit) {  // boxes + class * (max_coordinate + 1)
  const float off = c.cls * off_unit;
  return make_float4(c.x1 + off, c.y1 + off, c.x2 + off, c.y2 + off);
}

// grid (col block, row block, image); 64 threads: thread t owns row (row block * 64 + t)
__global__ __launch_bounds__(64) void nms_mask_kernel(const Cand* __restrict__ cands, const int* __restrict__ n_cand,
                                                      const float* __restrict__ max_coord, int A, float nms_thre,
                                                      int class_agnostic, unsigned long long* __restrict__ mask, int words) {
  const int b = blockIdx.z, rb = blockIdx.y, cb = blockIdx.x, n = n_cand[b];
  if (cb < rb || rb * 64 >= n || cb * 64 >= n) return;
  __shared__ float4 s_box[64];
  const Cand* c = cands + (size_t)b * A;
  const float unit = class_agnostic ? 0.f : max_coord[b] + 1;
  const int j0 = cb * 64, t = threadIdx.x;
  if (j0 + t < n) s_box[t] = class_agnostic ? make_float4(c[j0 + t].x1, c[j0 + t].y1, c[j0 + t].x2, c[j0 + t].y2) : nms_box(c[j0 + t], unit);
  __syncthreads();
  const int i = rb * 64 + t;
  if (i >= n) return;
  const float4 a = class_agnostic ? make_float4(c[i].x1, c[i].y1, c[i].x2, c[i].y2) : nms_box(c[i], unit);
  const float area_a = (a.z - a.x) * (a.w - a.y);
  unsigned long long bits = 0;
  const int jn = min(64, n - j0);
  for (int jj = (rb == cb ? t + 1 : 0); jj < jn; ++jj)
    if (suppresses(a, area_a, s_box[jj], nms_thre)) bits |= 1ull << jj;
  mask[((size_t)b * A + i) * words + cb] = bits;
}

// one wave per image
__global__ __launch_bounds__(64) void nms_scan_kernel(const Cand* __restrict__ cands, const int* __restrict__ n_cand, int A,
                                                      const unsigned long long* __restrict__ mask, int words,
                                                      float* __restrict__ out, int* __restrict__ out_count, int max_det) {
  extern __shared__ unsigned long long remv[];  // [words]
  const int b = blockIdx.x, lane = threadIdx.x, n = n_cand[b];
  for (int w = lane; w < words; w += 64) remv[w] = 0;
  __syncthreads();
  const Cand* c = cands + (size_t)b * A;
  int kept = 0;
  for (int i = 0; i < n; ++i) {
    const bool dead = (remv[i >> 6] >> (i & 63)) & 1ull;   // uniform across the wave
    if (dead) continue;
    if (kept < max_det && lane < 7) {
      const float* r = reinterpret_cast<const float*>(c + i);
      out[((size_t)b * max_det + kept) * 7 + lane] = r[lane];
    }
    ++kept;
    const unsigned long long* row = mask + ((size_t)b * A + i) * words;
    for (int w = (i >> 6) + lane; w < (n + 63) >> 6; w += 64) remv[w] |= row[w];
    __syncthreads();
  }
  if (lane == 0) out_count[b] = kept;
}

}  // namespace

extern "C" size_t gdrnpp_yolox_postprocess_workspace_bytes(int B, int A) {
  if (B <= 0 || A <= 0) return 0;
  const size_t words = ((size_t)A + 63) / 64;
  return (size_t)B * A * sizeof(Cand) + (size_t)B * A * words * 8 + (size_t)B * 8 + 256;
}

extern "C" int gdrnpp_yolox_postprocess(const float* det_preds, int B, int A, int C, float conf_thre, float nms_thre,
                                        int class_agnostic, float* out_dets, int* out_count, int max_det, void* workspace,
                                        size_t workspace_bytes, void* stream) {
  GDRNPP_REQUIRE(det_preds && out_dets && out_count && workspace, GDRNPP_EINVAL, "gdrnpp_yolox_postprocess: null pointer");
  GDRNPP_REQUIRE(B > 0 && A > 0 && C > 0 && max_det > 0, GDRNPP_EINVAL, "gdrnpp_yolox_postprocess: B=%d A=%d C=%d max_det=%d", B, A,
                 C, max_det);
  GDRNPP_REQUIRE(A <= kSortCap, GDRNPP_ELIMIT, "gdrnpp_yolox_postprocess: A=%d anchors exceed the %d sorted in LDS", A, kSortCap);
  GDRNPP_REQUIRE(workspace_bytes >= gdrnpp_yolox_postprocess_workspace_bytes(B, A), GDRNPP_EINVAL,
                 "gdrnpp_yolox_postprocess: workspace too small");
  const int words = (A + 63) / 64;
  char* ws = (char*)workspace;
  Cand* cands = (Cand*)ws;
  unsigned long long* mask = (unsigned long long*)(ws + (size_t)B * A * sizeof(Cand));
  int* n_cand = (int*)(ws + (size_t)B * A * sizeof(Cand) + (size_t)B * A * words * 8);
  float* max_coord = (float*)(n_cand + B);
  hipStream_t st = (hipStream_t)stream;
  int npad = 64;
  while (npad < A) npad <<= 1;
  const int lds1 = npad * 8;
  if (int rc = gdrnpp::ensure_dynamic_lds((const void*)yolox_decode_sort_kernel, lds1)) return rc;
  hipLaunchKernelGGL(yolox_decode_sort_kernel, dim3(B), dim3(1024), lds1, st, det_preds, A, C, conf_thre, cands, n_cand, max_coord);
  hipLaunchKernelGGL(nms_mask_kernel, dim3(words, words, B), dim3(64), 0, st, cands, n_cand, max_coord, A, nms_thre,
                     class_agnostic, mask, words);
  hipLaunchKernelGGL(nms_scan_kernel, dim3(B), dim3(64), words * 8, st, cands, n_cand, A, mask, words, out_dets, out_count, max_det);
  return gdrnpp::check_launch("gdrnpp_yolox_postprocess");
}
