// Farthest point sampling on gfx950 — SURVEY.md §8 row a9.
//
// Behavioural spec: core/csrc/fps/src/farthest_point_sampling.cpp
//   update_min_dist      :40-54   (skip selected points; strict '<' update)
//   find_max_dist_idx    :56-73   (max_d starts at 0, strict '>' => lowest
//                                  index wins ties; nothing > 0 => index 0)
//   sample_farthest_points:76-105 (explicit start index instead of rand())
//   ..._init_center      :122-160 (min_dist seeded with distance to bbox centre)
// Squared distance is ((dx*dx)+(dy*dy))+(dz*dz) in fp32 with no FMA
// contraction (the reference is built by plain `gcc -O2`, fps/setup.py:5-7);
// this file is compiled with -ffp-contract=off so indices are bit-exact.
//
// Design (one workgroup of 1024 threads = 16 waves per cloud; clouds of a batch
// run on different CUs):
//   tier 1  pn <= 12288: x/y/z staged once in LDS as SoA (conflict-free
//           ds_read_b32, 144 KiB of the 160 KiB), the running min-distance of a
//           thread's <=12 points lives in registers; per sample: broadcast read
//           of the chosen point, register update, 6-step wave64 shuffle argmax,
//           16-entry LDS cross-wave argmax.  HBM traffic = 12*pn once.
//   tier 2  larger clouds: points + min-distance streamed from global memory
//           (L2-resident for pn up to ~10^5), same reduction.
#include "common.hpp"
#include <cfloat>
#include <climits>
#include <ctime>

namespace {

constexpr int kThreads = 1024;
constexpr int kWaves = kThreads / 64;
constexpr int kPPT = 12;                    // points per thread held in registers
constexpr int kLdsPts = kThreads * kPPT;    // 12288 points, 147456 B of LDS

struct Best {
  float v;
  int i;
};

__device__ __forceinline__ Best better(Best a, Best b) {
  // larger value wins; equal value -> lower index (first strictly-greater scan)
  bool take_b = (b.v > a.v) || (b.v == a.v && b.i < a.i);
  return take_b ? b : a;
}

__device__ __forceinline__ Best wave_argmax(Best x) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    Best o;
    o.v = __shfl_xor(x.v, off, 64);
    o.i = __shfl_xor(x.i, off, 64);
    x = better(x, o);
  }
  return x;
}

// block-wide argmax; result valid in every thread
__device__ __forceinline__ int block_argmax(Best x, float* s_v, int* s_i, int* s_cur) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  x = wave_argmax(x);
  if (lane == 0) {
    s_v[wave] = x.v;
    s_i[wave] = x.i;
  }
  __syncthreads();
  if (wave == 0) {
    Best y;
    y.v = lane < kWaves ? s_v[lane] : 0.f;
    y.i = lane < kWaves ? s_i[lane] : INT_MAX;
    y = wave_argmax(y);
    if (lane == 0) *s_cur = (y.v > 0.f) ? y.i : 0;  // nothing > 0 -> index 0 (cpp:58-59)
  }
  __syncthreads();
  return *s_cur;
}

__device__ __forceinline__ float sqdist(float ax, float ay, float az, float bx, float by, float bz) {
  float dx = ax - bx, dy = ay - by, dz = az - bz;
  return (dx * dx + dy * dy) + dz * dz;
}

__global__ __launch_bounds__(kThreads) void fps_lds_kernel(const float* __restrict__ pts,
                                                            int* __restrict__ idxs,
                                                            const int* __restrict__ start_idx,
                                                            int pn, int sn, int mode) {
  extern __shared__ float lds[];
  float* sx = lds;
  float* sy = lds + kLdsPts;
  float* sz = lds + 2 * kLdsPts;
  __shared__ float s_v[kWaves];
  __shared__ int s_i[kWaves];
  __shared__ int s_cur;
  __shared__ float s_red[6 * kWaves];

  const int b = blockIdx.x;
  const float* p = pts + (size_t)b * pn * 3;
  int* out = idxs + (size_t)b * sn;
  const int tid = threadIdx.x;

  // coalesced stage: flat float index -> SoA
  for (int f = tid; f < pn * 3; f += kThreads) {
    float v = p[f];
    int i = f / 3, c = f - i * 3;
    (c == 0 ? sx : (c == 1 ? sy : sz))[i] = v;
  }
  __syncthreads();

  float md[kPPT];
#pragma unroll
  for (int k = 0; k < kPPT; ++k) md[k] = FLT_MAX;

  int cur;
  if (mode == 1) {
    // bbox centre (cpp:131-137): max/min reductions are order independent
    float mx = -FLT_MAX, my = -FLT_MAX, mz = -FLT_MAX, nx = FLT_MAX, ny = FLT_MAX, nz = FLT_MAX;
#pragma unroll
    for (int k = 0; k < kPPT; ++k) {
      int i = tid + k * kThreads;
      if (i < pn) {
        float x = sx[i], y = sy[i], z = sz[i];
        mx = fmaxf(mx, x); my = fmaxf(my, y); mz = fmaxf(mz, z);
        nx = fminf(nx, x); ny = fminf(ny, y); nz = fminf(nz, z);
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      mx = fmaxf(mx, __shfl_xor(mx, off, 64)); my = fmaxf(my, __shfl_xor(my, off, 64));
      mz = fmaxf(mz, __shfl_xor(mz, off, 64)); nx = fminf(nx, __shfl_xor(nx, off, 64));
      ny = fminf(ny, __shfl_xor(ny, off, 64)); nz = fminf(nz, __shfl_xor(nz, off, 64));
    }
    const int lane = tid & 63, wave = tid >> 6;
    if (lane == 0) {
      s_red[wave * 6 + 0] = mx; s_red[wave * 6 + 1] = my; s_red[wave * 6 + 2] = mz;
      s_red[wave * 6 + 3] = nx; s_red[wave * 6 + 4] = ny; s_red[wave * 6 + 5] = nz;
    }
    __syncthreads();
    mx = my = mz = -FLT_MAX; nx = ny = nz = FLT_MAX;
    for (int w = 0; w < kWaves; ++w) {
      mx = fmaxf(mx, s_red[w * 6 + 0]); my = fmaxf(my, s_red[w * 6 + 1]); mz = fmaxf(mz, s_red[w * 6 + 2]);
      nx = fminf(nx, s_red[w * 6 + 3]); ny = fminf(ny, s_red[w * 6 + 4]); nz = fminf(nz, s_red[w * 6 + 5]);
    }
    // (max+min)/2.f is implemented as *(1.f/2.f) (cpp:21,138)
    const float cx = (mx + nx) * 0.5f, cy = (my + ny) * 0.5f, cz = (mz + nz) * 0.5f;
    Best best{0.f, INT_MAX};
#pragma unroll
    for (int k = 0; k < kPPT; ++k) {
      int i = tid + k * kThreads;
      if (i < pn) {
        float d = sqdist(sx[i], sy[i], sz[i], cx, cy, cz);
        md[k] = (FLT_MAX < d) ? FLT_MAX : d;  // std::min(d, FLT_MAX) = (b<a)?b:a (cpp:141)
        if (md[k] > best.v) { best.v = md[k]; best.i = i; }
      }
    }
    cur = block_argmax(best, s_v, s_i, &s_cur);
  } else {
    cur = start_idx ? start_idx[b] : 0;
    if (cur < 0 || cur >= pn) cur = 0;
  }

  for (int s = 0; s < sn; ++s) {
    if (tid == 0) out[s] = cur;
    if (s == sn - 1) break;
    const float cx = sx[cur], cy = sy[cur], cz = sz[cur];
    Best best{0.f, INT_MAX};
#pragma unroll
    for (int k = 0; k < kPPT; ++k) {
      int i = tid + k * kThreads;
      if (i < pn) {
        if (i == cur) md[k] = -1.f;  // mask[cur]=true (cpp:98); never updated nor chosen again
        if (md[k] >= 0.f) {
          float d = sqdist(sx[i], sy[i], sz[i], cx, cy, cz);
          if (d < md[k]) md[k] = d;
          if (md[k] > best.v) { best.v = md[k]; best.i = i; }
        }
      }
    }
    cur = block_argmax(best, s_v, s_i, &s_cur);
  }
}

// tier 2: streaming from global memory; md f32[b,pn] workspace
__global__ __launch_bounds__(kThreads) void fps_global_kernel(const float* __restrict__ pts,
                                                               int* __restrict__ idxs,
                                                               const int* __restrict__ start_idx,
                                                               float* __restrict__ md_ws, int pn,
                                                               int sn, int mode) {
  __shared__ float s_v[kWaves];
  __shared__ int s_i[kWaves];
  __shared__ int s_cur;
  __shared__ float s_red[6 * kWaves];
  const int b = blockIdx.x;
  const float* p = pts + (size_t)b * pn * 3;
  float* md = md_ws + (size_t)b * pn;
  int* out = idxs + (size_t)b * sn;
  const int tid = threadIdx.x;

  int cur;
  if (mode == 1) {
    float mx = -FLT_MAX, my = -FLT_MAX, mz = -FLT_MAX, nx = FLT_MAX, ny = FLT_MAX, nz = FLT_MAX;
    for (int i = tid; i < pn; i += kThreads) {
      float x = p[3 * i], y = p[3 * i + 1], z = p[3 * i + 2];
      mx = fmaxf(mx, x); my = fmaxf(my, y); mz = fmaxf(mz, z);
      nx = fminf(nx, x); ny = fminf(ny, y); nz = fminf(nz, z);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      mx = fmaxf(mx, __shfl_xor(mx, off, 64)); my = fmaxf(my, __shfl_xor(my, off, 64));
      mz = fmaxf(mz, __shfl_xor(mz, off, 64)); nx = fminf(nx, __shfl_xor(nx, off, 64));
      ny = fminf(ny, __shfl_xor(ny, off, 64)); nz = fminf(nz, __shfl_xor(nz, off, 64));
    }
    const int lane = tid & 63, wave = tid >> 6;
    if (lane == 0) {
      s_red[wave * 6 + 0] = mx; s_red[wave * 6 + 1] = my; s_red[wave * 6 + 2] = mz;
      s_red[wave * 6 + 3] = nx; s_red[wave * 6 + 4] = ny; s_red[wave * 6 + 5] = nz;
    }
    __syncthreads();
    mx = my = mz = -FLT_MAX; nx = ny = nz = FLT_MAX;
    for (int w = 0; w < kWaves; ++w) {
      mx = fmaxf(mx, s_red[w * 6 + 0]); my = fmaxf(my, s_red[w * 6 + 1]); mz = fmaxf(mz, s_red[w * 6 + 2]);
      nx = fminf(nx, s_red[w * 6 + 3]); ny = fminf(ny, s_red[w * 6 + 4]); nz = fminf(nz, s_red[w * 6 + 5]);
    }
    const float cx = (mx + nx) * 0.5f, cy = (my + ny) * 0.5f, cz = (mz + nz) * 0.5f;
    Best best{0.f, INT_MAX};
    for (int i = tid; i < pn; i += kThreads) {
      float d = sqdist(p[3 * i], p[3 * i + 1], p[3 * i + 2], cx, cy, cz);
      float m = (FLT_MAX < d) ? FLT_MAX : d;  // std::min(d, FLT_MAX) (cpp:141)
      md[i] = m;
      if (m > best.v) { best.v = m; best.i = i; }
    }
    cur = block_argmax(best, s_v, s_i, &s_cur);
  } else {
    for (int i = tid; i < pn; i += kThreads) md[i] = FLT_MAX;
    cur = start_idx ? start_idx[b] : 0;
    if (cur < 0 || cur >= pn) cur = 0;
  }

  for (int s = 0; s < sn; ++s) {
    if (tid == 0) out[s] = cur;
    if (s == sn - 1) break;
    const float cx = p[3 * cur], cy = p[3 * cur + 1], cz = p[3 * cur + 2];
    Best best{0.f, INT_MAX};
    for (int i = tid; i < pn; i += kThreads) {  // each thread owns a fixed index set: no races on md
      float m = md[i];
      if (i == cur) { m = -1.f; md[i] = m; }
      if (m >= 0.f) {
        float d = sqdist(p[3 * i], p[3 * i + 1], p[3 * i + 2], cx, cy, cz);
        if (d < m) { m = d; md[i] = m; }
        if (m > best.v) { best.v = m; best.i = i; }
      }
    }
    cur = block_argmax(best, s_v, s_i, &s_cur);
  }
}

void host_fps(float* pts, int* idxs, int pn, int sn, int mode, int start) {
  auto fail = [&](const char* what, const char* why) {
    gdrnpp::set_error("farthest_point_sampling: %s: %s", what, why);
    fprintf(stderr, "[gdrnpp_hip] %s\n", gdrnpp_last_error());
    if (idxs) for (int i = 0; i < sn; ++i) idxs[i] = -1;  // loud: -1 is never a valid index
  };
  if (!pts || !idxs || pn <= 0 || sn <= 0) {
    gdrnpp::set_error("farthest_point_sampling: bad arguments pn=%d sn=%d", pn, sn);
    fprintf(stderr, "[gdrnpp_hip] %s\n", gdrnpp_last_error());
    return;
  }
  // one scratch block (kept per host thread): points | indices | start | workspace, 256-byte aligned pieces
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t n_pts = al(sizeof(float) * 3 * (size_t)pn), n_idx = al(sizeof(int) * (size_t)sn), n_start = 256;
  const size_t ws = gdrnpp_fps_workspace_bytes(1, pn);
  char* d = (char*)gdrnpp::shim_scratch(n_pts + n_idx + n_start + ws);
  if (!d) return fail("scratch", gdrnpp_last_error());
  float* d_pts = (float*)d;
  int* d_idx = (int*)(d + n_pts);
  int* d_start = (int*)(d + n_pts + n_idx);
  void* d_ws = ws ? (void*)(d + n_pts + n_idx + n_start) : nullptr;
  hipError_t e;
  if ((e = hipMemcpy(d_pts, pts, sizeof(float) * 3 * (size_t)pn, hipMemcpyHostToDevice)) != hipSuccess) return fail("hipMemcpy H2D", hipGetErrorString(e));
  if ((e = hipMemcpy(d_start, &start, sizeof(int), hipMemcpyHostToDevice)) != hipSuccess) return fail("hipMemcpy H2D", hipGetErrorString(e));
  const int rc = gdrnpp_fps(d_pts, d_idx, d_start, 1, pn, sn, mode, d_ws, nullptr);
  if (rc != 0) {
    fprintf(stderr, "[gdrnpp_hip] farthest_point_sampling failed (%d): %s\n", rc, gdrnpp_last_error());
    for (int i = 0; i < sn; ++i) idxs[i] = -1;
    return;
  }
  if ((e = hipMemcpy(idxs, d_idx, sizeof(int) * (size_t)sn, hipMemcpyDeviceToHost)) != hipSuccess) fail("hipMemcpy D2H", hipGetErrorString(e));  // syncs
}

}  // namespace

extern "C" {

size_t gdrnpp_fps_workspace_bytes(int b, int pn) {
  if (b <= 0 || pn <= kLdsPts) return 0;
  return sizeof(float) * (size_t)b * (size_t)pn;
}

int gdrnpp_fps(const float* pts, int* idxs, const int* start_idx, int b, int pn, int sn, int mode,
               void* workspace, void* stream) {
  GDRNPP_REQUIRE(pts && idxs, GDRNPP_EINVAL, "gdrnpp_fps: null pointer");
  GDRNPP_REQUIRE(b > 0 && pn > 0 && sn > 0, GDRNPP_EINVAL, "gdrnpp_fps: b=%d pn=%d sn=%d", b, pn, sn);
  GDRNPP_REQUIRE(mode == 0 || mode == 1, GDRNPP_EINVAL, "gdrnpp_fps: mode=%d", mode);
  hipStream_t st = (hipStream_t)stream;
  if (pn <= kLdsPts) {
    const int lds_bytes = 3 * kLdsPts * (int)sizeof(float);
    if (int rc = gdrnpp::ensure_dynamic_lds((const void*)fps_lds_kernel, lds_bytes)) return rc;
    hipLaunchKernelGGL(fps_lds_kernel, dim3(b), dim3(kThreads), lds_bytes, st, pts, idxs, start_idx, pn, sn, mode);
  } else {
    GDRNPP_REQUIRE(workspace, GDRNPP_EINVAL, "gdrnpp_fps: pn=%d needs a workspace of %zu bytes", pn,
                   gdrnpp_fps_workspace_bytes(b, pn));
    hipLaunchKernelGGL(fps_global_kernel, dim3(b), dim3(kThreads), 0, st, pts, idxs, start_idx,
                       (float*)workspace, pn, sn, mode);
  }
  return gdrnpp::check_launch("gdrnpp_fps");
}

void farthest_point_sampling(float* pts, int* idxs, int pn, int sn) {
  // farthest_point_sampling.cpp:93-94: srand(time(0)); rand()%pn
  if (pn <= 0) { host_fps(pts, idxs, pn, sn, 0, 0); return; }
  srand((unsigned)time(nullptr));
  int start = rand() % pn;
  host_fps(pts, idxs, pn, sn, 0, start);
}

void farthest_point_sampling_init_center(float* pts, int* idxs, int pn, int sn) {
  host_fps(pts, idxs, pn, sn, 1, 0);
}

}  // extern "C"
