// Bidirectional nearest-neighbour (chamfer) distance on gfx950 — SURVEY.md §8 row a12.
//
// Behavioural spec: core/csrc/torch_nndistance/src/nnd_cpu.cpp:3-25 (nnsearch),
// nnd_cuda_kernel.cu:8-162 (tiled CUDA forward), :164-222 (gradient scatter).
// Semantics kept: squared L2 in fp32 as ((dx*dx)+(dy*dy))+(dz*dz) without FMA,
// FIRST minimum wins (strict '<' scan in increasing k) => idx is bit-exact.
//
// Design: one launch covers both directions.  A workgroup of 4 waves owns 64
// query points (one per lane); the target set streams through LDS in 1024-point
// float4 tiles (one coalesced 16-byte load per thread per tile), each wave scans
// its 256-point quarter of every tile with broadcast ds_read_b128, and the four
// partial (dist, idx) pairs are merged lexicographically in LDS.  64 queries
// per workgroup (instead of the reference's fixed 32x16 grid of 512-thread
// blocks) gives b*(n+m)/64 workgroups, enough to cover 256 CUs at the sizes this
// op sees (b=10, n=1000, m=1500 -> 400 workgroups).
// Roofline: compute-side (10 VALU ops per pair), algorithmic HBM bytes
// 12*(n+m)*b read + 8*(n+m)*b written.
#include "common.hpp"
#include <cfloat>

namespace {

constexpr int kQ = 64;          // queries per workgroup (one wave-width)
constexpr int kWavesPerWG = 4;  // waves that split each target tile
constexpr int kTile = 1024;     // targets per LDS tile (16 KiB as float4)

__global__ __launch_bounds__(kQ* kWavesPerWG) void nnd_forward_kernel(
    const float* __restrict__ xyz1, const float* __restrict__ xyz2, float* __restrict__ dist1,
    float* __restrict__ dist2, int* __restrict__ idx1, int* __restrict__ idx2, int n, int m) {
  __shared__ float4 tile[kTile];
  __shared__ float s_d[kWavesPerWG][kQ];
  __shared__ int s_i[kWavesPerWG][kQ];

  const int bi = blockIdx.y;
  const int nblk1 = (n + kQ - 1) / kQ;
  int blk = blockIdx.x;
  // direction 0: queries = xyz1, targets = xyz2; direction 1: the other way
  const bool dir1 = blk >= nblk1;
  if (dir1) blk -= nblk1;
  const float* q = dir1 ? xyz2 : xyz1;
  const float* t = dir1 ? xyz1 : xyz2;
  const int nq = dir1 ? m : n, nt = dir1 ? n : m;
  float* dist = dir1 ? dist2 : dist1;
  int* idx = dir1 ? idx2 : idx1;
  q += (size_t)bi * nq * 3;
  t += (size_t)bi * nt * 3;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = blk * kQ + lane;
  float x1 = 0.f, y1 = 0.f, z1 = 0.f;
  if (j < nq) { x1 = q[3 * j]; y1 = q[3 * j + 1]; z1 = q[3 * j + 2]; }

  float best = FLT_MAX;
  int besti = 0;
  bool first = true;  // nnd_cpu.cpp:16 `k==0 || d<best`
  for (int k0 = 0; k0 < nt; k0 += kTile) {
    const int cnt = min(kTile, nt - k0);
    __syncthreads();
    for (int k = threadIdx.x; k < cnt; k += kQ * kWavesPerWG) {
      const float* p = t + 3 * (size_t)(k0 + k);
      tile[k] = make_float4(p[0], p[1], p[2], 0.f);
    }
    __syncthreads();
    const int lo = wave * (kTile / kWavesPerWG);
    const int hi = min(cnt, lo + kTile / kWavesPerWG);
#pragma unroll 4
    for (int k = lo; k < hi; ++k) {
      const float4 p = tile[k];  // wave-uniform address: LDS broadcast
      float dx = p.x - x1, dy = p.y - y1, dz = p.z - z1;
      float d = (dx * dx + dy * dy) + dz * dz;
      if (first || d < best) { best = d; besti = k0 + k; first = false; }
    }
  }
  s_d[wave][lane] = best;
  s_i[wave][lane] = first ? 0x7fffffff : besti;
  __syncthreads();
  if (wave == 0 && j < nq) {
    float bd = s_d[0][lane];
    int bidx = s_i[0][lane];
#pragma unroll
    for (int w = 1; w < kWavesPerWG; ++w) {
      float d = s_d[w][lane];
      int i = s_i[w][lane];
      // lexicographic (dist, idx): equal distance -> lowest index = first found by a serial scan
      if (i != 0x7fffffff && (bidx == 0x7fffffff || d < bd || (d == bd && i < bidx))) { bd = d; bidx = i; }
    }
    if (bidx == 0x7fffffff) { bd = 0.f; bidx = 0; }  // nt == 0: best=0, besti=0 (nnd_cpu.cpp:9-10)
    dist[(size_t)bi * nq + j] = bd;
    idx[(size_t)bi * nq + j] = bidx;
  }
}

// nnd_cuda_kernel.cu:164-183 — grad of sum over j of dist[j] w.r.t. both sets
__global__ void nnd_grad_kernel(const float* __restrict__ xyz1, const float* __restrict__ xyz2,
                                const float* __restrict__ graddist1, const int* __restrict__ idx1,
                                float* __restrict__ gradxyz1, float* __restrict__ gradxyz2, int n, int m) {
  const int bi = blockIdx.y;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const float* p1 = xyz1 + ((size_t)bi * n + j) * 3;
  const int j2 = idx1[(size_t)bi * n + j];
  const float* p2 = xyz2 + ((size_t)bi * m + j2) * 3;
  const float g = graddist1[(size_t)bi * n + j] * 2;
  float* g1 = gradxyz1 + ((size_t)bi * n + j) * 3;
  float* g2 = gradxyz2 + ((size_t)bi * m + j2) * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float v = g * (p1[c] - p2[c]);
    atomicAdd(g1 + c, v);
    atomicAdd(g2 + c, -v);
  }
}

}  // namespace

extern "C" {

int gdrnpp_nnd_forward(const float* xyz1, const float* xyz2, float* dist1, float* dist2, int* idx1,
                       int* idx2, int b, int n, int m, void* stream) {
  GDRNPP_REQUIRE(xyz1 && xyz2 && dist1 && dist2 && idx1 && idx2, GDRNPP_EINVAL, "gdrnpp_nnd_forward: null pointer");
  GDRNPP_REQUIRE(b > 0 && n > 0 && m > 0, GDRNPP_EINVAL, "gdrnpp_nnd_forward: b=%d n=%d m=%d", b, n, m);
  GDRNPP_REQUIRE(b <= 65535, GDRNPP_ELIMIT, "gdrnpp_nnd_forward: b=%d > 65535", b);
  dim3 grid((n + kQ - 1) / kQ + (m + kQ - 1) / kQ, b);
  hipLaunchKernelGGL(nnd_forward_kernel, grid, dim3(kQ * kWavesPerWG), 0, (hipStream_t)stream, xyz1, xyz2, dist1,
                     dist2, idx1, idx2, n, m);
  return gdrnpp::check_launch("gdrnpp_nnd_forward");
}

int gdrnpp_nnd_backward(const float* xyz1, const float* xyz2, float* gradxyz1, float* gradxyz2,
                        const float* graddist1, const float* graddist2, const int* idx1, const int* idx2,
                        int b, int n, int m, void* stream) {
  GDRNPP_REQUIRE(xyz1 && xyz2 && gradxyz1 && gradxyz2 && graddist1 && graddist2 && idx1 && idx2, GDRNPP_EINVAL,
                 "gdrnpp_nnd_backward: null pointer");
  GDRNPP_REQUIRE(b > 0 && n > 0 && m > 0 && b <= 65535, GDRNPP_EINVAL, "gdrnpp_nnd_backward: b=%d n=%d m=%d", b, n, m);
  hipStream_t st = (hipStream_t)stream;
  GDRNPP_HIP_TRY(hipMemsetAsync(gradxyz1, 0, sizeof(float) * 3 * (size_t)b * n, st));
  GDRNPP_HIP_TRY(hipMemsetAsync(gradxyz2, 0, sizeof(float) * 3 * (size_t)b * m, st));
  hipLaunchKernelGGL(nnd_grad_kernel, dim3((n + 255) / 256, b), dim3(256), 0, st, xyz1, xyz2, graddist1, idx1,
                     gradxyz1, gradxyz2, n, m);
  hipLaunchKernelGGL(nnd_grad_kernel, dim3((m + 255) / 256, b), dim3(256), 0, st, xyz2, xyz1, graddist2, idx2,
                     gradxyz2, gradxyz1, m, n);
  return gdrnpp::check_launch("gdrnpp_nnd_backward");
}

}  // extern "C"
