// fp32 GEMM with fused epilogues for the ConvNeXt MLP of GDRN_Net on gfx950 — SURVEY.md §8 row a3.
//
//   C[m,n] = sum_k A[m,k] * W[n,k] + bias[n]          A: activations [M,K] (NHWC rows), W: nn.Linear weight [N,K]
//   EPI_GELU      : C = gelu_erf(C)                                         (timm Mlp.fc1 + act)
//   EPI_SCALE_RES : C = resid[m,n] + gamma[n] * C                           (timm Mlp.fc2 + layer scale + residual)
// hipBLASLt already runs these GEMMs at ~139 TFLOP/s; what it cannot do is the exact-erf GELU and the
// layer-scale/residual in its epilogue, which cost two further HBM passes over the [M,4C] / [M,C] tensors
// (4.1 + 1.5 ms per 128-ROI forward).  This kernel keeps fp32 exactness (v_mfma_f32_32x32x2_f32 = a k-ordered
// fmaf chain) and fuses them.
//
// Tiling: workgroup 256 threads = 4 waves, block tile 128x128x32, each wave a 64x64 sub-tile = 2x2 MFMA tiles of
// 32x32 (64 accumulator VGPRs), operands staged through LDS as [row][k] with pitch 33 (conflict-free ds_read_b32:
// the 32 lanes of a group read 32 consecutive rows of one k), double-buffered, global loads for tile t+1 issued
// before the MFMAs of tile t.  LDS 2*2*128*33*4 = 67.6 KB -> 2 workgroups per CU (2 waves per SIMD).
#include "common.hpp"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

#ifndef GEMM_BK
#define GEMM_BK 32
#endif
#ifndef GEMM_NBUF
#define GEMM_NBUF 2
#endif
constexpr int BM = 128, BN = 128, BK = GEMM_BK, PITCH = BK + 1, NBUF = GEMM_NBUF;
enum { EPI_BIAS = 0, EPI_GELU = 1, EPI_SCALE_RES = 2 };

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }

template <int EPI>
__global__ __launch_bounds__(256, NBUF == 2 ? 2 : 3) void gemm_tn_f32_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                             const float* __restrict__ bias,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ resid, float* __restrict__ C,
                                                             int M, int N, int K) {
  extern __shared__ float lds[];
  float* As = lds;                         // [2][BM][PITCH]
  float* Bs = lds + NBUF * BM * PITCH;     // [NBUF][BN][PITCH]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;  // 2x2 waves over the 128x128 tile
  const int ntn = N / BN;
  const int tile_m = blockIdx.x / ntn, tile_n = blockIdx.x % ntn;  // n fastest: neighbours share the A rows in L2
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // global -> register staging: each thread moves 4 float4 of A and 4 of W per k-tile
  constexpr int TPR = BK / 4, RPP = 256 / TPR, NP = BM / RPP;  // threads per row, rows per pass, passes
  const int lrow = tid / TPR;
  const int lk4 = (tid % TPR) * 4;
  const float* Ag = A + (size_t)(m0 + lrow) * K + lk4;
  const float* Wg = W + (size_t)(n0 + lrow) * K + lk4;
  float4 ra[NP], rb[NP];
  auto gload = [&](int kt) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      ra[p] = *reinterpret_cast<const float4*>(Ag + (size_t)(p * RPP) * K + kt * BK);
      rb[p] = *reinterpret_cast<const float4*>(Wg + (size_t)(p * RPP) * K + kt * BK);
    }
  };
  auto lstore = [&](int buf) {
    float* a = As + buf * BM * PITCH;
    float* b = Bs + buf * BN * PITCH;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      float* pa = a + (p * RPP + lrow) * PITCH + lk4;
      float* pb = b + (p * RPP + lrow) * PITCH + lk4;
      pa[0] = ra[p].x; pa[1] = ra[p].y; pa[2] = ra[p].z; pa[3] = ra[p].w;
      pb[0] = rb[p].x; pb[1] = rb[p].y; pb[2] = rb[p].z; pb[3] = rb[p].w;
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = K / BK;
  gload(0);
  lstore(0);
  __syncthreads();
  const int frow = lane & 31, fk = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = (NBUF == 2) ? (kt & 1) : 0;
    if (kt + 1 < nk) gload(kt + 1);
    const float* a = As + buf * BM * PITCH + (wm * 64 + frow) * PITCH + fk;
    const float* b = Bs + buf * BN * PITCH + (wn * 64 + frow) * PITCH + fk;
    // fragments are fetched one step (8 MFMAs = 512 matrix-pipe cycles) ahead of their use, so the LDS latency
    // never sits in front of an MFMA
    float fa[2][4], fb[2][4];  // [stage][a0(k), a0(k+2), a1(k), a1(k+2)]
    auto fetch = [&](int st, int kk2) {
      fa[st][0] = a[kk2 * 4]; fa[st][1] = a[kk2 * 4 + 2]; fa[st][2] = a[32 * PITCH + kk2 * 4]; fa[st][3] = a[32 * PITCH + kk2 * 4 + 2];
      fb[st][0] = b[kk2 * 4]; fb[st][1] = b[kk2 * 4 + 2]; fb[st][2] = b[32 * PITCH + kk2 * 4]; fb[st][3] = b[32 * PITCH + kk2 * 4 + 2];
    };
    fetch(0, 0);
#pragma unroll
    for (int kk2 = 0; kk2 < BK / 4; ++kk2) {
      const int st = kk2 & 1;
      if (kk2 + 1 < BK / 4) fetch(st ^ 1, kk2 + 1);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[st][0], fb[st][0], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[st][0], fb[st][2], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[st][2], fb[st][0], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[st][2], fb[st][2], acc[1][1], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[st][1], fb[st][1], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[st][1], fb[st][3], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[st][3], fb[st][1], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[st][3], fb[st][3], acc[1][1], 0, 0, 0);
      // park the next k-tile in the other LDS buffer while the matrix pipe works through the second half of this
      // one: the ds_writes issue in the shadow of the 64-cycle MFMAs instead of in front of the barrier
      if (NBUF == 2 && kk2 == BK / 8 - 1 && kt + 1 < nk) lstore(buf ^ 1);
    }
    if (kt + 1 < nk) {
      __syncthreads();
      if (NBUF == 1) { lstore(0); __syncthreads(); }
    }
  }

  // epilogue: lane holds column (lane & 31) of rows (r&3) + 8*(r>>2) + 4*(lane>>5).  Each wave parks its 64x64 tile in
  // LDS (the operand buffers are dead now) and writes it back row-wise as float4: 16 lanes cover one 256-byte row
  // segment, 4x fewer store instructions than per-element stores and full-sector writes.
  __syncthreads();
  float* T = lds + wave * 64 * 65;  // [64][65] per wave
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        T[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 65 + j * 32 + (lane & 31)] = acc[i][j][r];
  // same wave reads back: no workgroup barrier needed, only the LDS writes of this wave must have landed
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
  const int c4 = (lane & 15) * 4;      // 4 consecutive columns
  const int nb = n0 + wn * 64 + c4;
  const float4 bv = bias ? *reinterpret_cast<const float4*>(bias + nb) : make_float4(0.f, 0.f, 0.f, 0.f);
  float4 gv = make_float4(1.f, 1.f, 1.f, 1.f);
  if (EPI == EPI_SCALE_RES) gv = *reinterpret_cast<const float4*>(gamma + nb);
#pragma unroll 4
  for (int rr = 0; rr < 16; ++rr) {
    const int row = rr * 4 + (lane >> 4);
    const float* t = T + row * 65 + c4;
    float4 v = make_float4(t[0] + bv.x, t[1] + bv.y, t[2] + bv.z, t[3] + bv.w);
    const size_t off = (size_t)(m0 + wm * 64 + row) * N + nb;
    if (EPI == EPI_GELU) { v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w); }
    if (EPI == EPI_SCALE_RES) {
      const float4 rs = *reinterpret_cast<const float4*>(resid + off);
      v.x = rs.x + gv.x * v.x; v.y = rs.y + gv.y * v.y; v.z = rs.z + gv.z * v.z; v.w = rs.w + gv.w * v.w;
    }
    *reinterpret_cast<float4*>(C + off) = v;
  }
}

}  // namespace

extern "C" int gdrnpp_linear_f32(const float* A, const float* W, const float* bias, const float* gamma,
                                 const float* resid, float* C, int M, int N, int K, int epilogue, void* stream) {
  GDRNPP_REQUIRE(A && W && C, GDRNPP_EINVAL, "gdrnpp_linear_f32: null pointer");
  GDRNPP_REQUIRE(M > 0 && N > 0 && K > 0 && M % BM == 0 && N % BN == 0 && K % BK == 0, GDRNPP_ELIMIT,
                 "gdrnpp_linear_f32: M=%d N=%d K=%d must be multiples of %d/%d/%d", M, N, K, BM, BN, BK);
  GDRNPP_REQUIRE(epilogue >= 0 && epilogue <= 2, GDRNPP_EINVAL, "gdrnpp_linear_f32: epilogue=%d", epilogue);
  GDRNPP_REQUIRE(epilogue != EPI_SCALE_RES || (gamma && resid), GDRNPP_EINVAL,
                 "gdrnpp_linear_f32: scale+residual epilogue needs gamma and resid");
  const long blocks = (long)(M / BM) * (N / BN);
  GDRNPP_REQUIRE(blocks < (1l << 31), GDRNPP_ELIMIT, "gdrnpp_linear_f32: grid too large");
  int lds = NBUF * (BM + BN) * PITCH * (int)sizeof(float);
  if (lds < 4 * 64 * 65 * (int)sizeof(float)) lds = 4 * 64 * 65 * (int)sizeof(float);  // epilogue staging
  hipStream_t st = (hipStream_t)stream;
  if (epilogue == EPI_BIAS) {
    GDRNPP_HIP_TRY(hipFuncSetAttribute((const void*)gemm_tn_f32_kernel<EPI_BIAS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(gemm_tn_f32_kernel<EPI_BIAS>, dim3((unsigned)blocks), dim3(256), lds, st, A, W, bias, gamma, resid, C, M, N, K);
  } else if (epilogue == EPI_GELU) {
    GDRNPP_HIP_TRY(hipFuncSetAttribute((const void*)gemm_tn_f32_kernel<EPI_GELU>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(gemm_tn_f32_kernel<EPI_GELU>, dim3((unsigned)blocks), dim3(256), lds, st, A, W, bias, gamma, resid, C, M, N, K);
  } else {
    GDRNPP_HIP_TRY(hipFuncSetAttribute((const void*)gemm_tn_f32_kernel<EPI_SCALE_RES>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(gemm_tn_f32_kernel<EPI_SCALE_RES>, dim3((unsigned)blocks), dim3(256), lds, st, A, W, bias, gamma, resid, C, M, N, K);
  }
  return gdrnpp::check_launch("gdrnpp_linear_f32");
}
