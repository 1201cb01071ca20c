// fp32-accurate GEMM on the bf16 matrix cores of gfx950 by exact 3-way operand splitting — SURVEY.md §8 row a3.
//
//   C[m,n] = sum_k A[m,k] * W[n,k] + bias[n]   (+ the GELU / layer-scale+residual epilogues of gemm_f32.hip)
//
// On CDNA4 the fp32 MFMA runs at 1/16 of the bf16 rate (157 vs 2500 TFLOP/s).  Every fp32 value is EXACTLY the sum
// of three bf16 values  x = h + m + l  (h = rn(x), m = rn(x-h), l = x-h-m: 8 significant bits each, 24 together, both
// subtractions exact), a bf16*bf16 product is exact in fp32, and the bf16 MFMA accumulates in fp32.  Of the nine
// partial products of (ha+ma+la)*(hb+mb+lb) the six kept here (hh, hm, mh, hl, lh, mm) carry every term above
// 2^-26 relative; the three dropped (ml, lm, ll) are below the fp32 rounding of the sum.  Measured against an fp64
// product (tools/microbench_split_gemm.py, tests/test_gpu_net_kernels.py) the result is CLOSER to exact than the
// fp32-MFMA fmaf chain (max error 1.4e-7 vs 6.1e-7 at K=128), because the individual products carry no rounding.
// Six bf16 MFMAs replace sixteen fp32-MFMA-equivalents of matrix-pipe time: a 2.67x higher roofline (417 TFLOP/s).
//
// Data flow per 128x128x32 block tile (256 threads = 4 waves, each a 64x64 sub-tile = 2x2 MFMA 32x32x16 tiles):
//   A  fp32 [M,K] in HBM -> registers (4 x float4 per thread; 8 lanes cover one 128-byte row segment, so a wave
//      instruction touches 8 full cache lines) -> split in registers with v_cvt_pk_bf16_f32 -> three bf16 planes in
//      LDS (ds_write_b64, conflict-free);
//   W  split AND tiled once per weight (gdrnpp_pack_weight_bf16x3) into the exact LDS image of every 128x32 tile,
//      [N/128][K/32][split][k-block][row][8] bf16, so the B stage is six lane-linear 16-byte loads and stores;
//   workgroup -> tile map is XCD-aware: each XCD walks a contiguous range of tiles (n fastest), so the n-tiles that
//      share A rows hit the same L2;
//   LDS image per operand: [split][k-block of 8][row] 16-byte slots, plane pitch 132 slots: the 32 lanes of a
//      ds_read_b128 lane group read 32 consecutive slots (conflict-free), a fragment is one ds_read_b128;
//   per k-step of 16: 12 fragment reads feed 24 MFMAs (768 matrix-pipe cycles).
#include "common.hpp"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

#ifndef SPLIT_NBUF
#define SPLIT_NBUF 1
#endif
#ifndef SPLIT_OCC
#define SPLIT_OCC 2
#endif
constexpr int BM = 128, BN = 128, BK = 32, KB = BK / 8, PLANE = BM + 4, NBUF = SPLIT_NBUF;
constexpr int OPER_SLOTS = 3 * KB * PLANE;  // uint4 slots per operand image
enum { EPI_BIAS = 0, EPI_GELU = 1, EPI_SCALE_RES = 2 };

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }

// two fp32 -> three packed bf16 pairs with x = h + m + l exactly
struct Split3 { unsigned h, m, l; };
__device__ __forceinline__ Split3 split_pair(float x0, float x1) {
  Split3 o;
  f32x2 v = {x0, x1};
  o.h = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
  f32x2 r = {x0 - __uint_as_float(o.h << 16), x1 - __uint_as_float(o.h & 0xffff0000u)};
  o.m = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
  f32x2 q = {r[0] - __uint_as_float(o.m << 16), r[1] - __uint_as_float(o.m & 0xffff0000u)};
  o.l = __builtin_bit_cast(unsigned, __builtin_convertvector(q, bf16x2));
  return o;
}

// W f32[N][K] -> packed bf16 [N/128][K/32][3][4][128][8]; one thread per (row, k-block) = 8 consecutive k
__global__ void pack_weight_kernel(const float* __restrict__ W, uint4* __restrict__ packed, int N, int K) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int kbs = K / 8;
  if (i >= (long)N * kbs) return;
  const int n = (int)(i / kbs), kb_g = (int)(i % kbs);
  const float4 v0 = *reinterpret_cast<const float4*>(W + (size_t)n * K + kb_g * 8);
  const float4 v1 = *reinterpret_cast<const float4*>(W + (size_t)n * K + kb_g * 8 + 4);
  const Split3 p0 = split_pair(v0.x, v0.y), p1 = split_pair(v0.z, v0.w), p2 = split_pair(v1.x, v1.y), p3 = split_pair(v1.z, v1.w);
  const int tn = n / BN, row = n % BN, tk = kb_g / KB, kb = kb_g % KB;
  uint4* img = packed + ((size_t)tn * (K / BK) + tk) * (3 * KB * BN);
  img[(0 * KB + kb) * BN + row] = make_uint4(p0.h, p1.h, p2.h, p3.h);
  img[(1 * KB + kb) * BN + row] = make_uint4(p0.m, p1.m, p2.m, p3.m);
  img[(2 * KB + kb) * BN + row] = make_uint4(p0.l, p1.l, p2.l, p3.l);
}

template <int EPI>
__global__ __launch_bounds__(256, SPLIT_OCC) void gemm_split_kernel(const float* __restrict__ A,
                                                                    const uint4* __restrict__ Wp,
                                                                    const float* __restrict__ bias,
                                                                    const float* __restrict__ gamma,
                                                                    const float* __restrict__ resid,
                                                                    float* __restrict__ C, int M, int N, int K) {
  extern __shared__ uint4 lds4[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntn = N / BN;
  // XCD-aware map: hardware deals consecutive workgroup ids round-robin to the 8 XCDs; give each XCD a contiguous
  // range of tile ids (bijective for any grid size)
  const int nwg = gridDim.x, xcd = blockIdx.x & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
  const int tile_m = tile / ntn, tile_n = tile % ntn;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int nk = K / BK;

  // A staging: pass p covers rows p*32 + tid/8, lane%8 picks 4 consecutive k (half a k-block)
  const int lrow = tid >> 3, lkq = tid & 7;
  const float* Ag = A + (size_t)(m0 + lrow) * K + lkq * 4;
  const uint4* Wg = Wp + (size_t)tile_n * nk * (3 * KB * BN) + tid;
  float4 ra[4];
  uint4 rb0, rb1, rb2, rb3, rb4, rb5;  // scalars: an indexed array lands in scratch
  auto gload = [&](int kt) {
#pragma unroll
    for (int p = 0; p < 4; ++p) ra[p] = *reinterpret_cast<const float4*>(Ag + (size_t)(p * 32) * K + kt * BK);
    const uint4* w = Wg + (size_t)kt * (3 * KB * BN);
    rb0 = w[0]; rb1 = w[256]; rb2 = w[512]; rb3 = w[768]; rb4 = w[1024]; rb5 = w[1280];
  };
  auto lstore = [&](int buf) {
    uint4* a = lds4 + buf * 2 * OPER_SLOTS;
    uint4* b = a + OPER_SLOTS;
    const int kb = lkq >> 1, half = lkq & 1;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const Split3 p0 = split_pair(ra[p].x, ra[p].y), p1 = split_pair(ra[p].z, ra[p].w);
      uint2* dst = reinterpret_cast<uint2*>(a + kb * PLANE + p * 32 + lrow) + half;
      dst[0] = make_uint2(p0.h, p1.h);
      dst[2 * KB * PLANE] = make_uint2(p0.m, p1.m);
      dst[4 * KB * PLANE] = make_uint2(p0.l, p1.l);
    }
    // image slot i*256 + tid = plane (i*2 + tid/128), row tid%128
    uint4* bd = b + (tid >> 7) * PLANE + (tid & 127);
    bd[0] = rb0; bd[2 * PLANE] = rb1; bd[4 * PLANE] = rb2; bd[6 * PLANE] = rb3; bd[8 * PLANE] = rb4; bd[10 * PLANE] = rb5;
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  gload(0);
  lstore(0);
  __syncthreads();
  const int frow = lane & 31, fk = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = (NBUF == 2) ? (kt & 1) : 0;
    if (kt + 1 < nk) gload(kt + 1);
    const uint4* a = lds4 + buf * 2 * OPER_SLOTS + wm * 64 + frow;
    const uint4* b = lds4 + buf * 2 * OPER_SLOTS + OPER_SLOTS + wn * 64 + frow;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int kb = ks * 2 + fk;
      bf16x8 fa[3][2], fb[3][2];
#pragma unroll
      for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          fa[s][i] = __builtin_bit_cast(bf16x8, a[(s * KB + kb) * PLANE + i * 32]);
          fb[s][i] = __builtin_bit_cast(bf16x8, b[(s * KB + kb) * PLANE + i * 32]);
        }
      // smallest partial products first; the four accumulators rotate so no MFMA waits on its predecessor
      constexpr int TA[6] = {2, 0, 1, 1, 0, 0};
      constexpr int TB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[TA[t]][0], fb[TB[t]][0], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[TA[t]][0], fb[TB[t]][1], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[TA[t]][1], fb[TB[t]][0], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[TA[t]][1], fb[TB[t]][1], acc[1][1], 0, 0, 0);
      }
      if (NBUF == 2 && ks == 0 && kt + 1 < nk) lstore(buf ^ 1);
    }
    if (kt + 1 < nk) {
      __syncthreads();
      if (NBUF == 1) { lstore(0); __syncthreads(); }
    }
  }

  // epilogue: lane holds column (lane & 31) of rows (r&3) + 8*(r>>2) + 4*(lane>>5).  Each wave parks one 32x64 half
  // of its tile in LDS (the operand images are dead) and writes it back row-wise as float4.
  __syncthreads();
  float* T = reinterpret_cast<float*>(lds4) + wave * 32 * 65;  // [32][65] per wave
  const int c4 = (lane & 15) * 4;
  const int nb = n0 + wn * 64 + c4;
  const float4 bv = bias ? *reinterpret_cast<const float4*>(bias + nb) : make_float4(0.f, 0.f, 0.f, 0.f);
  float4 gv = make_float4(1.f, 1.f, 1.f, 1.f);
  if (EPI == EPI_SCALE_RES) gv = *reinterpret_cast<const float4*>(gamma + nb);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        T[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 65 + j * 32 + (lane & 31)] = acc[i][j][r];
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): same wave reads back
#pragma unroll 4
    for (int rr = 0; rr < 8; ++rr) {
      const int row = rr * 4 + (lane >> 4);
      const float* t = T + row * 65 + c4;
      float4 v = make_float4(t[0] + bv.x, t[1] + bv.y, t[2] + bv.z, t[3] + bv.w);
      const size_t off = (size_t)(m0 + wm * 64 + i * 32 + row) * N + nb;
      if (EPI == EPI_GELU) { v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w); }
      if (EPI == EPI_SCALE_RES) {
        const float4 rs = *reinterpret_cast<const float4*>(resid + off);
        v.x = rs.x + gv.x * v.x; v.y = rs.y + gv.y * v.y; v.z = rs.z + gv.z * v.z; v.w = rs.w + gv.w * v.w;
      }
      *reinterpret_cast<float4*>(C + off) = v;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);  // reads done before the second half overwrites T
  }
}

}  // namespace

extern "C" int gdrnpp_pack_weight_bf16x3(const float* W, void* packed, int N, int K, void* stream) {
  GDRNPP_REQUIRE(W && packed, GDRNPP_EINVAL, "gdrnpp_pack_weight_bf16x3: null pointer");
  GDRNPP_REQUIRE(N > 0 && K > 0 && N % BN == 0 && K % BK == 0, GDRNPP_ELIMIT,
                 "gdrnpp_pack_weight_bf16x3: N=%d K=%d must be multiples of %d/%d", N, K, BN, BK);
  const long threads = (long)N * (K / 8);
  hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W,
                     (uint4*)packed, N, K);
  return gdrnpp::check_launch("gdrnpp_pack_weight_bf16x3");
}

extern "C" int gdrnpp_linear_f32_split(const float* A, const void* W_packed, const float* bias, const float* gamma,
                                       const float* resid, float* C, int M, int N, int K, int epilogue,
                                       void* stream) {
  GDRNPP_REQUIRE(A && W_packed && C, GDRNPP_EINVAL, "gdrnpp_linear_f32_split: null pointer");
  GDRNPP_REQUIRE(M > 0 && N > 0 && K > 0 && M % BM == 0 && N % BN == 0 && K % BK == 0, GDRNPP_ELIMIT,
                 "gdrnpp_linear_f32_split: M=%d N=%d K=%d must be multiples of %d/%d/%d", M, N, K, BM, BN, BK);
  GDRNPP_REQUIRE(epilogue >= 0 && epilogue <= 2, GDRNPP_EINVAL, "gdrnpp_linear_f32_split: epilogue=%d", epilogue);
  GDRNPP_REQUIRE(epilogue != EPI_SCALE_RES || (gamma && resid), GDRNPP_EINVAL,
                 "gdrnpp_linear_f32_split: scale+residual epilogue needs gamma and resid");
  const long blocks = (long)(M / BM) * (N / BN);
  GDRNPP_REQUIRE(blocks < (1l << 31), GDRNPP_ELIMIT, "gdrnpp_linear_f32_split: grid too large");
  int lds = NBUF * 2 * OPER_SLOTS * (int)sizeof(uint4);
  if (lds < 4 * 32 * 65 * (int)sizeof(float)) lds = 4 * 32 * 65 * (int)sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  const uint4* Wp = (const uint4*)W_packed;
  if (epilogue == EPI_BIAS) {
    GDRNPP_HIP_TRY(hipFuncSetAttribute((const void*)gemm_split_kernel<EPI_BIAS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(gemm_split_kernel<EPI_BIAS>, dim3((unsigned)blocks), dim3(256), lds, st, A, Wp, bias, gamma, resid, C, M, N, K);
  } else if (epilogue == EPI_GELU) {
    GDRNPP_HIP_TRY(hipFuncSetAttribute((const void*)gemm_split_kernel<EPI_GELU>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(gemm_split_kernel<EPI_GELU>, dim3((unsigned)blocks), dim3(256), lds, st, A, Wp, bias, gamma, resid, C, M, N, K);
  } else {
    GDRNPP_HIP_TRY(hipFuncSetAttribute((const void*)gemm_split_kernel<EPI_SCALE_RES>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(gemm_split_kernel<EPI_SCALE_RES>, dim3((unsigned)blocks), dim3(256), lds, st, A, Wp, bias, gamma, resid, C, M, N, K);
  }
  return gdrnpp::check_launch("gdrnpp_linear_f32_split");
}
