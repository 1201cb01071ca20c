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
// Data flow per 128x128x16 block tile (256 threads = 4 waves, each a 64x64 sub-tile = 2x2 MFMA 32x32x16 tiles):
//   A  fp32 [M,K] in HBM -> registers (2 x float4 per thread; 4 lanes cover one 64-byte row segment) -> split in
//      registers with v_cvt_pk_bf16_f32 -> three bf16 planes in LDS (ds_write_b64);
//   W  split AND tiled once per weight (gdrnpp_pack_weight_bf16x3) into the exact LDS image of every 128x16 tile,
//      [N/128][K/16][split][k-block][row][8] bf16, so the B stage is three lane-linear 16-byte loads and stores;
//   pipeline: two LDS stages and two register stages.  While the matrix pipe works on tile t (24 MFMAs = 768 cycles
//      per wave), the same wave splits tile t+1 (in registers since the previous iteration) into the other LDS stage
//      in the shadow of its own MFMAs, and the loads of tile t+2 are in flight; one barrier per k-tile;
//   workgroup -> tile map is XCD-aware: each XCD walks a contiguous range of tiles (n fastest), so the n-tiles that
//      share A rows hit the same L2;
//   LDS image per operand: [split][k-block of 8][row] 16-byte slots, plane pitch 132 slots: the 32 lanes of a
//      ds_read_b128 lane group read 32 consecutive slots (conflict-free), a fragment is one ds_read_b128;
//   per k-tile: 12 fragment reads (ds_read_b128) feed 24 MFMAs.
#include "common.hpp"
#include <cstdlib>

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using f32x4v = __attribute__((ext_vector_type(4))) float;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

#ifndef SPLIT_OCC
#define SPLIT_OCC 2
#endif

constexpr int BM = 128, BN = 128, BK = 16, KB = BK / 8, PLANE = BM + 4;
constexpr int OPER_SLOTS = 3 * KB * PLANE;  // uint4 slots per operand image
constexpr int W_TILE_SLOTS = 3 * KB * BN;   // uint4 slots of one packed 128x16 weight tile
enum { EPI_BIAS = 0, EPI_GELU = 1, EPI_SCALE_RES = 2 };

// exact-erf GELU (ocml erff).  A 23-instruction fitted erf was tried in its place: no measurable change end to end
// (3071 vs 3077 ROIs/s on one box) — the epilogue's VALU work already overlaps other waves' MFMAs — so it was dropped.
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

// W f32[N][K] -> packed bf16 [N/128][K/16][3][2][128][8]; one thread per (row, k-block) = 8 consecutive k
__global__ void pack_weight_kernel(const float* __restrict__ W, uint4* __restrict__ packed, int N, int K) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int kbs = K / 8;
  if (i >= (long)N * kbs) return;
  const int n = (int)(i / kbs), kb_g = (int)(i % kbs);
  const float4 v0 = *reinterpret_cast<const float4*>(W + (size_t)n * K + kb_g * 8);
  const float4 v1 = *reinterpret_cast<const float4*>(W + (size_t)n * K + kb_g * 8 + 4);
  const Split3 p0 = split_pair(v0.x, v0.y), p1 = split_pair(v0.z, v0.w), p2 = split_pair(v1.x, v1.y), p3 = split_pair(v1.z, v1.w);
  const int tn = n / BN, row = n % BN, tk = kb_g / KB, kb = kb_g % KB;
  uint4* img = packed + ((size_t)tn * (K / BK) + tk) * W_TILE_SLOTS;
  img[(0 * KB + kb) * BN + row] = make_uint4(p0.h, p1.h, p2.h, p3.h);
  img[(1 * KB + kb) * BN + row] = make_uint4(p0.m, p1.m, p2.m, p3.m);
  img[(2 * KB + kb) * BN + row] = make_uint4(p0.l, p1.l, p2.l, p3.l);
}

// CONV: A is an NHWC image [.,H,W,Cin] and the GEMM row m = output pixel, k = (tap, channel) of a KHxKW convolution with
// stride and symmetric zero padding (implicit im2col: the k-tile's 16 channels of one tap are 64 contiguous bytes per
// pixel).  Every row keeps a pointer to its anchor input pixel (oy*stride, ox*stride), which is always inside the image.
// H, W, C: input image; OH, OW: output image; KW x (K / (KW*C)) taps, stride, zero padding `pad` on every side.
// nk_split > 0: split-K (linear only), blockIdx.y-th chunk of nk_split k-tiles -> partial C
struct ConvGeom { int H, W, C, OH, OW, KW, stride, pad; int nk_split; };

// MI = 32-row MFMA tiles per wave along M: 2 -> 128x128 block tile (three workgroups per CU), 4 -> 256x128 block tile
// (each wave 128x64: 18 fragment reads feed 48 MFMAs per k-tile and per barrier, two workgroups per CU).
// CONV: 0 = linear, 1 = 3x3 / stride 1 / pad 1 (geometry folded at compile time: the head's hot convolutions), 2 = general
template <int EPI, int CONV, int MI>
__global__ __launch_bounds__(256, MI == 2 ? SPLIT_OCC : 2) void gemm_split_kernel(const float* __restrict__ A,
                                                                    const uint4* __restrict__ Wp,
                                                                    const float* __restrict__ bias,
                                                                    const float* __restrict__ gamma,
                                                                    const float* __restrict__ resid,
                                                                    float* __restrict__ C, int M, int N, int K,
                                                                    ConvGeom cg) {
  constexpr int BMT = MI * 64, PLANE_A = BMT + 4, A_SLOTS = 3 * KB * PLANE_A;
  // static LDS objects per operand and stage (MI = 4: 75 KB -> two workgroups per CU, MI = 2: 50.7 KB -> three)
  __shared__ uint4 sA[2 * A_SLOTS];
  __shared__ uint4 sB0[OPER_SLOTS];
  __shared__ uint4 sB1[OPER_SLOTS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntn = N / BN;
  // XCD-aware map: hardware deals consecutive workgroup ids round-robin to the 8 XCDs; give each XCD a contiguous
  // range of tile ids (bijective for any grid size)
  const int nwg = gridDim.x, xcd = blockIdx.x & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
  const int tile_m = tile / ntn, tile_n = tile % ntn;
  const int m0 = tile_m * BMT, n0 = tile_n * BN;  // the last m-tile may hang over M: its loads are clamped to row
                                                   // M-1 and its stores masked, so any M >= 1 is accepted
  // split-K (skinny problems: few rows, long K): workgroup (x, y) accumulates k-tiles [y*nk, (y+1)*nk) into its own
  // partial result C + y*M*N; gdrnpp_linear_f32_splitk sums the partials and applies the bias afterwards
  const int nk_total = K / BK;
  const int nk = cg.nk_split > 0 ? cg.nk_split : nk_total;
  const int kt0 = cg.nk_split > 0 ? (int)blockIdx.y * nk : 0;
  if (cg.nk_split > 0) C += (size_t)blockIdx.y * M * N;

  // A staging: rows tid/4 + 64*p (p < MI), lane%4 picks 4 consecutive k (half a k-block)
  const int lrow = tid >> 2, lkq = tid & 3;
  const float* ap[MI];  // this thread's staging rows (pixels), clamped into [0, M): the overhang re-reads row M-1
  int pyx[MI], cpt = 1;  // CONV: (y << 16 | x) of the anchor input pixel of this thread's rows, k-tiles per tap
#pragma unroll
  for (int p = 0; p < MI; ++p) {
    const int arow = min(m0 + lrow + 64 * p, M - 1);
    if (CONV) {
      const int OH = CONV == 1 ? cg.H : cg.OH, OW = CONV == 1 ? cg.W : cg.OW, stride = CONV == 1 ? 1 : cg.stride;
      const int img = arow / (OH * OW), pp = arow - img * (OH * OW);
      const int iy = (pp / OW) * stride, ix = (pp % OW) * stride;
      pyx[p] = (iy << 16) | ix;
      ap[p] = A + (((size_t)img * cg.H + iy) * cg.W + ix) * cg.C + lkq * 4;
    } else {
      pyx[p] = 0;
      ap[p] = A + (size_t)arow * K + lkq * 4;
    }
  }
  if (CONV) cpt = cg.C / BK;
  const uint4* Wg = Wp + ((size_t)tile_n * nk_total + kt0) * W_TILE_SLOTS + tid;
  struct Stage { float4 a[MI]; uint4 b0, b1, b2; };
  auto gload = [&](int kt) {
    Stage r;
    if (CONV) {
      // the loads are unconditional (address clamped to the centre pixel, value zeroed afterwards): a predicated load
      // would make the outstanding-load count unknown to the compiler and collapse the software pipeline
      const int tap = kt / cpt, c0 = (kt - tap * cpt) * BK;
      const int KW = CONV == 1 ? 3 : cg.KW, pad = CONV == 1 ? 1 : cg.pad;
      const int ky = tap / KW, dy = ky - pad, dx = tap - ky * KW - pad;
      const int off = (dy * cg.W + dx) * cg.C;
#pragma unroll
      for (int p = 0; p < MI; ++p) {
        const bool ok = (unsigned)((pyx[p] >> 16) + dy) < (unsigned)cg.H && (unsigned)((pyx[p] & 0xffff) + dx) < (unsigned)cg.W;
        const float4 v = *reinterpret_cast<const float4*>(ap[p] + (ok ? off : 0) + c0);
        r.a[p] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    } else {
#pragma unroll
      for (int p = 0; p < MI; ++p) r.a[p] = *reinterpret_cast<const float4*>(ap[p] + (kt0 + kt) * BK);
    }
    const uint4* w = Wg + (size_t)kt * W_TILE_SLOTS;
    r.b0 = w[0]; r.b1 = w[256]; r.b2 = w[512];
    return r;
  };
  const int skb = lkq >> 1, shalf = lkq & 1;
  auto lstore = [&](const Stage r, int buf) {
    uint4* a = sA + buf * A_SLOTS;
    uint4* b = buf ? sB1 : sB0;
#pragma unroll
    for (int p = 0; p < MI; ++p) {
      const Split3 p0 = split_pair(r.a[p].x, r.a[p].y), p1 = split_pair(r.a[p].z, r.a[p].w);
      uint2* dst = reinterpret_cast<uint2*>(a + skb * PLANE_A + 64 * p + lrow) + shalf;
      dst[0] = make_uint2(p0.h, p1.h);
      dst[2 * KB * PLANE_A] = make_uint2(p0.m, p1.m);
      dst[4 * KB * PLANE_A] = make_uint2(p0.l, p1.l);
    }
    // image slot i*256 + tid = plane (i*2 + tid/128), row tid%128
    uint4* bd = b + (tid >> 7) * PLANE + (tid & 127);
    bd[0] = r.b0; bd[2 * PLANE] = r.b1; bd[4 * PLANE] = r.b2;
  };

  f32x16 acc[MI][2];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frow = lane & 31, fk = lane >> 5;
  // one k-tile: 3*(MI+2) fragment reads, 12*MI MFMAs; `mid` runs after the first third of the MFMAs is queued (the
  // split + LDS stores of the next tile, hidden behind the matrix pipe)
  auto compute = [&](int buf, auto&& mid) {
    const uint4* a = sA + buf * A_SLOTS + fk * PLANE_A + wm * (MI * 32) + frow;
    const uint4* b = (buf ? sB1 : sB0) + fk * PLANE + wn * 64 + frow;
    bf16x8 fa[3][MI], fb[3][2];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
#pragma unroll
      for (int i = 0; i < MI; ++i) fa[s][i] = __builtin_bit_cast(bf16x8, a[s * KB * PLANE_A + i * 32]);
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[s][j] = __builtin_bit_cast(bf16x8, b[s * KB * PLANE + j * 32]);
    }
    // smallest partial products first; the accumulators rotate so no MFMA waits on its predecessor
    constexpr int TA[6] = {2, 0, 1, 1, 0, 0};
    constexpr int TB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int t = 0; t < 6; ++t) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[TA[t]][i], fb[TB[t]][j], acc[i][j], 0, 0, 0);
      if (t == 1) mid();
    }
  };

  // nk is even (K % 32 == 0): the loop is unrolled by two so the register stages r0 / r1 stay in fixed registers.
  // Loads and stores are UNCONDITIONAL (the tile index is clamped, the tail re-stages the last tile into the idle
  // buffer): a conditional load would force the compiler to drain vmcnt to 0 in the middle of every iteration.
  Stage r0 = gload(0);
  lstore(r0, 0);
  Stage r1 = gload(1);
  __syncthreads();
  for (int kt = 0; kt < nk; kt += 2) {
    r0 = gload(min(kt + 2, nk - 1));
    compute(0, [&] { lstore(r1, 1); });
    __syncthreads();
    r1 = gload(min(kt + 3, nk - 1));
    compute(1, [&] { lstore(r0, 0); });
    __syncthreads();
  }

  // epilogue: lane holds column (lane & 31) of rows (r&3) + 8*(r>>2) + 4*(lane>>5).  Each wave parks one 16x64 slice
  // of its tile in LDS (the A images are dead: the loop ends on a barrier) and writes it back row-wise as float4.
  static_assert(2 * A_SLOTS * sizeof(uint4) >= 4 * 16 * 65 * sizeof(float), "epilogue staging fits the A images");
  float* T = reinterpret_cast<float*>(sA) + wave * 16 * 65;  // [16][65] per wave
  const int c4 = (lane & 15) * 4;
  const int nb = n0 + wn * 64 + c4;
  const float4 bv = bias ? *reinterpret_cast<const float4*>(bias + nb) : make_float4(0.f, 0.f, 0.f, 0.f);
  float4 gv = make_float4(1.f, 1.f, 1.f, 1.f);
  if (EPI == EPI_SCALE_RES) gv = *reinterpret_cast<const float4*>(gamma + nb);
#pragma unroll
  for (int ih = 0; ih < 2 * MI; ++ih) {
    const int i = ih >> 1, h = ih & 1;  // 32-row MFMA tile i, its 16-row half h (accumulator registers h*8 .. h*8+7)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 8; ++r)
        T[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 65 + j * 32 + (lane & 31)] = acc[i][j][h * 8 + r];
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): same wave reads back
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int row = rr * 4 + (lane >> 4);
      const float* t = T + row * 65 + c4;
      float4 v = make_float4(t[0] + bv.x, t[1] + bv.y, t[2] + bv.z, t[3] + bv.w);
      const int grow = m0 + wm * (MI * 32) + i * 32 + h * 16 + row;
      if (grow >= M) continue;  // overhang of the last m-tile
      const size_t off = (size_t)grow * N + nb;
      if (EPI == EPI_GELU) { v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w); }
      if (EPI == EPI_SCALE_RES) {
        const float4 rs = *reinterpret_cast<const float4*>(resid + off);
        v.x = rs.x + gv.x * v.x; v.y = rs.y + gv.y * v.y; v.z = rs.z + gv.z * v.z; v.w = rs.w + gv.w * v.w;
      }
      // streaming store: the result is not read again by this kernel, keep it from displacing the weights in L2
      // (-4 % on the fc1 shapes, whose output is 4x their input)
      { const f32x4v t4 = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(t4, reinterpret_cast<f32x4v*>(C + off)); }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);  // reads done before the next slice overwrites T
  }
}

// ----------------------------------------------------------------------------------------------------------------
// EXPERIMENTAL (opt-in, see launch_split_epi): 256x128 tile, 8 waves, ping-pong schedule.
// The 4-wave kernel above keeps the matrix pipe ~60 % busy (PMC:
// SQ_VALU_MFMA_BUSY_CYCLES / elapsed): its waves sit in barriers and LDS latencies at the same moments.  Here the two
// waves that share a SIMD (wave w and w+4 of the workgroup) alternate roles every phase:
//   phase 1: group 0 (waves 0-3, output rows 0-127) issues its 24 MFMAs of k-tile t from fragments already in
//            registers; group 1 splits + stores its share of k-tile t+1 into the other LDS stage, requests its share of
//            k-tile t+2 from HBM and pre-reads its own fragments of k-tile t;
//   phase 2: the roles swap (group 1: MFMAs of k-tile t on output rows 128-255; group 0: stage / request / pre-read).
// Group 1 stages A rows 0-127 and the whole B tile — exactly what group 0 consumes next — and group 0 stages A rows
// 128-255, so every fragment pre-read only depends on stores finished one barrier earlier and an MFMA phase starts
// with all its operands in registers.  One workgroup per CU (512 threads, up to 256 VGPRs per wave).
// ----------------------------------------------------------------------------------------------------------------
constexpr int BM8 = 256, PLANE8 = BM8 + 4;
constexpr int A8_SLOTS = 3 * KB * PLANE8;          // uint4 slots of the A image of one stage
constexpr int STAGE8_SLOTS = A8_SLOTS + OPER_SLOTS;  // + B image

template <int EPI, bool CONV>
__global__ __launch_bounds__(512) void gemm_split_kernel8(const float* __restrict__ A, const uint4* __restrict__ Wp,
                                                          const float* __restrict__ bias,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ resid, float* __restrict__ C,
                                                          int M, int N, int K, ConvGeom cg) {
  extern __shared__ uint4 lds4[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // Roles follow the hardware placement, not the wave index: the two waves that the dispatcher put on the same SIMD
  // must land in different groups, or one SIMD gets both MFMA phases and its neighbour none.  Every wave publishes
  // its SIMD id (HW_ID[5:4]); group = rank among the waves of that SIMD, quadrant = SIMD id.  If the placement is not
  // two-per-SIMD the wave index decides.
  __shared__ int s_simd[8];
  const int my_simd = (int)__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4) & 3;  // HW_REG_HW_ID, offset 4, 2 bits
  if (lane == 0) s_simd[wave] = my_simd;
  __syncthreads();
  int grp = 0, cnt = 0, ok_placement = 1;
#pragma unroll
  for (int w = 0; w < 8; ++w) {
    int c = 0;
#pragma unroll
    for (int v = 0; v < 8; ++v) c += s_simd[v] == s_simd[w];
    ok_placement &= c == 2;
    if (s_simd[w] == my_simd) { grp += w < wave; ++cnt; }
  }
  (void)cnt;
  int wq = my_simd;
  if (!ok_placement) { grp = wave >> 2; wq = wave & 3; }
  grp = __builtin_amdgcn_readfirstlane(grp);
  wq = __builtin_amdgcn_readfirstlane(wq);
  const int wm = wq >> 1, wn = wq & 1, tig = wq * 64 + lane;  // staging index inside the group's 256 threads
  const int ntn = N / BN;
  const int nwg = gridDim.x, xcd = blockIdx.x & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
  const int tile_m = tile / ntn, tile_n = tile % ntn;
  const int m0 = tile_m * BM8, n0 = tile_n * BN;
  const int nk = K / BK;

  // staging share of this thread: group 1 -> A rows 0..127 (+ B), group 0 -> A rows 128..255
  const int arow = (grp == 1 ? 0 : 128) + (tig >> 2), lkq = tig & 3;
  const float* Ag = CONV ? A + (size_t)(m0 + arow) * cg.C + lkq * 4 : A + (size_t)(m0 + arow) * K + lkq * 4;
  const uint4* Wg = Wp + (size_t)tile_n * nk * W_TILE_SLOTS + tig;
  int py0 = 0, px0 = 0, py1 = 0, px1 = 0, cpt = 1;
  if (CONV) {
    const int p0 = (m0 + arow) % (cg.H * cg.W), p1 = (m0 + arow + 64) % (cg.H * cg.W);
    py0 = p0 / cg.W; px0 = p0 % cg.W; py1 = p1 / cg.W; px1 = p1 % cg.W;
    cpt = cg.C / BK;
  }
  struct StageA { float4 a0, a1; };
  struct StageB { uint4 b0, b1, b2; };
  auto gload_a = [&](int kt) {
    StageA r;
    if (CONV) {
      const int tap = kt / cpt, c0 = (kt - tap * cpt) * BK;
      const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
      const bool ok0 = (unsigned)(py0 + dy) < (unsigned)cg.H && (unsigned)(px0 + dx) < (unsigned)cg.W;
      const bool ok1 = (unsigned)(py1 + dy) < (unsigned)cg.H && (unsigned)(px1 + dx) < (unsigned)cg.W;
      const int off = (dy * cg.W + dx) * cg.C;
      const float4 v0 = *reinterpret_cast<const float4*>(Ag + (ok0 ? off : 0) + c0);
      const float4 v1 = *reinterpret_cast<const float4*>(Ag + (size_t)64 * cg.C + (ok1 ? off : 0) + c0);
      r.a0 = ok0 ? v0 : make_float4(0.f, 0.f, 0.f, 0.f);
      r.a1 = ok1 ? v1 : make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      r.a0 = *reinterpret_cast<const float4*>(Ag + kt * BK);
      r.a1 = *reinterpret_cast<const float4*>(Ag + (size_t)64 * K + kt * BK);
    }
    return r;
  };
  auto gload_b = [&](int kt) {
    StageB r;
    const uint4* w = Wg + (size_t)kt * W_TILE_SLOTS;
    r.b0 = w[0]; r.b1 = w[256]; r.b2 = w[512];
    return r;
  };
  const int skb = lkq >> 1, shalf = lkq & 1;
  auto lstore_a = [&](const StageA r, int buf) {
    uint4* a = lds4 + buf * STAGE8_SLOTS;
    {
      const Split3 p0 = split_pair(r.a0.x, r.a0.y), p1 = split_pair(r.a0.z, r.a0.w);
      uint2* dst = reinterpret_cast<uint2*>(a + skb * PLANE8 + arow) + shalf;
      dst[0] = make_uint2(p0.h, p1.h);
      dst[2 * KB * PLANE8] = make_uint2(p0.m, p1.m);
      dst[4 * KB * PLANE8] = make_uint2(p0.l, p1.l);
    }
    {
      const Split3 p0 = split_pair(r.a1.x, r.a1.y), p1 = split_pair(r.a1.z, r.a1.w);
      uint2* dst = reinterpret_cast<uint2*>(a + skb * PLANE8 + 64 + arow) + shalf;
      dst[0] = make_uint2(p0.h, p1.h);
      dst[2 * KB * PLANE8] = make_uint2(p0.m, p1.m);
      dst[4 * KB * PLANE8] = make_uint2(p0.l, p1.l);
    }
  };
  auto lstore_b = [&](const StageB r, int buf) {
    uint4* bd = lds4 + buf * STAGE8_SLOTS + A8_SLOTS + (tig >> 7) * PLANE + (tig & 127);
    bd[0] = r.b0; bd[2 * PLANE] = r.b1; bd[4 * PLANE] = r.b2;
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frow = lane & 31, fk = lane >> 5;
  bf16x8 fa[3][2], fb[3][2];
  auto read_frags = [&](int buf) {
    const uint4* a = lds4 + buf * STAGE8_SLOTS + fk * PLANE8 + grp * 128 + wm * 64 + frow;
    const uint4* b = lds4 + buf * STAGE8_SLOTS + A8_SLOTS + fk * PLANE + wn * 64 + frow;
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        fa[s][i] = __builtin_bit_cast(bf16x8, a[s * KB * PLANE8 + i * 32]);
        fb[s][i] = __builtin_bit_cast(bf16x8, b[s * KB * PLANE + i * 32]);
      }
  };
  auto mfma24 = [&]() {
    constexpr int TA[6] = {2, 0, 1, 1, 0, 0};
    constexpr int TB[6] = {0, 2, 1, 0, 1, 0};
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int t = 0; t < 6; ++t) {
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[TA[t]][0], fb[TB[t]][0], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[TA[t]][0], fb[TB[t]][1], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[TA[t]][1], fb[TB[t]][0], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[TA[t]][1], fb[TB[t]][1], acc[1][1], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
  };

  // The two groups run separate loops (same barrier count) so that neither carries the other's registers or waits:
  // every s_barrier below is executed once per phase by all eight waves.  Global loads run TWO k-tiles (four phases)
  // ahead of their LDS store through two register stages; the loop is unrolled by two (nk is even) so each stage
  // keeps its registers.  With one workgroup per CU nothing else hides HBM latency: at a distance of one k-tile the
  // loop ran at exactly the memory latency (0.8 us per k-tile).
#define KCLAMP(kt) min((kt), nk - 1)
#define G0_STEP(ST, KT)                                                                               \
  {                                                                                                   \
    const int buf = (KT)&1;                                                                           \
    mfma24();              /* phase 1: k-tile KT, output rows 0-127 */                                \
    __syncthreads();                                                                                  \
    lstore_a(ST, buf ^ 1); /* phase 2: A rows 128-255 of k-tile KT+1 */                               \
    ST = gload_a(KCLAMP((KT) + 3));                                                                   \
    read_frags(buf ^ 1);   /* own fragments of k-tile KT+1 (group 1 stored them in phase 1) */        \
    __syncthreads();                                                                                  \
  }
#define G1_STEP(ST, TT, KT)                                                                           \
  {                                                                                                   \
    const int buf = (KT)&1;                                                                           \
    lstore_a(ST, buf ^ 1); /* phase 1: A rows 0-127 + B of k-tile KT+1 */                             \
    lstore_b(TT, buf ^ 1);                                                                            \
    ST = gload_a(KCLAMP((KT) + 3));                                                                   \
    TT = gload_b(KCLAMP((KT) + 3));                                                                   \
    read_frags(buf);       /* own fragments of k-tile KT (rows 128-255 landed one barrier ago) */     \
    __syncthreads();                                                                                  \
    mfma24();              /* phase 2: k-tile KT, output rows 128-255 */                              \
    __syncthreads();                                                                                  \
  }
  if (grp == 0) {
    StageA s0 = gload_a(0);
    lstore_a(s0, 0);
    StageA s1 = gload_a(KCLAMP(1));
    s0 = gload_a(KCLAMP(2));
    __syncthreads();
    read_frags(0);
    for (int kt = 0; kt < nk; kt += 2) {
      G0_STEP(s1, kt)
      G0_STEP(s0, kt + 1)
    }
  } else {
    StageA s0 = gload_a(0);
    StageB t0 = gload_b(0);
    lstore_a(s0, 0);
    lstore_b(t0, 0);
    StageA s1 = gload_a(KCLAMP(1));
    StageB t1 = gload_b(KCLAMP(1));
    s0 = gload_a(KCLAMP(2));
    t0 = gload_b(KCLAMP(2));
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
      G1_STEP(s1, t1, kt)
      G1_STEP(s0, t0, kt + 1)
    }
  }
#undef G0_STEP
#undef G1_STEP
#undef KCLAMP

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
      for (int rr = 0; rr < 16; ++rr)
        T[((rr & 3) + 8 * (rr >> 2) + 4 * (lane >> 5)) * 65 + j * 32 + (lane & 31)] = acc[i][j][rr];
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): same wave reads back
#pragma unroll 4
    for (int rr = 0; rr < 8; ++rr) {
      const int row = rr * 4 + (lane >> 4);
      const float* t = T + row * 65 + c4;
      float4 v = make_float4(t[0] + bv.x, t[1] + bv.y, t[2] + bv.z, t[3] + bv.w);
      const size_t off = (size_t)(m0 + grp * 128 + wm * 64 + i * 32 + row) * N + nb;
      if (EPI == EPI_GELU) { v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w); }
      if (EPI == EPI_SCALE_RES) {
        const float4 rs = *reinterpret_cast<const float4*>(resid + off);
        v.x = rs.x + gv.x * v.x; v.y = rs.y + gv.y * v.y; v.z = rs.z + gv.z * v.z; v.w = rs.w + gv.w * v.w;
      }
      *reinterpret_cast<float4*>(C + off) = v;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
  }
}

}  // namespace

extern "C" int gdrnpp_pack_weight_bf16x3(const float* W, void* packed, int N, int K, void* stream) {
  GDRNPP_REQUIRE(W && packed, GDRNPP_EINVAL, "gdrnpp_pack_weight_bf16x3: null pointer");
  GDRNPP_REQUIRE(N > 0 && K > 0 && N % BN == 0 && K % 32 == 0, GDRNPP_ELIMIT,
                 "gdrnpp_pack_weight_bf16x3: N=%d K=%d must be multiples of %d/32", N, K, BN);
  const long threads = (long)N * (K / 8);
  hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W,
                     (uint4*)packed, N, K);
  return gdrnpp::check_launch("gdrnpp_pack_weight_bf16x3");
}

namespace {

template <int EPI, bool CONV>
int launch_split_epi(const float* A, const uint4* Wp, const float* bias, const float* gamma, const float* resid, float* C,
                     int M, int N, int K, ConvGeom cg, hipStream_t st, const char* what) {
  // The 256x128 ping-pong kernel is opt-in (GDRNPP_SPLIT_8WAVE=1, read per launch so tests can toggle it): measured on
  // MI355X it reaches 133-174 TFLOP/s fp32-equivalent on the stage-2 MLP shapes against 144-177 for the 4-wave kernel
  // at three workgroups per CU (tools/microbench_gemm_s2.py, same box) — see DESIGN.md §5.
  const bool fast3x3 = CONV && cg.KW == 3 && cg.stride == 1 && cg.pad == 1 && K == 9 * cg.C;
  const char* use8 = getenv("GDRNPP_SPLIT_8WAVE");
  if (M % BM8 == 0 && use8 && use8[0] == '1' && (!CONV || fast3x3)) {
    const long blocks = (long)(M / BM8) * (N / BN);
    GDRNPP_REQUIRE(blocks < (1l << 31), GDRNPP_ELIMIT, "%s: grid too large", what);
    const int lds = 2 * STAGE8_SLOTS * (int)sizeof(uint4);  // 75 KB >= 8 * 32 * 65 * 4 (epilogue staging)
    GDRNPP_HIP_TRY(hipFuncSetAttribute((const void*)gemm_split_kernel8<EPI, CONV>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL((gemm_split_kernel8<EPI, CONV>), dim3((unsigned)blocks), dim3(512), lds, st, A, Wp, bias, gamma, resid, C, M, N, K, cg);
    return gdrnpp::check_launch(what);
  }
  // 256x128 tiles (MI = 4) when they still give every CU its two workgroups; measured +3 % (fc1) / +7 % (fc2) on the
  // stage-2 MLP shapes over 128x128 tiles at three workgroups per CU.  GDRNPP_SPLIT_MI4=0/1 forces the choice (A/B).
  const char* mi4 = getenv("GDRNPP_SPLIT_MI4");
  // (the general convolution form needs a few more registers than 256x128 tiles leave: it stays on 128x128 tiles)
  const bool big = (mi4 ? mi4[0] == '1' : (long)(M / 256) * (N / BN) >= 512) && !(CONV && !fast3x3);
  if (M % 256 == 0 && big) {
    const long blocks = (long)(M / 256) * (N / BN);
    if (CONV && fast3x3) hipLaunchKernelGGL((gemm_split_kernel<EPI, CONV ? 1 : 0, 4>), dim3((unsigned)blocks), dim3(256), 0, st, A, Wp, bias, gamma, resid, C, M, N, K, cg);
    else hipLaunchKernelGGL((gemm_split_kernel<EPI, CONV ? 2 : 0, 4>), dim3((unsigned)blocks), dim3(256), 0, st, A, Wp, bias, gamma, resid, C, M, N, K, cg);
    return gdrnpp::check_launch(what);
  }
  const long blocks = (long)((M + BM - 1) / BM) * (N / BN);
  GDRNPP_REQUIRE(blocks < (1l << 31), GDRNPP_ELIMIT, "%s: grid too large", what);
  if (CONV && fast3x3) hipLaunchKernelGGL((gemm_split_kernel<EPI, CONV ? 1 : 0, 2>), dim3((unsigned)blocks), dim3(256), 0, st, A, Wp, bias, gamma, resid, C, M, N, K, cg);
  else hipLaunchKernelGGL((gemm_split_kernel<EPI, CONV ? 2 : 0, 2>), dim3((unsigned)blocks), dim3(256), 0, st, A, Wp, bias, gamma, resid, C, M, N, K, cg);
  return gdrnpp::check_launch(what);
}

template <bool CONV>
int launch_split(const float* A, const uint4* Wp, const float* bias, const float* gamma, const float* resid, float* C,
                 int M, int N, int K, int epilogue, ConvGeom cg, hipStream_t st, const char* what) {
  if (epilogue == EPI_BIAS) return launch_split_epi<EPI_BIAS, CONV>(A, Wp, bias, gamma, resid, C, M, N, K, cg, st, what);
  if (epilogue == EPI_GELU) return launch_split_epi<EPI_GELU, CONV>(A, Wp, bias, gamma, resid, C, M, N, K, cg, st, what);
  return launch_split_epi<EPI_SCALE_RES, CONV>(A, Wp, bias, gamma, resid, C, M, N, K, cg, st, what);
}

}  // namespace

namespace {

// out = epilogue(bias[n] + sum over s of part[s][m][n])   (fixed order: deterministic)
template <int EPI>
__global__ void splitk_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                     const float* __restrict__ gamma, const float* __restrict__ resid, float* __restrict__ out,
                                     long mn4, int N, int splits) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= mn4) return;
  const int n = (int)((i * 4) % N);
  float4 acc = bias ? *reinterpret_cast<const float4*>(bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
  for (int s = 0; s < splits; ++s) {
    const float4 v = reinterpret_cast<const float4*>(part)[(size_t)s * mn4 + i];
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  if (EPI == EPI_GELU) { acc.x = gelu_erf(acc.x); acc.y = gelu_erf(acc.y); acc.z = gelu_erf(acc.z); acc.w = gelu_erf(acc.w); }
  if (EPI == EPI_SCALE_RES) {
    const float4 g = *reinterpret_cast<const float4*>(gamma + n), r = reinterpret_cast<const float4*>(resid)[i];
    acc.x = r.x + g.x * acc.x; acc.y = r.y + g.y * acc.y; acc.z = r.z + g.z * acc.z; acc.w = r.w + g.w * acc.w;
  }
  reinterpret_cast<float4*>(out)[i] = acc;
}

// k-tiles (of 16) per split: halve the chunk (keeping it even) until every CU has about three workgroups
inline int splitk_chunk(int M, int N, int K) {
  const long tiles = (long)((M + BM - 1) / BM) * (N / BN);
  const int nk = K / BK;
  int c = nk;
  while (c % 4 == 0 && tiles * (nk / c) < 768) c /= 2;
  return c;
}

}  // namespace

extern "C" size_t gdrnpp_linear_f32_splitk_workspace_bytes(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0 || K % 32 || N % BN) return 0;
  return (size_t)((K / BK) / splitk_chunk(M, N, K)) * M * N * sizeof(float);
}

extern "C" int gdrnpp_linear_f32_splitk(const float* A, const void* W_packed, const float* bias, const float* gamma,
                                        const float* resid, float* C, int M, int N, int K, int epilogue, void* workspace,
                                        size_t workspace_bytes, void* stream) {
  GDRNPP_REQUIRE(A && W_packed && C && workspace, GDRNPP_EINVAL, "gdrnpp_linear_f32_splitk: null pointer");
  GDRNPP_REQUIRE(M > 0 && N > 0 && K > 0 && N % BN == 0 && K % 32 == 0, GDRNPP_ELIMIT,
                 "gdrnpp_linear_f32_splitk: N=%d K=%d must be multiples of %d/32 (M=%d is free)", N, K, BN, M);
  GDRNPP_REQUIRE(epilogue >= 0 && epilogue <= 2, GDRNPP_EINVAL, "gdrnpp_linear_f32_splitk: epilogue=%d", epilogue);
  GDRNPP_REQUIRE(epilogue != EPI_SCALE_RES || (gamma && resid), GDRNPP_EINVAL,
                 "gdrnpp_linear_f32_splitk: scale+residual epilogue needs gamma and resid");
  GDRNPP_REQUIRE(workspace_bytes >= gdrnpp_linear_f32_splitk_workspace_bytes(M, N, K), GDRNPP_EINVAL,
                 "gdrnpp_linear_f32_splitk: workspace too small");
  const int nkc = splitk_chunk(M, N, K);
  const int splits = (K / BK) / nkc;
  const long tiles = (long)((M + BM - 1) / BM) * (N / BN);
  GDRNPP_REQUIRE(tiles < (1l << 31) && splits < 65536, GDRNPP_ELIMIT, "gdrnpp_linear_f32_splitk: grid too large");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL((gemm_split_kernel<EPI_BIAS, 0, 2>), dim3((unsigned)tiles, (unsigned)splits), dim3(256), 0, st, A,
                     (const uint4*)W_packed, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                     (float*)workspace, M, N, K, ConvGeom{0, 0, 0, 0, 0, 0, 0, 0, nkc});
  const long mn4 = (long)M * N / 4;
  const dim3 grid((unsigned)((mn4 + 255) / 256));
  const float* ws = (const float*)workspace;
  if (epilogue == EPI_BIAS) hipLaunchKernelGGL(splitk_reduce_kernel<EPI_BIAS>, grid, dim3(256), 0, st, ws, bias, gamma, resid, C, mn4, N, splits);
  else if (epilogue == EPI_GELU) hipLaunchKernelGGL(splitk_reduce_kernel<EPI_GELU>, grid, dim3(256), 0, st, ws, bias, gamma, resid, C, mn4, N, splits);
  else hipLaunchKernelGGL(splitk_reduce_kernel<EPI_SCALE_RES>, grid, dim3(256), 0, st, ws, bias, gamma, resid, C, mn4, N, splits);
  return gdrnpp::check_launch("gdrnpp_linear_f32_splitk");
}

extern "C" int gdrnpp_linear_f32_split(const float* A, const void* W_packed, const float* bias, const float* gamma,
                                       const float* resid, float* C, int M, int N, int K, int epilogue,
                                       void* stream) {
  GDRNPP_REQUIRE(A && W_packed && C, GDRNPP_EINVAL, "gdrnpp_linear_f32_split: null pointer");
  GDRNPP_REQUIRE(M > 0 && N > 0 && K > 0 && N % BN == 0 && K % 32 == 0, GDRNPP_ELIMIT,
                 "gdrnpp_linear_f32_split: N=%d K=%d must be multiples of %d/32 (M=%d is free)", N, K, BN, M);
  GDRNPP_REQUIRE(epilogue >= 0 && epilogue <= 2, GDRNPP_EINVAL, "gdrnpp_linear_f32_split: epilogue=%d", epilogue);
  GDRNPP_REQUIRE(epilogue != EPI_SCALE_RES || (gamma && resid), GDRNPP_EINVAL,
                 "gdrnpp_linear_f32_split: scale+residual epilogue needs gamma and resid");
  return launch_split<false>(A, (const uint4*)W_packed, bias, gamma, resid, C, M, N, K, epilogue, ConvGeom{0, 0, 0, 0, 0, 0, 0, 0, 0},
                             (hipStream_t)stream, "gdrnpp_linear_f32_split");
}

extern "C" int gdrnpp_conv2d_f32_split(const float* x_nhwc, const void* W_packed, const float* bias, float* y_nhwc,
                                       int n_img, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                                       int epilogue, void* stream) {
  GDRNPP_REQUIRE(x_nhwc && W_packed && y_nhwc, GDRNPP_EINVAL, "gdrnpp_conv2d_f32_split: null pointer");
  GDRNPP_REQUIRE(n_img > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && H < 32768 && W < 32768 && KH > 0 && KW > 0 &&
                     stride > 0 && pad >= 0 && pad < KH && pad < KW,
                 GDRNPP_EINVAL, "gdrnpp_conv2d_f32_split: bad shape");
  const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
  GDRNPP_REQUIRE(OH > 0 && OW > 0 && (OH - 1) * stride < H && (OW - 1) * stride < W, GDRNPP_EINVAL,
                 "gdrnpp_conv2d_f32_split: empty output or anchor pixel outside the image");
  const long M = (long)n_img * OH * OW;
  GDRNPP_REQUIRE(M < (1l << 31) && Cout % BN == 0 && Cin % 32 == 0, GDRNPP_ELIMIT,
                 "gdrnpp_conv2d_f32_split: Cout=%d Cin=%d must be multiples of %d/32 (pixels=%ld is free)", Cout, Cin, BN, M);
  GDRNPP_REQUIRE(epilogue == EPI_BIAS || epilogue == EPI_GELU, GDRNPP_EINVAL, "gdrnpp_conv2d_f32_split: epilogue=%d", epilogue);
  return launch_split<true>(x_nhwc, (const uint4*)W_packed, bias, nullptr, nullptr, y_nhwc, (int)M, Cout, KH * KW * Cin, epilogue,
                            ConvGeom{H, W, Cin, OH, OW, KW, stride, pad, 0}, (hipStream_t)stream, "gdrnpp_conv2d_f32_split");
}

extern "C" int gdrnpp_conv3x3_f32_split(const float* x_nhwc, const void* W_packed, const float* bias, float* y_nhwc,
                                        int n_img, int H, int W, int Cin, int Cout, int epilogue, void* stream) {
  return gdrnpp_conv2d_f32_split(x_nhwc, W_packed, bias, y_nhwc, n_img, H, W, Cin, Cout, 3, 3, 1, 1, epilogue, stream);
}
