// Fused ConvNeXt MLP in the three-product (fp16x2) form for the two shallow stages (C = 128 / 256, hidden 4 C) — SURVEY.md §8 row a3:
//     y = resid + gamma * (fc2(gelu(fc1(x))))          timm ConvNeXtBlock tail, one launch, the hidden tensor never leaves the CU.
//
// Why: at 128 ROIs the two stage-0 launches move 2.95 GB through HBM per block (1.07 GB of hidden tensor written and read back)
// for 0.81 GB of input, residual and output, and run at the HBM rate, not the matrix rate.
//
// How — everything is computed TRANSPOSED, so that no operand ever has to change its register layout:
//     H^T[hidden, pixel] = W1[hidden, c] . X^T[c, pixel]            MFMA A operand = weight tile, B operand = 32 pixels of x
//     Y^T[out, pixel]    = W2[out, hidden] . gelu(H^T + b1)[hidden, pixel]
// * a wave owns 32 pixels for the whole kernel.  Its x rows (NHWC: 8 consecutive channels of a pixel = 32 contiguous bytes = the B
//   operand of one lane) are loaded ONCE, split ONCE into fp16 h + l (64 VGPRs for C = 128, 128 for C = 256) and stay in registers;
// * the 32 x 32 accumulator tile of H^T holds, per lane, 16 hidden values OF THE LANE'S OWN PIXEL: after bias + GELU + split they
//   ARE the B operand of the second GEMM (two k-steps of 8 values per lane) — no transpose, no LDS round trip.  The k order inside
//   a 32-wide hidden tile is whatever the accumulator layout dictates (hidden = 32 t + (q & 3) + 8 (2 s + (q >> 2)) + 4 kb for k-step
//   s, k-block kb, slot q); W2 is PACKED in that order (gdrnpp_pack_mlp_fused_f16x2), so both operands agree;
// * Y^T accumulates in C / 32 x 16 registers over the hidden tiles; the epilogue applies bias, layer scale and the residual and
//   stores whole rows (through LDS: a lane holds 4 consecutive output channels of ITS pixel per register quad);
// * weights stream through LDS: per hidden tile one 32 KB image (W1 rows of the tile for all 8 k-steps + the W2 columns of the
//   tile for the 4 output tiles, h and l planes, lane-linear 1 KB fragment blocks) by LDS-DMA into two stages — ONE barrier per
//   hidden tile instead of one per 16-wide k-tile.  The 4 waves of a workgroup share every fragment.  C = 256: 64 KB per tile,
//   128 KB of ring, 464 registers per lane: one workgroup per CU, one wave per SIMD — the hand-placed pipeline below is what
//   overlaps its VALU work with its MFMAs.
// Per lane and hidden tile: 48 MFMAs, 32 ds_read_b128, ~270 VALU (the GELU is 13 of them per hidden value).
//
// Numerics: the three products of gemm_split2_pipe.hip (l_w.h_x, h_w.l_x, h_w.h_x per k-step, small terms first, fp32
// accumulation, the weight scale 2^-e applied to the accumulator, bias, gelu_erf, the same epilogue expressions).  The first GEMM sums
// its even and odd k-steps in two accumulators, the second runs the k of each group of 32 hidden units in accumulator order: the
// result equals the two launches to fp32 rounding, not bit for bit (tests/test_gpu_mlp_fused.py: both against fp64).  Range: both A sides of the two GEMMs are judged per pixel like the
// two launches do (x row: once, at load; hidden row: while it is split) and reported in TWO range words (fc1's, fc2's).
#include "split2_common.hpp"

#ifndef GDRNPP_MLPF_ABL
#define GDRNPP_MLPF_ABL 0   // timing-only ablations (tools/build_variant.sh).  Plain loop: 1 no GELU, 2 no DMA / barrier, 4 no second GEMM, 8 no first GEMM, 16 no tile loop.  Pipelined loop: 32 first GEMM on four accumulators, 64 no VALU work
#endif

namespace {

using namespace gdrnpp::split2;

template <int FC, int FH>
struct FusedGeom {
  static constexpr int NK1 = FC / 16;                 // k-steps of the first GEMM
  static constexpr int NT = FH / 32;                  // hidden tiles
  static constexpr int NOT = FC / 32;                 // output tiles of the second GEMM
  static constexpr int W1_SLOTS = NK1 * 2 * 64;       // uint4 slots of a tile's W1 part: [ks][plane][64 lanes]
  static constexpr int W2_SLOTS = 2 * NOT * 2 * 64;   // ... of its W2 part: [s][ot][plane][64 lanes]
  static constexpr int TILE_SLOTS = W1_SLOTS + W2_SLOTS;
  static constexpr int PIECES = TILE_SLOTS / 64;      // 1 KB DMA pieces per tile
};

// 8 consecutive floats -> fp16 h / l fragments (4 + 4 registers) and the running sum of squares of h
__device__ __forceinline__ void split8(const float (&x)[8], unsigned (&h)[4], unsigned (&l)[4], float& ss) {
#pragma unroll
  for (int p = 0; p < 4; ++p) h[p] = cvt_pk_f16(x[2 * p], x[2 * p + 1]);
#pragma unroll
  for (int p = 0; p < 4; ++p) l[p] = cvt_pk_f16(residual<0>(x[2 * p], h[p]), residual<1>(x[2 * p + 1], h[p]));
#pragma unroll
  for (int p = 0; p < 4; ++p) ss = sumsq2(h[p], ss);
}
__device__ __forceinline__ f16x8 frag(const unsigned (&s)[4]) { return __builtin_bit_cast(f16x8, make_uint4(s[0], s[1], s[2], s[3])); }

// WAVES: waves per workgroup (32 pixels each) = 4: two independent workgroups per CU (66 KB of LDS each): while one is in its
// prologue / epilogue (memory) the other's MFMAs use the matrix pipe.  (One 8-wave workgroup per CU measured 4-5 % slower.)
// PIPE: software pipeline over the hidden tiles — the first GEMM of tile t + 1 (independent of everything tile t still has to do)
// is issued together with the bias + GELU + split of tile t, and the second half of that VALU work together with the first k-step
// of tile t's second GEMM (sched_group_barrier interleave: one MFMA, one fragment read, a share of the VALU work), so that the
// ~320 VALU operations per tile run in the shadow of MFMAs instead of between them.  The weight images then live in two rings of
// two slots (W1 of tiles t + 1 / t + 2, W2 of tiles t / t + 1): the same 64 KB.
template <int FC, int FH, int WAVES, bool PIPE>
__global__ __launch_bounds__(64 * WAVES, FC <= 128 ? 2 : 1) void mlp_fused_x3_kernel(const float* __restrict__ X, const uint4* __restrict__ Wp,
                                                             const float* __restrict__ b1, const float* __restrict__ b2,
                                                             const float* __restrict__ gamma, const float* __restrict__ resid,
                                                             float* __restrict__ Y, int M, int* flag1, int* flag2) {
  using G = FusedGeom<FC, FH>;
  constexpr int NK1 = G::NK1, NT = G::NT, NOT = G::NOT, TILE_SLOTS = G::TILE_SLOTS;
  extern __shared__ uint4 smem[];                       // [2][TILE_SLOTS] weight ring, then b1 (FH floats)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 5;                              // k-block of the lane in a B operand / row group in an accumulator
  const float* trailer = reinterpret_cast<const float*>(Wp + (size_t)NT * TILE_SLOTS);
  const float wsc1 = trailer[1], wsc2 = trailer[5];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)smem;

  // 1 KB pieces of tile t's packed image: W1 part = pieces [0, W1_SLOTS / 64), W2 part behind it; each wave moves its share
  auto dma_part = [&](int t, int src_slot0, int n_pieces, unsigned dst_slot0) {
    for (int c = 0; c < n_pieces / WAVES; ++c) {
      const int piece = wave * (n_pieces / WAVES) + c;
      dma_v(Wp + (size_t)t * TILE_SLOTS + src_slot0 + piece * 64 + lane, lds0 + (dst_slot0 + (unsigned)piece * 64u) * 16u);
    }
  };
  constexpr int W1S = G::W1_SLOTS, W2S = G::W2_SLOTS;
  // non-PIPE: [2][W1 | W2] whole-tile stages.  PIPE: W1 ring [2][W1S] then W2 ring [2][W2S]
  auto dma_w1 = [&](int t, int slot) { dma_part(t, 0, W1S / 64, PIPE ? (unsigned)(slot * W1S) : (unsigned)(slot * TILE_SLOTS)); };
  auto dma_w2 = [&](int t, int slot) { dma_part(t, W1S, W2S / 64, PIPE ? (unsigned)(2 * W1S + slot * W2S) : (unsigned)(slot * TILE_SLOTS + W1S)); };
  // ---- the wave's 32 pixels: x rows -> LDS (coalesced LDS-DMA, 512 B per row) -> registers, split once.  A lane's B operand is
  // eight consecutive channels of ITS pixel per k-step: loading that straight from global memory is 32 rows x 16 B per instruction
  // (measured: prologue + epilogue of that form alone took 437 of the kernel's 675 us at 128 ROIs).  Row r of the wave sits in 1 KB
  // piece r >> 1, half r & 1, and its 16-byte chunk c at position c ^ (r & 15): the 16 lanes of a ds_read_b128 group then read 16
  // different bank groups (same chunk of 16 rows), and the DMA side only permutes the chunks inside a row.
  static_assert((FC == 128 || FC == 256) && WAVES == 4, "row staging: 32 / 64 chunks per row, the waves' 32 rows cover the weight ring");
  constexpr int CPR = FC / 4;                           // 16-byte chunks per row
  constexpr int RPP = 64 / CPR;                         // rows per 1 KB piece (= per DMA instruction)
  const long pix0 = (long)blockIdx.x * (32 * WAVES) + wave * 32;
  const int rl = lane & 31;                             // the lane's pixel in the operand / accumulator layouts
  const int cl = lane % CPR, rsub = lane / CPR;         // the lane's chunk / row of the piece in the row-major (coalesced) layout
  uint4* rows = smem + wave * (32 * CPR);               // the wave's 16 / 32 KB: row r at r * CPR, chunk c at position c ^ (r & 15)
  unsigned xh[NK1][4], xl[NK1][4];
  float ssx = 0.f;
  {
#pragma unroll
    for (int p = 0; p < 32 / RPP; ++p) {
      const int row = RPP * p + rsub;
      const long rp = pix0 + row < M ? pix0 + row : (long)M - 1;
      dma_v(reinterpret_cast<const uint4*>(X + rp * FC) + (cl ^ (row & 15)), lds0 + (unsigned)(wave * (32 * CPR) + p * 64) * 16u);
    }
    wait_vmcnt<0>();
    const uint4* xr = rows + rl * CPR;
#pragma unroll
    for (int ks = 0; ks < NK1; ++ks) {
      const float4 a = __builtin_bit_cast(float4, xr[(4 * ks + 2 * g) ^ (rl & 15)]), b = __builtin_bit_cast(float4, xr[(4 * ks + 2 * g + 1) ^ (rl & 15)]);
      const float x8[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
      split8(x8, xh[ks], xl[ks], ssx);
    }
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_s_barrier();                         // every wave has its rows in registers: the ring is free for the weights
  dma_w1(0, 0);
  dma_w2(0, 0);
  if (PIPE && NT > 1) dma_w1(1, 1);
  // b1 goes through LDS: a global load of it inside the tile loop would make the compiler place an s_waitcnt vmcnt(n) in front of
  // its first use, and that counter also counts the LDS-DMA loads issued just before
  float4* b1s = reinterpret_cast<float4*>(smem + 2 * TILE_SLOTS);
  for (int i = tid; i < FH / 4; i += 64 * WAVES) b1s[i] = reinterpret_cast<const float4*>(b1)[i];
  f32x16 acc2[NOT];
#pragma unroll
  for (int o = 0; o < NOT; ++o)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[o][r] = 0.f;
  float ssh = 0.f;

  wait_vmcnt<0>();
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_s_barrier();

  // H^T tile = W1 tile . X^T, even / odd k-steps in two accumulators (consecutive MFMAs never wait for each other), summed
  auto fc1_dual = [&](const uint4* w1) {
    f32x16 a, b2nd;
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = b2nd[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < NK1; ks += 2) {
      const f16x8 wh0 = __builtin_bit_cast(f16x8, w1[(ks * 2 + 0) * 64]), wl0 = __builtin_bit_cast(f16x8, w1[(ks * 2 + 1) * 64]);
      const f16x8 wh1 = __builtin_bit_cast(f16x8, w1[(ks * 2 + 2) * 64]), wl1 = __builtin_bit_cast(f16x8, w1[(ks * 2 + 3) * 64]);
      a = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl0, frag(xh[ks]), a, 0, 0, 0);
      b2nd = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl1, frag(xh[ks + 1]), b2nd, 0, 0, 0);
      a = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh0, frag(xl[ks]), a, 0, 0, 0);
      b2nd = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh1, frag(xl[ks + 1]), b2nd, 0, 0, 0);
      a = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh0, frag(xh[ks]), a, 0, 0, 0);
      b2nd = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh1, frag(xh[ks + 1]), b2nd, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] += b2nd[r];
    return a;
  };
  // scale, bias, GELU, split of half s of a tile's accumulator: register r of the lane = hidden unit 32 t + (r & 3) + 8 (r >> 2) + 4 g
  // of ITS pixel; registers 8 s .. 8 s + 7 are the lane's eight k of k-step s of the second GEMM
  auto gelu_half = [&](const f32x16& acc1, int t, int sidx, unsigned (&hh)[4], unsigned (&hl)[4]) {
    float v[8];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int j = 2 * sidx + jj;
      const float4 bv = b1s[8 * t + 2 * j + g];
      auto act = [](float q) { return (GDRNPP_MLPF_ABL & 1) ? q : gelu_erf(q); };
      v[4 * jj + 0] = act(acc1[4 * j + 0] * wsc1 + bv.x);
      v[4 * jj + 1] = act(acc1[4 * j + 1] * wsc1 + bv.y);
      v[4 * jj + 2] = act(acc1[4 * j + 2] * wsc1 + bv.z);
      v[4 * jj + 3] = act(acc1[4 * j + 3] * wsc1 + bv.w);
    }
    split8(v, hh, hl, ssh);
  };
  // Y^T += W2[:, k-step s of the tile] . hidden: product-major (consecutive MFMAs hit different accumulators)
  auto fc2_step = [&](const uint4* w2, int sidx, const unsigned (&hh)[4], const unsigned (&hl)[4]) {
    f16x8 wh[NOT], wl[NOT];
#pragma unroll
    for (int o = 0; o < NOT; ++o) {
      wh[o] = __builtin_bit_cast(f16x8, w2[((sidx * NOT + o) * 2 + 0) * 64]);
      wl[o] = __builtin_bit_cast(f16x8, w2[((sidx * NOT + o) * 2 + 1) * 64]);
    }
#pragma unroll
    for (int o = 0; o < NOT; ++o) acc2[o] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[o], frag(hh), acc2[o], 0, 0, 0);
#pragma unroll
    for (int o = 0; o < NOT; ++o) acc2[o] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[o], frag(hl), acc2[o], 0, 0, 0);
#pragma unroll
    for (int o = 0; o < NOT; ++o) acc2[o] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[o], frag(hh), acc2[o], 0, 0, 0);
  };

  if constexpr (!PIPE) {
    for (int t = 0; t < ((GDRNPP_MLPF_ABL & 16) ? 0 : NT); ++t) {
      const int stage = (GDRNPP_MLPF_ABL & 2) ? 0 : (t & 1);
      if (!(GDRNPP_MLPF_ABL & 2) && t + 1 < NT) { dma_w1(t + 1, stage ^ 1); dma_w2(t + 1, stage ^ 1); }   // the other stage was last read in iteration t - 1
      const uint4* w1 = smem + stage * TILE_SLOTS + lane;
      f32x16 acc1;
      if constexpr (GDRNPP_MLPF_ABL & 8) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[r] = __builtin_bit_cast(float, xh[r & 7][r >> 3]) + (float)t;
      } else {
        acc1 = fc1_dual(w1);
      }
      unsigned hh[2][4], hl[2][4];
      gelu_half(acc1, t, 0, hh[0], hl[0]);
      gelu_half(acc1, t, 1, hh[1], hl[1]);
      if constexpr (GDRNPP_MLPF_ABL & 4) {
#pragma unroll
        for (int p = 0; p < 4; ++p) { acc2[0][p] += __builtin_bit_cast(float, hh[0][p] ^ hl[1][p]); acc2[1][p] += __builtin_bit_cast(float, hh[1][p] ^ hl[0][p]); }
      } else {
        fc2_step(w1 + W1S, 0, hh[0], hl[0]);
        fc2_step(w1 + W1S, 1, hh[1], hl[1]);
      }
      if constexpr (!(GDRNPP_MLPF_ABL & 2)) {
        wait_vmcnt<0>();                                    // this wave's pieces of tile t + 1 have landed
        __builtin_amdgcn_s_waitcnt(0xc07f);                 // its fragment reads of tile t have returned
        __builtin_amdgcn_s_barrier();                       // tile t + 1 visible to all; stage of tile t free
      }
    }
  } else {
    // ---- software pipeline, pinned slot by slot (one MFMA + its share of everything else per slot, sched_barrier between slots):
    //   phase A  3 NK1 slots  first GEMM of tile t + 1          | fragment reads of the next k-step; W2 fragments of k-step 0
    //   phase B  3 NOT slots  second GEMM of tile t, k-step 0   | W2 l fragments of k-step 1
    //   phase C  3 NOT slots  second GEMM of tile t, k-step 1   | W2 h fragments of k-step 1, biases of tile t + 1
    // and the VALU work of tile t — 16 values x 3 GELU sub-steps of 5 operations + 8 pair splits of 5 — dealt over the slots of
    // phases A and B in value order (1.5 micro-steps per slot on average: 7-8 VALU operations in the shadow of each MFMA); the
    // first half (the operand of k-step 0) is complete well before phase B starts.
    constexpr int NSA = 3 * NK1, NSB = 3 * NOT, NSC = 3 * NOT, NMICRO = 56;
    f32x16 acc1 = fc1_dual(smem + lane);                  // tile 0 (W1 ring slot 0), not overlapped with anything
    float4 bvq[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bvq[j] = b1s[2 * j + g];
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();                         // every wave has read W1(0): its slot may take W1(2)
    f16x8 w1h[2], w1l[2], w2h[NOT], w2l[NOT];
    if (NT > 1) {
      w1h[0] = __builtin_bit_cast(f16x8, smem[W1S + lane]);            // k-step 0 of tile 1 (W1 ring slot 1)
      w1l[0] = __builtin_bit_cast(f16x8, smem[W1S + 64 + lane]);
    }
    float gpre = 0.f, gu = 0.f, gq = 0.f;                 // the one value in flight through the GELU sub-steps
    unsigned hh[2][4], hl[2][4];
    if constexpr (GDRNPP_MLPF_ABL & 64)
      for (int i = 0; i < 4; ++i) { hh[0][i] = xh[0][i]; hh[1][i] = xh[1][i]; hl[0][i] = xl[0][i]; hl[1][i] = xl[1][i]; }

    auto body = [&](int t, auto last_) {
      constexpr bool LAST = decltype(last_)::value;
      if (t + 2 < NT) dma_w1(t + 2, t & 1);
      if (t + 1 < NT) dma_w2(t + 1, (t + 1) & 1);
      const uint4* w1n = smem + ((t + 1) & 1) * W1S + lane;
      const uint4* w2c = smem + 2 * W1S + (t & 1) * W2S + lane;
      f32x16 acc1n;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc1n[r] = 0.f;
      // micro-step m of the tile's VALU work: half = m / 28; inside a half 4 groups of 7 = value 2 p (3 sub-steps), value 2 p + 1 (3), pair p
      auto micro = [&](auto m_) {
        constexpr int m = decltype(m_)::value, hf = m / 28, w = m % 28, pp = w / 7, k = w % 7;
        if constexpr (k < 6) {
          constexpr int r = 8 * hf + 2 * pp + (k >= 3 ? 1 : 0), sub = k % 3;
          if constexpr (sub == 0) {
            const float bias = (r & 3) == 0 ? bvq[r >> 2].x : ((r & 3) == 1 ? bvq[r >> 2].y : ((r & 3) == 2 ? bvq[r >> 2].z : bvq[r >> 2].w));
            gpre = fmaf(acc1[r], wsc1, bias);
            gu = fminf(fabsf(gpre), 5.9f);
            gq = fmaf(-2.7721912374545354e-06f, gu, 3.862218727590516e-05f);
            gq = fmaf(gq, gu, -0.00018255332543049008f);
            gq = fmaf(gq, gu, -0.000145858692121692f);
          } else if constexpr (sub == 1) {
            gq = fmaf(gq, gu, 0.007075459696352482f);
            gq = fmaf(gq, gu, -0.052505023777484894f);
            gq = fmaf(gq, gu, -0.45920491218566895f);
            gq = fmaf(gq, gu, -1.1511057615280151f);
            gq = fmaf(gq, gu, -1.0f);
          } else {                                        // = gelu_erf (common.hpp), split over the three sub-steps
            const float rr = fmaf(-gu, __builtin_amdgcn_exp2f(gq), fmaxf(gpre, 0.f));
            acc1[r] = gpre != gpre ? gpre : rr;
          }
        } else {
          constexpr int r0 = 8 * hf + 2 * pp;
          hh[hf][pp] = cvt_pk_f16(acc1[r0], acc1[r0 + 1]);
          hl[hf][pp] = cvt_pk_f16(residual<0>(acc1[r0], hh[hf][pp]), residual<1>(acc1[r0 + 1], hh[hf][pp]));
          ssh = sumsq2(hh[hf][pp], ssh);
        }
      };
      static_for<0, NSA + NSB + NSC>([&](auto s_) {
        constexpr int S = decltype(s_)::value;
        // ---- the slot's MFMA
        if constexpr (S < NSA) {
          if constexpr (!LAST) {
            constexpr int ks = S / 3, pr = S % 3;
            if constexpr (GDRNPP_MLPF_ABL & 32)       // timing only: the 24 MFMAs of phase A on four accumulators instead of one
              acc2[S & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pr == 0 ? w1l[ks & 1] : w1h[ks & 1], pr == 1 ? frag(xl[ks]) : frag(xh[ks]), acc2[S & 3], 0, 0, 0);
            else
            acc1n = __builtin_amdgcn_mfma_f32_32x32x16_f16(pr == 0 ? w1l[ks & 1] : w1h[ks & 1], pr == 1 ? frag(xl[ks]) : frag(xh[ks]), acc1n, 0, 0, 0);
            if constexpr (pr == 0 && ks + 1 < NK1) {       // fragments of the next k-step into the other register pair
              w1h[(ks + 1) & 1] = __builtin_bit_cast(f16x8, w1n[((ks + 1) * 2 + 0) * 64]);
              w1l[(ks + 1) & 1] = __builtin_bit_cast(f16x8, w1n[((ks + 1) * 2 + 1) * 64]);
            }
          }
          if constexpr (S >= NSA - 2 * NOT && S < NSA - NOT) w2l[S - (NSA - 2 * NOT)] = __builtin_bit_cast(f16x8, w2c[((0 * NOT + (S - (NSA - 2 * NOT))) * 2 + 1) * 64]);
          if constexpr (S >= NSA - NOT) w2h[S - (NSA - NOT)] = __builtin_bit_cast(f16x8, w2c[((0 * NOT + (S - (NSA - NOT))) * 2 + 0) * 64]);
        } else if constexpr (S < NSA + NSB) {
          constexpr int idx = S - NSA, pr = idx / NOT, o = idx % NOT;
          acc2[o] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pr == 0 ? w2l[o] : w2h[o], pr == 1 ? frag(hl[0]) : frag(hh[0]), acc2[o], 0, 0, 0);
          if constexpr (pr == 1) w2l[o] = __builtin_bit_cast(f16x8, w2c[((1 * NOT + o) * 2 + 1) * 64]);    // l of k-step 1 (its k-step-0 copy is spent)
        } else {
          constexpr int idx = S - NSA - NSB, pr = idx / NOT, o = idx % NOT;
          if constexpr (pr == 0) {
            acc2[o] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2l[o], frag(hh[1]), acc2[o], 0, 0, 0);
            w2h[o] = __builtin_bit_cast(f16x8, w2c[((1 * NOT + o) * 2 + 0) * 64]);                          // h of k-step 1
            if constexpr (!LAST && o < 4) bvq[o] = b1s[8 * (t + 1) + 2 * o + g];   // the 4 bias quads of the next tile
          } else {
            acc2[o] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2h[o], pr == 1 ? frag(hl[1]) : frag(hh[1]), acc2[o], 0, 0, 0);
          }
        }
        // ---- the slot's share of the VALU work: phases A and B, alternately 2 and 1 (A) / 2, 2, 1 (B) micro-steps
        // (36 micro-steps over phase A, 20 over phase B; NSA = 24, NSB = 12: (3 S + 1) / 2 and 36 + (5 s + 2) / 3)
        constexpr auto mstart = [](int sl) { return sl < NSA ? (36 * sl + NSA / 2) / NSA : (sl < NSA + NSB ? 36 + (20 * (sl - NSA) + 2 * NSB / 3) / NSB : NMICRO); };
        constexpr int m0 = mstart(S), m1 = mstart(S + 1);
        if constexpr (!(GDRNPP_MLPF_ABL & 64)) static_for<(m0 < NMICRO ? m0 : NMICRO), (m1 < NMICRO ? m1 : NMICRO)>(micro);
        __builtin_amdgcn_sched_barrier(0);
      });
      if constexpr (!LAST) acc1 = acc1n;
      wait_vmcnt<0>();
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_s_barrier();
      if constexpr (!LAST) {
        if (t + 2 < NT) {                                 // k-step 0 of tile t + 2 for the next trip's first slot
          const uint4* w1nn = smem + (t & 1) * W1S + lane;
          w1h[0] = __builtin_bit_cast(f16x8, w1nn[0]);
          w1l[0] = __builtin_bit_cast(f16x8, w1nn[64]);
        }
      }
    };
    for (int t = 0; t + 1 < NT; ++t) body(t, std::integral_constant<bool, false>{});
    body(NT - 1, std::integral_constant<bool, true>{});
  }

  // ---- epilogue: y = resid + gamma * (acc2 * 2^-e2 + b2).  Register quad j of output tile o = channels 32 o + 8 j + 4 g .. + 3 of
  // the lane's pixel: through the wave's 16 KB of LDS (same layout as the x rows; the loop's last barrier freed the ring) into the
  // row-major form, where lane = (row parity, chunk): resid loads and y stores are whole 512-byte rows, b2 / gamma one quad per lane.
  {
    uint4* yr = rows + rl * CPR;
#pragma unroll
    for (int o = 0; o < NOT; ++o)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        yr[(8 * o + 2 * j + g) ^ (rl & 15)] = __builtin_bit_cast(uint4, make_float4(acc2[o][4 * j], acc2[o][4 * j + 1], acc2[o][4 * j + 2], acc2[o][4 * j + 3]));
  }
  bool bad = false;
  {
    long pix0e = pix0;                                  // opaque copies: the row addresses are recomputed here instead of being kept
    int cle = cl, rse = rsub;                           // (spilled) across the tile loop
    asm volatile("" : "+s"(pix0e), "+v"(cle), "+v"(rse));
    const float4 bv = *reinterpret_cast<const float4*>(b2 + 4 * cle), gv = *reinterpret_cast<const float4*>(gamma + 4 * cle);
#pragma unroll
    for (int p = 0; p < 32 / RPP; ++p) {
      const int row = RPP * p + rse;
      const long rp = pix0e + row;
      const long rpc = rp < M ? rp : (long)M - 1;
      const float4 rs = *reinterpret_cast<const float4*>(resid + rpc * FC + 4 * cle);
      const float4 a = __builtin_bit_cast(float4, rows[p * 64 + rse * CPR + (cle ^ (row & 15))]);
      float4 v = make_float4(a.x * wsc2 + bv.x, a.y * wsc2 + bv.y, a.z * wsc2 + bv.z, a.w * wsc2 + bv.w);
      v.x = rs.x + gv.x * v.x; v.y = rs.y + gv.y * v.y; v.z = rs.z + gv.z * v.z; v.w = rs.w + gv.w * v.w;
      constexpr unsigned kInfNan = 0x203u;
      bad |= __builtin_amdgcn_class(v.x, kInfNan) | __builtin_amdgcn_class(v.y, kInfNan) | __builtin_amdgcn_class(v.z, kInfNan) |
             __builtin_amdgcn_class(v.w, kInfNan);
      if (rp < M) {
        const f32x4v t4 = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(t4, reinterpret_cast<f32x4v*>(Y + rp * FC + 4 * cle));
      }
    }
  }
  // ---- range verdicts per pixel: lanes l and l ^ 32 hold the two k-block halves of its x row / hidden row
  const float sx = ssx + __shfl_xor(ssx, 32, 64), sh = ssh + __shfl_xor(ssh, 32, 64);
  const bool small1 = sx > 0.f && sx < (float)FC * 0x1p-8f, bad1 = !(sx < __builtin_inff());
  const bool small2 = sh > 0.f && sh < (float)FH * 0x1p-8f, bad2 = !(sh < __builtin_inff()) || bad;
  const int word1 = (__builtin_amdgcn_ballot_w64(bad1) != 0 ? GDRNPP_SPLIT2_NONFINITE : 0) |
                    (__builtin_amdgcn_ballot_w64(small1) != 0 ? GDRNPP_SPLIT2_SMALL_ROWS : 0);
  const int word2 = (__builtin_amdgcn_ballot_w64(bad2) != 0 ? GDRNPP_SPLIT2_NONFINITE : 0) |
                    (__builtin_amdgcn_ballot_w64(small2) != 0 ? GDRNPP_SPLIT2_SMALL_ROWS : 0);
  if (lane == 0) {
    if (word1) atomicOr(flag1, word1);
    if (word2) atomicOr(flag2, word2);
  }
}

// W1 f32[FH][FC], W2 f32[FC][FH] -> the per-hidden-tile LDS images (see the header comment).  One thread per uint4 slot.
// trailer (32 B behind the tiles): {amax1 bits, 2^-e1, 2^e1, rows1 flag, amax2 bits, 2^-e2, 2^e2, rows2 flag}
template <int FC, int FH>
__global__ void pack_mlp_fused_kernel(const float* __restrict__ W1, const float* __restrict__ W2, uint4* __restrict__ packed) {
  using G = FusedGeom<FC, FH>;
  unsigned* trailer = reinterpret_cast<unsigned*>(packed + (size_t)G::NT * G::TILE_SLOTS);
  const int e1 = weight_exp(trailer[0]), e2 = weight_exp(trailer[4]);
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) {
    trailer[1] = __float_as_uint(__builtin_ldexpf(1.f, -e1));
    trailer[2] = __float_as_uint(__builtin_ldexpf(1.f, e1));
    trailer[5] = __float_as_uint(__builtin_ldexpf(1.f, -e2));
    trailer[6] = __float_as_uint(__builtin_ldexpf(1.f, e2));
  }
  if (i >= (long)G::NT * G::TILE_SLOTS) return;
  const int t = (int)(i / G::TILE_SLOTS), u = (int)(i % G::TILE_SLOTS);
  const int lane = u & 63, row = lane & 31, kb = lane >> 5;
  float x[8];
  float sc;
  int plane;
  if (u < G::W1_SLOTS) {          // [ks][plane][lane]: W1[32 t + row][16 ks + 8 kb + q]
    const int ks = u / 128;
    plane = (u / 64) & 1;
    sc = __builtin_ldexpf(1.f, e1);
    const float* src = W1 + (size_t)(32 * t + row) * FC + 16 * ks + 8 * kb;
#pragma unroll
    for (int q = 0; q < 8; ++q) x[q] = src[q] * sc;
  } else {                        // [s][ot][plane][lane]: W2[32 ot + row][32 t + (q & 3) + 8 (2 s + (q >> 2)) + 4 kb]
    const int v = u - G::W1_SLOTS;
    const int s = v / (G::NOT * 128), ot = (v / 128) % G::NOT;
    plane = (v / 64) & 1;
    sc = __builtin_ldexpf(1.f, e2);
    const float* src = W2 + (size_t)(32 * ot + row) * FH + 32 * t + 4 * kb;
#pragma unroll
    for (int q = 0; q < 8; ++q) x[q] = src[(q & 3) + 8 * (2 * s + (q >> 2))] * sc;
  }
  unsigned h[4], l[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    h[p] = cvt_pk_f16(x[2 * p], x[2 * p + 1]);
    const f16x2 hv = __builtin_bit_cast(f16x2, h[p]);
    l[p] = cvt_pk_f16(x[2 * p] - (float)hv[0], x[2 * p + 1] - (float)hv[1]);
  }
  packed[i] = plane == 0 ? make_uint4(h[0], h[1], h[2], h[3]) : make_uint4(l[0], l[1], l[2], l[3]);
}

// the two shapes the fused form exists for: ConvNeXt-B stage 0 (two workgroups per CU) and stage 1 (512 registers per lane, one)
template <int FC, int FH>
size_t fused_bytes() { return (size_t)FusedGeom<FC, FH>::NT * FusedGeom<FC, FH>::TILE_SLOTS * 16 + 32; }

template <int FC, int FH>
int pack_fused(const float* W1, const float* W2, void* packed, hipStream_t st) {
  using Geom = FusedGeom<FC, FH>;
  unsigned* trailer = reinterpret_cast<unsigned*>(reinterpret_cast<uint4*>(packed) + (size_t)Geom::NT * Geom::TILE_SLOTS);
  GDRNPP_HIP_TRY(hipMemsetAsync(trailer, 0, 32, st));
  const long n = (long)FC * FH;
  hipLaunchKernelGGL(amax_kernel, dim3(64), dim3(256), 0, st, W1, n, trailer);
  hipLaunchKernelGGL(amax_kernel, dim3(64), dim3(256), 0, st, W2, n, trailer + 4);
  const long slots = (long)Geom::NT * Geom::TILE_SLOTS;
  hipLaunchKernelGGL((pack_mlp_fused_kernel<FC, FH>), dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, st, W1, W2, (uint4*)packed);
  hipLaunchKernelGGL(weight_rows_range_kernel, dim3((unsigned)FH), dim3(256), 0, st, W1, trailer, trailer + 3, FC);
  hipLaunchKernelGGL(weight_rows_range_kernel, dim3((unsigned)FC), dim3(256), 0, st, W2, trailer + 4, trailer + 7, FH);
  return gdrnpp::check_launch("gdrnpp_pack_mlp_fused_f16x2");
}

template <int FC, int FH>
int launch_fused(const float* x, const void* W_packed, const float* b1, const float* b2, const float* gamma, const float* resid, float* y,
                 int M, int* range_flag_fc1, int* range_flag_fc2, hipStream_t st) {
  using Geom = FusedGeom<FC, FH>;
  constexpr int lds_bytes = 2 * Geom::TILE_SLOTS * 16 + FH * 4;   // weight ring + b1
  const int pipe = gdrnpp::option_mlp_fused_pipe();
  auto go = [&](auto kernel, int w) -> int {
    if (int rc = gdrnpp::ensure_dynamic_lds((const void*)kernel, lds_bytes)) return rc;
    hipLaunchKernelGGL(kernel, dim3((unsigned)((M + 32 * w - 1) / (32 * w))), dim3(64 * w), lds_bytes, st, x,
                       (const uint4*)W_packed, b1, b2, gamma, resid, y, M, range_flag_fc1, range_flag_fc2);
    return 0;
  };
  if (int rc = pipe ? go(mlp_fused_x3_kernel<FC, FH, 4, true>, 4) : go(mlp_fused_x3_kernel<FC, FH, 4, false>, 4)) return rc;
  return gdrnpp::check_launch("gdrnpp_convnext_mlp_f32_fused");
}

}  // namespace

extern "C" size_t gdrnpp_pack_mlp_fused_f16x2_bytes(int C, int hidden) {
  if (C == 128 && hidden == 512) return fused_bytes<128, 512>();
  if (C == 256 && hidden == 1024) return fused_bytes<256, 1024>();
  return 0;
}

extern "C" int gdrnpp_pack_mlp_fused_f16x2(const float* W1, const float* W2, void* packed, int C, int hidden, void* stream) {
  GDRNPP_REQUIRE(W1 && W2 && packed, GDRNPP_EINVAL, "gdrnpp_pack_mlp_fused_f16x2: null pointer");
  if (C == 128 && hidden == 512) return pack_fused<128, 512>(W1, W2, packed, (hipStream_t)stream);
  if (C == 256 && hidden == 1024) return pack_fused<256, 1024>(W1, W2, packed, (hipStream_t)stream);
  GDRNPP_REQUIRE(false, GDRNPP_ELIMIT, "gdrnpp_pack_mlp_fused_f16x2: C=%d hidden=%d (the fused form exists for 128 / 512 and 256 / 1024)", C, hidden);
  return 0;
}

extern "C" int gdrnpp_convnext_mlp_f32_fused(const float* x, const void* W_packed, const float* b1, const float* b2, const float* gamma,
                                             const float* resid, float* y, int M, int C, int hidden, int* range_flag_fc1,
                                             int* range_flag_fc2, void* stream) {
  GDRNPP_REQUIRE(x && W_packed && b1 && b2 && gamma && resid && y && range_flag_fc1 && range_flag_fc2, GDRNPP_EINVAL,
                 "gdrnpp_convnext_mlp_f32_fused: null pointer");
  GDRNPP_REQUIRE(M > 0, GDRNPP_EINVAL, "gdrnpp_convnext_mlp_f32_fused: M=%d", M);
  if (C == 128 && hidden == 512)
    return launch_fused<128, 512>(x, W_packed, b1, b2, gamma, resid, y, M, range_flag_fc1, range_flag_fc2, (hipStream_t)stream);
  if (C == 256 && hidden == 1024)
    return launch_fused<256, 1024>(x, W_packed, b1, b2, gamma, resid, y, M, range_flag_fc1, range_flag_fc2, (hipStream_t)stream);
  GDRNPP_REQUIRE(false, GDRNPP_ELIMIT, "gdrnpp_convnext_mlp_f32_fused: C=%d hidden=%d (the fused form exists for 128 / 512 and 256 / 1024)", C, hidden);
  return 0;
}
