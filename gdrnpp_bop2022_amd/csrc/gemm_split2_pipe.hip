// Three-product form of the split GEMM ("fp16x2"): the default of the ConvNeXt MLPs, the convolutions of the head / downsample
// layers / Patch-PnP and the transposed-convolution GEMM where a launch has at least 256 tiles of 256x128
// (hip_layers.set_gemm_products; entry points gdrnpp_linear_f32_split2 / gdrnpp_conv3x3_f32_split2 / gdrnpp_conv2d_f32_split2)
// — SURVEY.md §8 row a3.
//
// Numerical scheme.  Every fp32 operand is written as x = h + l + e with h = rn_f16(x), l = rn_f16(x - h) (the subtraction is
// exact) and |e| <= 2^-22 |x| as long as l stays in the normal fp16 range (else |e| <= 2^-25 absolute): 22 significant bits in
// two fp16 values.  Three of the four partial products go through v_mfma_f32_32x32x16_f16 with fp32 accumulation
// (h*l, l*h, h*h, small terms first); l*l (2^-22 relative) is dropped.  fp16 products are exact in fp32, so what is lost
// against the six-product bf16x3 form (gemm_split.hip: exact to 2^-26) is the operand representation: 1.3e-7 .. 2.2e-7 of the
// output scale (the same three products evaluated in fp64).  On operands with outlier channels (ConvNeXt activations) that sits
// below the error every fp32-accumulating GEMM makes in its accumulation chain: measured against fp64 at K = 128 .. 4096 the
// three-product result carries 3.9e-7 .. 3.0e-6, the six-product form 4.8e-7 .. 3.0e-6, hipBLASLt's fp32 GEMM 6.2e-7 .. 3.9e-6
// (tools/split2_error_probe.py, profiles/r03y_split2_accuracy.txt; tests/test_gpu_split2.py pins the three against each other), and
// the network outputs sit at the same 4e-6 .. 8e-6 from the reference's recorded forward with any of the three engines.
// Half the matrix-pipe work of the six-product form for the accuracy of the arithmetic the reference itself runs in.
//
// Range.  fp16 has 5 exponent bits, so the operands are brought into range by exact power-of-two scaling:
//   * weights: gdrnpp_pack_weight_f16x2 scales the tensor by 2^e with max|w| * 2^e in [2^13, 2^14) before splitting and stores
//     2^-e behind the packed tiles; the epilogue multiplies the accumulator by it (exact);
//   * activations are split as they are: |x| up to 65504 is representable, elements below 2^-2 lose l's low bits to the fp16
//     subnormal spacing (absolute error 2^-25, i.e. still 2^-22 of any tensor whose scale is 2^-3 or more — LayerNorm /
//     GroupNorm / GELU outputs).  An activation beyond 65504 becomes inf and the output non-finite: the epilogue checks every
//     value it stores and raises bit 0 of the launch's range word (the range_flag argument: one word per layer and stream on
//     the Python side);
//   * the small side is checked on every launch as well, per A row: the k-loop accumulates the sum of squares of the row's h
//     halves (v_dot2c_f32_f16, 8 per k-tile and wave), and a row that is not all zero with rms below 2^-4 raises bit 1 — below
//     that scale the 2^-25 absolute operand error of the subnormal l would exceed 2^-21 of the row's own output scale (zero-padded
//     taps of a convolution count as zeros; so does a row whose every element is below half the fp16 subnormal spacing, 3e-8:
//     its h and l are all zero, what it would have contributed — less than 3e-8 * sum|w| — is dropped, which is the 2^-25 absolute
//     operand error again).  A non-finite A element makes the sum non-finite and raises bit 0.
//   On either bit the host side re-runs the step in the six-product form and keeps the flagged layer there
//   (engine.run_with_range_check) — never silently wrong, on either side of the range, on every launch;
//   * weights: gdrnpp_pack_weight_f16x2 applies the same per-row test to the scaled weight rows (trailer word 3); such a layer
//     is not eligible for this form (hip_layers._packed_weight).
//
// Kernel: the software-pipelined LDS-DMA kernel of gemm_split_pipe.hip with 24 instead of 48 MFMA slots per k-tile: block tile
// 256x128x16, 4 waves stacked along M (2 x 4 MFMA tiles each), fp32 A by LDS-DMA into three 16 KB stages private to the waves,
// pre-packed fp16 weight tiles (8 KB: [split][k-block][128][8]) by LDS-DMA into two stages, the split of k-tile t+1 (cvt_pk,
// v_fma_mix_f32 residual, cvt_pk: 32 VALU operations) spread over the MFMA slots of k-tile t, both weight fragment sets read
// just in time (l behind the barrier of the previous k-tile, h in slots 1 and 3), one barrier per k-tile behind slot 19.
// 80 KB of dynamic LDS (three A stages, four weight slots: one barrier per two k-tiles), two workgroups per CU.  Linear form, the 3x3 / stride 1 / pad 1 convolution and the general K x K /
// stride / pad convolution (implicit im2col, k-tile order of gemm_split.hpp), optional GroupNorm statistics in the epilogue (as
// gemm_split_glds_kernel<.., GNS>).  -DGDRNPP2_TIMING_NO_{BREAD,SPLIT,SYNC,DMA}: timing-only builds (results invalid) behind
// profiles/r03y_split2_kloop_dissection.txt.
#include "split2_common.hpp"

namespace {

using namespace gdrnpp::split2;

constexpr int NA = 3;                          // A stages: the A DMA runs two k-tiles ahead
constexpr int A_STAGE_B = 256 * BK * 4;        // fp32 A image of one k-tile: 16 KB
constexpr int W2_TILE_SLOTS = 2 * KB * BN;     // uint4 slots of one packed 128x16 fp16x2 weight tile
constexpr int W2_TILE_B = W2_TILE_SLOTS * 16;  // 8 KB

__device__ int g_split2_range_word;  // sticky range word (GDRNPP_SPLIT2_NONFINITE | GDRNPP_SPLIT2_SMALL_ROWS) of launches that pass no word of their own
__device__ __attribute__((aligned(64))) float g_split2_zero_page[16];

// One half of a wave's A tile for one k-tile: 32 rows x 16 k, 8 consecutive k of one row per lane.  x ~ h + l in 8 steps of two
// VALU operations (the residual overwrites x).  APRE: A is an "f16x2 rows" tensor (split2_common.hpp) — the two 16-byte chunks the
// lane reads ARE its h and l fragments, there is nothing to split.
template <bool APRE>
struct HalfSplit2T {
  float x[APRE ? 1 : 8];
  unsigned h[4], l[4];

  template <int S>
  __device__ __forceinline__ void step() {
    if constexpr (APRE) {
    } else if constexpr (S < 2) {
      h[2 * S] = cvt_pk_f16(x[4 * S], x[4 * S + 1]);
      h[2 * S + 1] = cvt_pk_f16(x[4 * S + 2], x[4 * S + 3]);
    } else if constexpr (S < 6) {
      constexpr int p = S - 2;
      x[2 * p] = residual<0>(x[2 * p], h[p]);
      x[2 * p + 1] = residual<1>(x[2 * p + 1], h[p]);
    } else {
      constexpr int q = S - 6;
      l[2 * q] = cvt_pk_f16(x[4 * q], x[4 * q + 1]);
      l[2 * q + 1] = cvt_pk_f16(x[4 * q + 2], x[4 * q + 3]);
    }
  }
  __device__ __forceinline__ void load(const uint4* lds, int slot0, int slot1) {
    if constexpr (APRE) {
      const uint4 a = lds[slot0], b = lds[slot1];
      h[0] = a.x; h[1] = a.y; h[2] = a.z; h[3] = a.w;
      l[0] = b.x; l[1] = b.y; l[2] = b.z; l[3] = b.w;
    } else {
      const float4 a = __builtin_bit_cast(float4, lds[slot0]), b = __builtin_bit_cast(float4, lds[slot1]);
      x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w;
      x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
    }
  }
  template <int SPLIT>
  __device__ __forceinline__ f16x8 frag() const {
    const unsigned* s = SPLIT == 0 ? h : l;
    return __builtin_bit_cast(f16x8, make_uint4(s[0], s[1], s[2], s[3]));
  }
};

#ifdef GDRNPP2_TIMING_A_DIRECT
// Timing-only experiment (results invalid; profiles/r04_a_direct.txt): what would the k-loop cost if the fp32 A rows of the
// linear form went from global memory straight to VGPRs — no LDS-DMA fill and no ds_read of the A stages (they are private to
// their wave, LDS buys them only asynchrony)?  A lane issues the loads of its MFMA operand rows (8 consecutive k per half: two
// global_load_dwordx4 per half and k-tile) for k-tile kt + 2 in slots 4 / 5 of k-tile kt and waits for them at the end of the
// same k-tile; the data lands in four scratch registers and is NOT used — the split keeps working on whatever its x registers
// hold.  (A build that used the data faulted: a value that is still in flight must not be copied, and the register allocator
// copies loop-carried values at the back edge — a production form needs the k-loop in assembly.)
using u32x4v = __attribute__((ext_vector_type(4))) unsigned;
struct ADirectScratch { u32x4v d[4]; };
__device__ __forceinline__ void gload_scratch(u32x4v& a, u32x4v& b, unsigned voff, const void* sbase) {
  asm volatile("global_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %3 offset:16"
               : "=&v"(a), "=&v"(b) : "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ void wait_scratch(ADirectScratch& s) {
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(s.d[0]), "+v"(s.d[1]), "+v"(s.d[2]), "+v"(s.d[3]) : : "memory");
}
#endif

// W f32[N][K] -> packed fp16 [N/128][K/16][2][2][128][8]; one thread per (row, k-block) = 8 consecutive k.
// trailer (16 B behind the tiles): {amax bits, 2^-e, 2^e, rows-below-range bit (weight_rows_range_kernel)}
__global__ void pack_weight2_kernel(const float* __restrict__ W, uint4* __restrict__ packed, int N, int K) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int kbs = K / 8;
  unsigned* trailer = reinterpret_cast<unsigned*>(packed + (size_t)(N / BN) * (K / BK) * W2_TILE_SLOTS);
  const int e = weight_exp(trailer[0]);
  const float sc = __builtin_ldexpf(1.f, e);
  if (i == 0) {
    trailer[1] = __float_as_uint(__builtin_ldexpf(1.f, -e));
    trailer[2] = __float_as_uint(sc);
    trailer[3] = 0u;
  }
  if (i >= (long)N * kbs) return;
  const int n = (int)(i / kbs), kb_g = (int)(i % kbs);
  const float4 v0 = *reinterpret_cast<const float4*>(W + (size_t)n * K + kb_g * 8);
  const float4 v1 = *reinterpret_cast<const float4*>(W + (size_t)n * K + kb_g * 8 + 4);
  const float x[8] = {v0.x * sc, v0.y * sc, v0.z * sc, v0.w * sc, v1.x * sc, v1.y * sc, v1.z * sc, v1.w * sc};
  unsigned h[4], l[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    h[p] = cvt_pk_f16(x[2 * p], x[2 * p + 1]);
    const f16x2 hv = __builtin_bit_cast(f16x2, h[p]);
    l[p] = cvt_pk_f16(x[2 * p] - (float)hv[0], x[2 * p + 1] - (float)hv[1]);
  }
  const int tn = n / BN, row = n % BN, tk = kb_g / KB, kb = kb_g % KB;
  uint4* img = packed + ((size_t)tn * (K / BK) + tk) * W2_TILE_SLOTS;
  img[(0 * KB + kb) * BN + row] = make_uint4(h[0], h[1], h[2], h[3]);
  img[(1 * KB + kb) * BN + row] = make_uint4(l[0], l[1], l[2], l[3]);
}

struct GnStats2 { double* part; int G; int tiles_per_img; };   // as GnStats of gemm_split.hip

// CONV: 0 = linear (A row-major [M,K]), 1 = 3x3 / stride 1 / pad 1 convolution over an NHWC image (geometry folded at compile
//       time), 2 = any KH x KW / stride / zero padding with at most 32 taps (ConvNeXt's 2x2/2 downsamples, Patch-PnP's 3x3/2)
// GNS: GroupNorm (sum, sum of squares) partials of the stored result per wave (64 rows x 8-channel groups), CONV only
// NJ: 32-column MFMA tiles per wave.  4: block tile 256 x 128, 24 slots per k-tile, 80 KB LDS, two workgroups per CU.
//     8: block tile 256 x 256 (two packed weight tiles side by side), 48 slots per k-tile, 256 accumulator registers — one wave
//     per SIMD, 80 KB LDS, one workgroup per CU: per MFMA half the A traffic (LDS-DMA, raw fragment reads, split arithmetic) of NJ = 4.
// APRE: A is an "f16x2 rows" tensor (linear form only).  c_rows (run time): C is written as one (bias / GELU epilogues).
template <int EPI, int CONV, bool GNS, int NJ, bool APRE = false>
__global__ __launch_bounds__(256, NJ == 4 ? 2 : 1) void gemm_split2_pipe_kernel(const float* __restrict__ A, const uint4* __restrict__ Wp,
                                                                  const float* __restrict__ bias,
                                                                  const float* __restrict__ gamma,
                                                                  const float* __restrict__ resid, float* __restrict__ C,
                                                                  int M, int N, int K, ConvGeom cg, int panel, GnStats2 gn, int* range_flag, int c_rows) {
  static_assert(!APRE || CONV == 0, "f16x2-rows A: linear form only");
  using HalfSplit2 = HalfSplit2T<APRE>;
  constexpr int BNB = NJ * 32;                     // block columns
  constexpr int NWT = NJ / 4;                      // packed 128-column weight tiles per block
  constexpr int B_STAGE_B = NWT * W2_TILE_B;       // 8 / 16 KB
#ifndef GDRNPP2_SINGLE_BARRIER   // -DGDRNPP2_SINGLE_BARRIER: the two-slot form with a barrier per k-tile (A/B; bitwise equal, 0.75 % slower per step)
  // PAIR: FOUR weight slots (k-tile kt in slot kt & 3), the weight DMA runs two k-tiles ahead and the workgroup meets at ONE barrier per
  // TWO k-tiles (behind the odd one): the slots a pair of k-tiles reads were complete at the previous barrier, the slots its DMA
  // fills were last read before it.  Waves of a workgroup may then drift by a whole k-tile instead of waiting for each other 2 x.
  constexpr bool PAIR = NJ == 4;
#else
  constexpr bool PAIR = false;
#endif
  constexpr int NS = 6 * NJ;                       // MFMA slots per k-tile
  constexpr int NBP = 2 * NWT;                     // weight DMA pieces per wave and k-tile
  constexpr int SB = NS - NJ / 2 - 3;              // slot of the barrier: behind it NJ/2 slots of fragment reads + one raw A read
  extern __shared__ uint4 smem[];  // the only LDS object: [NA][1024] A slots | [2][NWT * 512] weight slots
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntn = N / BNB;
  // XCD-aware tile order, as in gemm_split.hip
  const int nwg = gridDim.x, xcd = blockIdx.x & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
  int tile_m, tile_n;
  if (panel > 1) {  // panel order for wide layers, see launch_split_pipe (gemm_split_pipe.hip)
    const int ntm = (M + 255) >> 8, per = panel * ntn, p = tile / per, w = tile - p * per;
    const int rows = min(panel, ntm - p * panel);
    tile_n = w / rows;
    tile_m = p * panel + (w - tile_n * rows);
  } else {
    tile_m = tile / ntn;
    tile_n = tile - tile_m * ntn;
  }
  const int m0 = tile_m * 256, n0 = tile_n * BNB;
  const int nk = K / BK;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)smem;
  const float wsc = reinterpret_cast<const float*>(Wp + (size_t)(N / BN) * nk * W2_TILE_SLOTS)[1];   // 2^-e of the weight scale (trailer)

  // ---- DMA lanes: piece c (0..3) of this wave fills A slots (wave*4 + c)*64 + lane = rows wave*64 + c*16 + lane/4, chunk
  // q = (lane & 3) ^ ((row >> 2) & 3) of the row's 64-byte k segment (the swizzle is on the source address)
  const int prow = lane >> 2, pq = lane & 3;
  unsigned aoff[4];        // linear: byte offset of the lane's chunk from A + kt*64
  const float* ap[4];      // conv: anchor pixel of the lane's row (+ chunk)
  unsigned okmask[4];      // conv: bit tap = the tap lies inside the image
  int cpt = 1;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int lrow = wave * 64 + c * 16 + prow;
    const int q = pq ^ ((lrow >> 2) & 3);
    const int arow = min(m0 + lrow, M - 1);
    if constexpr (CONV == 1) {
      const int img = arow / (cg.H * cg.W), pp = arow - img * (cg.H * cg.W);
      const int iy = pp / cg.W, ix = pp - iy * cg.W;
      ap[c] = A + ((size_t)arow) * cg.C + q * 4;
      unsigned mk = 0;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int dy = t / 3 - 1, dx = t % 3 - 1;
        if ((unsigned)(iy + dy) < (unsigned)cg.H && (unsigned)(ix + dx) < (unsigned)cg.W) mk |= 1u << t;
      }
      okmask[c] = mk;
      aoff[c] = 0;
    } else if constexpr (CONV == 2) {   // general KH x KW / stride / pad: row = output pixel, anchor = its input pixel (oy*s, ox*s)
      const int img = arow / (cg.OH * cg.OW), pp = arow - img * (cg.OH * cg.OW);
      const int iy = (pp / cg.OW) * cg.stride, ix = (pp % cg.OW) * cg.stride;
      ap[c] = A + (((size_t)img * cg.H + iy) * cg.W + ix) * cg.C + q * 4;
      const int ntaps = K / cg.C;     // <= 32 (checked by the entry point)
      unsigned mk = 0;
      for (int t = 0; t < ntaps; ++t) {
        const int ky = t / cg.KW, dy = ky - cg.pad, dx = t - ky * cg.KW - cg.pad;
        if ((unsigned)(iy + dy) < (unsigned)cg.H && (unsigned)(ix + dx) < (unsigned)cg.W) mk |= 1u << t;
      }
      okmask[c] = mk;
      aoff[c] = 0;
    } else {
      aoff[c] = (unsigned)arow * (unsigned)(K * 4) + (unsigned)(q * 16);
      ap[c] = nullptr;
      okmask[c] = 0;
    }
  }
  if constexpr (CONV) cpt = cg.C / BK;
  const int ntaps = CONV == 1 ? 9 : (CONV == 2 ? K / cg.C : 1), ckw = CONV == 1 ? 3 : cg.KW, cpad = CONV == 1 ? 1 : cg.pad;
  const unsigned boff = (unsigned)((wave * 2) * 64 + lane) * 16u;
  const char* const wbase = reinterpret_cast<const char*>(Wp + (size_t)tile_n * NWT * nk * W2_TILE_SLOTS);
  const unsigned ldsA = lds0 + (unsigned)(wave * 4) * 1024u;
  const unsigned ldsB = lds0 + (unsigned)(NA * A_STAGE_B) + (unsigned)(wave * 2) * 1024u;

  // piece c of the A image of k-tile kt -> stage byte offset sb
  auto dma_a = [&](int kt, unsigned sb, auto cc) {
    constexpr int c = decltype(cc)::value;
    if constexpr (CONV) {
      const int cps = (cpt & 1) ? 1 : 2, sup = kt / (ntaps * cps), rem = kt - sup * (ntaps * cps), tap = rem / cps;  // gemm_split.hpp
      const int c0 = (sup * cps + (rem - tap * cps)) * BK;
      const int ky = tap / ckw, dy = ky - cpad, dx = tap - ky * ckw - cpad;
      const long off = ((long)dy * cg.W + dx) * cg.C + c0;
      const bool ok = (okmask[c] >> tap) & 1u;
      dma_v(ok ? (const void*)(ap[c] + off) : (const void*)g_split2_zero_page, ldsA + sb + c * 1024u);
    } else {
      dma_s(aoff[c], reinterpret_cast<const char*>(A) + (size_t)kt * (BK * 4), ldsA + sb + c * 1024u);
    }
  };
  auto dma_b = [&](int kt, unsigned sb, auto cc) {
    constexpr int c = decltype(cc)::value;
    int wkt = kt;
    if constexpr (CONV) {
      const int cps = (cpt & 1) ? 1 : 2, sup = kt / (ntaps * cps), rem = kt - sup * (ntaps * cps), tap = rem / cps;
      wkt = tap * cpt + sup * cps + (rem - tap * cps);
    }
    constexpr int t = c >> 1, sub = c & 1;   // piece c = half `sub` of this wave's 2 KB share of the packed 128-column tile t
    dma_s(boff, wbase + (((size_t)t * nk + wkt) * W2_TILE_B + sub * 1024), ldsB + sb + (unsigned)(t * W2_TILE_B + sub * 1024));
  };

  f32x16 acc[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- fragment lanes: MFMA operand lane = row/column (lane & 31), k-block fk = lane >> 5
  const int frow = lane & 31, fk = lane >> 5;
  const int lrow0 = wave * 64 + frow, g0 = (lrow0 >> 2) & 3;  // rows +32 (second half) have the same swizzle
  const int aslot0 = 4 * lrow0 + ((2 * fk) ^ g0), aslot1 = 4 * lrow0 + ((2 * fk + 1) ^ g0);
  const uint4* const sA = smem;
  const uint4* const sBf = smem + NA * (A_STAGE_B / 16) + fk * BN + frow;
#ifdef GDRNPP2_TIMING_A_DIRECT
  constexpr bool ADIR = CONV == 0;     // linear form only
  const unsigned gvoff0 = (unsigned)min(m0 + lrow0, M - 1) * (unsigned)(K * 4) + (unsigned)(fk * 32);
  const unsigned gvoff1 = (unsigned)min(m0 + lrow0 + 32, M - 1) * (unsigned)(K * 4) + (unsigned)(fk * 32);
  [[maybe_unused]] ADirectScratch ads;
#else
  constexpr bool ADIR = false;
#endif

  auto load_half = [&](HalfSplit2& hs, int stage, int half) {
    hs.load(sA + stage * (A_STAGE_B / 16) + half * 128, aslot0, aslot1);
  };

  // fragment slot of column tile J, split plane P, relative to sBf (+ stage)
  auto bslot = [](int P, int J) { return (J >> 2) * W2_TILE_SLOTS + P * KB * BN + (J & 3) * 32; };

  // range check: per lane the sum of squares of the h halves it has multiplied (row frow / frow + 32 of the wave, k-blocks fk),
  // two accumulators per row half; -DGDRNPP2_NO_RANGE_CHECK: timing-only build without it
  [[maybe_unused]] float ssq[4] = {0.f, 0.f, 0.f, 0.f};

  // One k-tile, NS = 6 NJ slots (slot S: product group G = S / (2 NJ) in the order h*l, l*h, h*h; row half I, column tile J).
  // cur: split A fragments of k-tile kt; nxt: receives the split of k-tile kt+1 (its raw first half is already in nxt[0].x).
  // fbL holds the weight split l of kt on entry (dead after slot 2 NJ - 1, refilled with the split l of kt+1 behind the
  // barrier); fbH is read in slots 1, 3, .. NJ - 1 and used from slot 2 NJ.  BS: weight stage of kt (compile-time parity).
  // sa1 / sa2 / sa_wr: A stages of kt+1, of kt+2, and the one receiving kt+NA (= the stage kt has left).
  auto ktile = [&](int kt, HalfSplit2 (&cur)[2], HalfSplit2 (&nxt)[2], f16x8 (&fbL)[NJ], f16x8 (&fbH)[NJ], auto bs_, int sa1, int sa2,
                   int sa_wr) {
    constexpr int BS = decltype(bs_)::value;
    const uint4* const b = sBf + (PAIR ? (kt & 3) : BS) * (B_STAGE_B / 16);
    const uint4* const bn = sBf + (PAIR ? ((kt + 1) & 3) : (BS ^ 1)) * (B_STAGE_B / 16);
    const int kt_b = min(kt + (PAIR ? 2 : 1), nk - 1), kt_a = min(kt + NA, nk - 1);
    const unsigned sb_wr = (unsigned)((PAIR ? ((kt + 2) & 3) : (BS ^ 1)) * B_STAGE_B), sa_wr_b = (unsigned)(sa_wr * A_STAGE_B);
    static_for<0, NS>([&](auto s_) {
      constexpr int S = decltype(s_)::value;
      constexpr int G = S / (2 * NJ), I = (S % (2 * NJ)) / NJ, J = S % NJ;
      const f16x8 fa = (G == 1) ? cur[I].template frag<1>() : cur[I].template frag<0>();
      const f16x8 fb = (G == 0) ? fbL[J] : fbH[J];
      acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc[I][J], 0, 0, 0);
      // weight DMA of k-tile kt+1, then A DMA of k-tile kt+NA (the A pieces are the newest four loads at the wait)
#ifndef GDRNPP2_TIMING_NO_DMA
      if constexpr (S % 2 == 0 && S < 2 * NBP) dma_b(kt_b, sb_wr, std::integral_constant<int, S / 2>{});
      if constexpr (!ADIR && S % 2 == 0 && S >= 2 * NBP && S < 2 * NBP + 8) dma_a(kt_a, sa_wr_b, std::integral_constant<int, (S - 2 * NBP) / 2>{});
#endif
#ifdef GDRNPP2_TIMING_A_DIRECT
      if constexpr (ADIR) {
        const char* gsrc = reinterpret_cast<const char*>(A) + (size_t)min(kt + 2, nk - 1) * (BK * 4);
        if constexpr (S == 2 * NBP) gload_scratch(ads.d[0], ads.d[1], gvoff0, gsrc);
        if constexpr (S == 2 * NBP + 1) gload_scratch(ads.d[2], ads.d[3], gvoff1, gsrc);
        if constexpr (S == NS - 1) wait_scratch(ads);
      }
#endif
#ifndef GDRNPP2_TIMING_NO_BREAD   // timing-only builds (results invalid)
      if constexpr (S % 2 == 1 && S < NJ) {   // weight split h of kt
        constexpr int j0 = S - 1;
        fbH[j0] = __builtin_bit_cast(f16x8, b[bslot(0, j0)]);
        fbH[j0 + 1] = __builtin_bit_cast(f16x8, b[bslot(0, j0 + 1)]);
      }
#endif
#ifndef GDRNPP2_NO_RANGE_CHECK
      // sums of squares of cur's h halves in slots without split arithmetic (cur[..].h stays live through group h*h);
      // -DGDRNPP2_RC_SLOTS=a,b,c,d moves the four pairs for A/B timing (profiles/r04c_range_check_slots.txt)
#ifndef GDRNPP2_RC_SLOTS
#define GDRNPP2_RC_SLOTS 0, 2, 11, 12
#endif
      {
        constexpr int rc[4] = {GDRNPP2_RC_SLOTS};
        if constexpr (S == rc[0]) { ssq[2] = sumsq2(cur[1].h[0], ssq[2]); ssq[3] = sumsq2(cur[1].h[1], ssq[3]); }
        if constexpr (S == rc[1]) { ssq[2] = sumsq2(cur[1].h[2], ssq[2]); ssq[3] = sumsq2(cur[1].h[3], ssq[3]); }
        if constexpr (S == rc[2]) { ssq[0] = sumsq2(cur[0].h[0], ssq[0]); ssq[1] = sumsq2(cur[0].h[1], ssq[1]); }
        if constexpr (S == rc[3]) { ssq[0] = sumsq2(cur[0].h[2], ssq[0]); ssq[1] = sumsq2(cur[0].h[3], ssq[1]); }
      }
#endif
      constexpr int S1 = NJ == 4 ? 13 : 20;   // first split slot of the second half (its raw read four slots earlier)
      if constexpr (!ADIR && S == S1 - 4) load_half(nxt[1], sa1, 1);   // APRE: nxt[1].h / .l directly (dead since the previous k-tile)
      // split of the next k-tile: first half in slots 3..10, second half in slots S1..S1+7
#ifndef GDRNPP2_TIMING_NO_SPLIT
      if constexpr (S >= 3 && S < 11) nxt[0].template step<S - 3>();
      if constexpr (S >= S1 && S < S1 + 8) nxt[1].template step<S - S1>();
#else
      if constexpr (S == 3) { nxt[0].h[0] = __float_as_uint(nxt[0].x[0]); nxt[0].h[1] = __float_as_uint(nxt[0].x[1]); nxt[0].h[2] = __float_as_uint(nxt[0].x[2]); nxt[0].h[3] = __float_as_uint(nxt[0].x[3]);
                              nxt[0].l[0] = __float_as_uint(nxt[0].x[4]); nxt[0].l[1] = __float_as_uint(nxt[0].x[5]); nxt[0].l[2] = __float_as_uint(nxt[0].x[6]); nxt[0].l[3] = __float_as_uint(nxt[0].x[7]); }
      if constexpr (S == S1) { nxt[1].h[0] = __float_as_uint(nxt[1].x[0]); nxt[1].h[1] = __float_as_uint(nxt[1].x[1]); nxt[1].h[2] = __float_as_uint(nxt[1].x[2]); nxt[1].h[3] = __float_as_uint(nxt[1].x[3]);
                               nxt[1].l[0] = __float_as_uint(nxt[1].x[4]); nxt[1].l[1] = __float_as_uint(nxt[1].x[5]); nxt[1].l[2] = __float_as_uint(nxt[1].x[6]); nxt[1].l[3] = __float_as_uint(nxt[1].x[7]); }
#endif
      if constexpr (S == SB) {
#ifndef GDRNPP2_TIMING_NO_SYNC
        wait_vmcnt<4>();
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): every LDS read of the stages about to be refilled has returned
        if constexpr (!PAIR || BS == 1) __builtin_amdgcn_s_barrier();
#endif
      }
      // behind the barrier: weight split l of k-tile kt+1 (fbL is dead since slot 7) and the raw first half of k-tile kt+2
      // (cur[0] is nxt[0] of the next k-tile; only cur[0].h / .l are still in use)
      // (APRE: the A read goes first — it IS the operand of slot 0 of the next k-tile, cur[0].h / .l are dead since slots 4 NJ + NJ - 1 / 3 NJ - 1)
      constexpr int SL = APRE ? SB + 1 : SB;         // slot before the first weight read
#ifndef GDRNPP2_TIMING_NO_BREAD
      if constexpr (S > SL && S <= SL + NJ / 2) {
        constexpr int j0 = 2 * (S - SL - 1);
        fbL[j0] = __builtin_bit_cast(f16x8, bn[bslot(1, j0)]);
        fbL[j0 + 1] = __builtin_bit_cast(f16x8, bn[bslot(1, j0 + 1)]);
      }
#endif
      if constexpr (!ADIR && S == (APRE ? SB + 1 : SB + NJ / 2 + 1)) load_half(cur[0], sa2, 0);
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  // ---- prologue: k-tiles 0 .. 2 of A and k-tile 0 of the weights; split k-tile 0; first fragments of the loop
  HalfSplit2 f0[2], f1[2];
  f16x8 fbL[NJ], fbH[NJ];
  if constexpr (!ADIR) {
    static_for<0, 4>([&](auto c) { dma_a(0, 0u, c); });
    static_for<0, NBP>([&](auto c) { dma_b(0, 0u, c); });
    if constexpr (PAIR) static_for<0, NBP>([&](auto c) { dma_b(min(1, nk - 1), (unsigned)B_STAGE_B, c); });
    static_for<0, 4>([&](auto c) { dma_a(min(1, nk - 1), (unsigned)A_STAGE_B, c); });
    static_for<0, 4>([&](auto c) { dma_a(min(2, nk - 1), 2u * A_STAGE_B, c); });
    wait_vmcnt<4>();
    __builtin_amdgcn_s_barrier();
    load_half(f0[0], 0, 0);
    load_half(f0[1], 0, 1);
  } else {   // timing-only: the x registers start from finite values and are never refilled
    static_for<0, NBP>([&](auto c) { dma_b(0, 0u, c); });
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int e = 0; e < 8; ++e) { f0[0].x[e] = f0[1].x[e] = f1[0].x[e] = f1[1].x[e] = 1.0f + 0.01f * (float)(e + lane); }
  }
  static_for<0, 8>([&](auto s) { f0[0].template step<decltype(s)::value>(); f0[1].template step<decltype(s)::value>(); });
#pragma unroll
  for (int j = 0; j < NJ; ++j) fbL[j] = __builtin_bit_cast(f16x8, sBf[bslot(1, j)]);
  if constexpr (!ADIR) load_half(f1[0], 1, 0);

  // ---- main loop, two k-tiles per trip (nk is even: K % 32 == 0)
  int sa = 0;  // kt % NA
  for (int kt = 0; kt < nk; kt += 2) {
    const int sa1 = sa + 1 == NA ? 0 : sa + 1, sa2 = sa1 + 1 == NA ? 0 : sa1 + 1, sa3 = sa2 + 1 == NA ? 0 : sa2 + 1;
    ktile(kt, f0, f1, fbL, fbH, std::integral_constant<int, 0>{}, sa1, sa2, sa);
    ktile(kt + 1, f1, f0, fbL, fbH, std::integral_constant<int, 1>{}, sa2, sa3, sa1);
    sa = sa2;
  }
  wait_vmcnt<0>();               // the clamped A pieces of the last trip are still in flight
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_s_barrier();  // every wave's stages are dead: the epilogue reuses them

  // ---- epilogue: per wave one 16x64 slice at a time through LDS, written back row-wise as float4 (as in gemm_split.hip)
  float* T = reinterpret_cast<float*>(smem) + wave * 16 * 65;
  const int c4 = (lane & 15) * 4;
  bool bad = false;
#pragma unroll
  for (int jh = 0; jh < NJ / 2; ++jh) {
    const int nb = n0 + jh * 64 + c4;
    const float4 bv = bias ? *reinterpret_cast<const float4*>(bias + nb) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 gv = make_float4(1.f, 1.f, 1.f, 1.f);
    if (EPI == EPI_SCALE_RES) gv = *reinterpret_cast<const float4*>(gamma + nb);
    double gs = 0.0, gss = 0.0;  // GNS: this lane's four columns over its 16 rows
#pragma unroll
    for (int ih = 0; ih < 4; ++ih) {
      const int i = ih >> 1, h = ih & 1;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 8; ++r)
          T[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 65 + j * 32 + (lane & 31)] = acc[i][jh * 2 + j][h * 8 + r];
      __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int row = rr * 4 + (lane >> 4);
        const float* t = T + row * 65 + c4;
        float4 v = make_float4(t[0] * wsc + bv.x, t[1] * wsc + bv.y, t[2] * wsc + bv.z, t[3] * wsc + bv.w);
        const int grow = m0 + wave * 64 + i * 32 + h * 16 + row;
        if (grow >= M) continue;
        const size_t off = (size_t)grow * N + nb;
        if (GNS) {
          gs += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
          gss += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
        }
        if (EPI == EPI_GELU) { v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w); }
        if (EPI == EPI_SCALE_RES) {
          const float4 rs = *reinterpret_cast<const float4*>(resid + off);
          v.x = rs.x + gv.x * v.x; v.y = rs.y + gv.y * v.y; v.z = rs.z + gv.z * v.z; v.w = rs.w + gv.w * v.w;
        }
        constexpr unsigned kInfNan = 0x203u;   // v_cmp_class_f32: signalling / quiet NaN, -inf, +inf
        bad |= __builtin_amdgcn_class(v.x, kInfNan) | __builtin_amdgcn_class(v.y, kInfNan) |
               __builtin_amdgcn_class(v.z, kInfNan) | __builtin_amdgcn_class(v.w, kInfNan);
        if (EPI != EPI_SCALE_RES && c_rows) {   // lanes 2k / 2k + 1 hold columns 8k .. 8k + 3 / 8k + 4 .. 8k + 7 of the same row
          const uint4 o = f16x2_rows_quad(v.x, v.y, v.z, v.w, lane & 1);
          const f32x4v t4 = {__uint_as_float(o.x), __uint_as_float(o.y), __uint_as_float(o.z), __uint_as_float(o.w)};
          __builtin_nontemporal_store(t4, reinterpret_cast<f32x4v*>(C + off));
        } else {
          const f32x4v t4 = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(t4, reinterpret_cast<f32x4v*>(C + off));
        }
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);
    }
    if (GNS) {
      // a group's 8 channels are the column quads of lanes 2k, 2k+1; its 64 rows sit in the four 16-lane row groups
      gs += __shfl_xor(gs, 1, 64);   gss += __shfl_xor(gss, 1, 64);
      gs += __shfl_xor(gs, 16, 64);  gss += __shfl_xor(gss, 16, 64);
      gs += __shfl_xor(gs, 32, 64);  gss += __shfl_xor(gss, 32, 64);
      if ((lane & 0x31) == 0) {
        const int hw = cg.H * cg.W, img = m0 / hw, mt = (m0 - img * hw) >> 8;
        const int g = ((n0 + jh * 64) >> 3) + (lane >> 1);
        double* o = gn.part + (((size_t)img * (4 * gn.tiles_per_img) + 4 * mt + wave) * gn.G + g) * 2;
        o[0] = gs;
        o[1] = gss;
      }
    }
  }
  // range verdict of the wave's 64 A rows: lanes l and l ^ 32 hold the two k-block halves of rows frow and frow + 32
  bool small = false;
#ifndef GDRNPP2_NO_RANGE_CHECK
  {
    float s0 = ssq[0] + ssq[1], s1 = ssq[2] + ssq[3];
    s0 += __shfl_xor(s0, 32, 64);
    s1 += __shfl_xor(s1, 32, 64);
    const float thr = (float)K * 0x1p-8f;                         // rms 2^-4 over the K elements of the row
    small = (s0 > 0.f && s0 < thr) || (s1 > 0.f && s1 < thr);
    bad |= !(s0 < __builtin_inff()) || !(s1 < __builtin_inff());  // an inf / NaN element of A
  }
#endif
  const int word = (__builtin_amdgcn_ballot_w64(bad) != 0 ? GDRNPP_SPLIT2_NONFINITE : 0) |
                   (__builtin_amdgcn_ballot_w64(small) != 0 ? GDRNPP_SPLIT2_SMALL_ROWS : 0);
  if (word && lane == 0) atomicOr(range_flag ? range_flag : &g_split2_range_word, word);
}

template <int EPI, int CONV, bool GNS, int NJ, bool APRE = false>
int launch_nj(const float* A, const uint4* Wp, const float* bias, const float* gamma, const float* resid, float* C, int M, int N,
              int K, ConvGeom cg, int panel, GnStats2 gn, int* range_flag, hipStream_t st, const char* what, int c_rows = 0) {
#ifndef GDRNPP2_SINGLE_BARRIER
  constexpr int lds_bytes = NA * A_STAGE_B + (NJ == 4 ? 4 : 2) * (NJ / 4) * W2_TILE_B;   // NJ = 4: 80 KB, two workgroups fill the CU's 160 KB
#else
  constexpr int lds_bytes = NA * A_STAGE_B + 2 * (NJ / 4) * W2_TILE_B;
#endif
  const int rc = gdrnpp::ensure_dynamic_lds((const void*)gemm_split2_pipe_kernel<EPI, CONV, GNS, NJ, APRE>, lds_bytes);
  if (rc) return rc;
  const long tiles = (long)((M + 255) / 256) * (N / (NJ * 32));
  GDRNPP_REQUIRE(tiles < (1l << 30), GDRNPP_ELIMIT, "%s: grid too large", what);
  hipLaunchKernelGGL((gemm_split2_pipe_kernel<EPI, CONV, GNS, NJ, APRE>), dim3((unsigned)tiles), dim3(256), lds_bytes, st, A, Wp, bias, gamma,
                     resid, C, M, N, K, cg, panel, gn, range_flag, c_rows);
  return gdrnpp::check_launch(what);
}

// 256 x 256 block tiles (NJ = 8) are kept for A/B only (option split2_wide = 1, N % 256 == 0): bitwise equal, and measured slower
// or equal on every ConvNeXt-B MLP shape of 128 ROIs (36 blocks 18.5 vs 17.0 ms; stage-2 fc2 207 vs 207 us, stage-3 fc2 246 vs
// 189 us, stage-0 fc1 623 vs 533 us: one wave per SIMD has nobody to hide its LDS / DMA latencies behind)
template <int EPI, int CONV, bool GNS>
int launch_one(const float* A, const uint4* Wp, const float* bias, const float* gamma, const float* resid, float* C, int M, int N,
               int K, ConvGeom cg, int panel, GnStats2 gn, int* range_flag, hipStream_t st, const char* what, int a_rows = 0, int c_rows = 0) {
  if constexpr (CONV == 0 && EPI != EPI_BIAS) {   // the f16x2-rows forms exist for the two MLP layers (GELU / scale + residual)
    if (a_rows) return launch_nj<EPI, CONV, GNS, 4, true>(A, Wp, bias, gamma, resid, C, M, N, K, cg, panel, gn, range_flag, st, what, c_rows);
  }
  const int opt = gdrnpp::option_split2_wide();
  const bool wide = N % 256 == 0 && opt == 1;
  if (wide) return launch_nj<EPI, CONV, GNS, 8>(A, Wp, bias, gamma, resid, C, M, N, K, cg, panel, gn, range_flag, st, what, c_rows);
  return launch_nj<EPI, CONV, GNS, 4>(A, Wp, bias, gamma, resid, C, M, N, K, cg, panel, gn, range_flag, st, what, c_rows);
}

}  // namespace

extern "C" size_t gdrnpp_pack_weight_f16x2_bytes(int N, int K) {
  return (N > 0 && K > 0) ? (size_t)N * (size_t)K * 4 + 16 : 0;
}

extern "C" int gdrnpp_pack_weight_f16x2(const float* W, void* packed, int N, int K, void* stream) {
  GDRNPP_REQUIRE(W && packed, GDRNPP_EINVAL, "gdrnpp_pack_weight_f16x2: null pointer");
  GDRNPP_REQUIRE(N > 0 && K > 0 && N % BN == 0 && K % 32 == 0, GDRNPP_ELIMIT,
                 "gdrnpp_pack_weight_f16x2: N=%d K=%d must be multiples of %d/32", N, K, BN);
  hipStream_t st = (hipStream_t)stream;
  unsigned* trailer = reinterpret_cast<unsigned*>(reinterpret_cast<uint4*>(packed) + (size_t)(N / BN) * (K / BK) * W2_TILE_SLOTS);
  GDRNPP_HIP_TRY(hipMemsetAsync(trailer, 0, 16, st));
  const long n = (long)N * K;
  hipLaunchKernelGGL(amax_kernel, dim3((unsigned)min((n + 1023) / 1024, 1024l)), dim3(256), 0, st, W, n, trailer);
  const long threads = (long)N * (K / 8);
  hipLaunchKernelGGL(pack_weight2_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, W, (uint4*)packed, N, K);
  hipLaunchKernelGGL(weight_rows_range_kernel, dim3((unsigned)N), dim3(256), 0, st, W, trailer, trailer + 3, K);   // behind the pack: it writes trailer[3] = 0
  return gdrnpp::check_launch("gdrnpp_pack_weight_f16x2");
}

extern "C" int gdrnpp_linear_f32_split2_rows(const float* A, const void* W_packed, const float* bias, const float* gamma,
                                             const float* resid, float* C, int M, int N, int K, int epilogue, int rows,
                                             int* range_flag, void* stream) {
  GDRNPP_REQUIRE(A && W_packed && C, GDRNPP_EINVAL, "gdrnpp_linear_f32_split2: null pointer");
  GDRNPP_REQUIRE(M > 0 && N > 0 && K > 0 && N % BN == 0 && K % 32 == 0, GDRNPP_ELIMIT,
                 "gdrnpp_linear_f32_split2: N=%d K=%d must be multiples of %d/32 (M=%d is free)", N, K, BN, M);
  GDRNPP_REQUIRE((unsigned long long)M * (unsigned long long)K * 4ull < (1ull << 32), GDRNPP_ELIMIT,
                 "gdrnpp_linear_f32_split2: M*K*4 must stay below 4 GiB (32-bit lane offsets)");
  GDRNPP_REQUIRE(epilogue >= 0 && epilogue <= 2, GDRNPP_EINVAL, "gdrnpp_linear_f32_split2: epilogue=%d", epilogue);
  GDRNPP_REQUIRE(epilogue != EPI_SCALE_RES || (gamma && resid), GDRNPP_EINVAL,
                 "gdrnpp_linear_f32_split2: scale+residual epilogue needs gamma and resid");
  GDRNPP_REQUIRE((rows & ~(GDRNPP_A_F16X2_ROWS | GDRNPP_C_F16X2_ROWS)) == 0, GDRNPP_EINVAL, "gdrnpp_linear_f32_split2_rows: rows=%d", rows);
  GDRNPP_REQUIRE(!(rows & GDRNPP_C_F16X2_ROWS) || epilogue != EPI_SCALE_RES, GDRNPP_EINVAL,
                 "gdrnpp_linear_f32_split2_rows: the scale+residual epilogue writes fp32");
  GDRNPP_REQUIRE(!(rows & GDRNPP_A_F16X2_ROWS) || epilogue != EPI_BIAS, GDRNPP_ELIMIT,
                 "gdrnpp_linear_f32_split2_rows: an f16x2-rows A needs the GELU or the scale+residual epilogue (the two MLP layers)");
  // wide layers walk the tiles in panels (gemm_split_pipe.hip: launch_split_pipe); the fp16x2 image is 4 bytes per weight
  const int panel = (N / BN >= 8 && (long)N * K * 4 > (2l << 20)) ? gdrnpp::option_split_gemm_panel() : 0;
  const ConvGeom cg{0, 0, 0, 0, 0, 0, 0, 0, 0};
  const GnStats2 gn{nullptr, 0, 0};
  hipStream_t st = (hipStream_t)stream;
  const uint4* Wp = (const uint4*)W_packed;
  const char* what = "gdrnpp_linear_f32_split2";
  const int ar = rows & GDRNPP_A_F16X2_ROWS, cr = (rows & GDRNPP_C_F16X2_ROWS) ? 1 : 0;
  if (epilogue == EPI_BIAS) return launch_one<EPI_BIAS, 0, false>(A, Wp, bias, gamma, resid, C, M, N, K, cg, panel, gn, range_flag, st, what, 0, cr);
  if (epilogue == EPI_GELU) return launch_one<EPI_GELU, 0, false>(A, Wp, bias, gamma, resid, C, M, N, K, cg, panel, gn, range_flag, st, what, ar, cr);
  return launch_one<EPI_SCALE_RES, 0, false>(A, Wp, bias, gamma, resid, C, M, N, K, cg, panel, gn, range_flag, st, what, ar, 0);
}

extern "C" int gdrnpp_linear_f32_split2(const float* A, const void* W_packed, const float* bias, const float* gamma,
                                        const float* resid, float* C, int M, int N, int K, int epilogue, int* range_flag,
                                        void* stream) {
  return gdrnpp_linear_f32_split2_rows(A, W_packed, bias, gamma, resid, C, M, N, K, epilogue, 0, range_flag, stream);
}

extern "C" int gdrnpp_conv3x3_f32_split2(const float* x_nhwc, const void* W_packed, const float* bias, float* y_nhwc,
                                         double* gn_partials, int n_img, int H, int W, int Cin, int Cout, int groups, int epilogue,
                                         int* range_flag, void* stream) {
  GDRNPP_REQUIRE(x_nhwc && W_packed && y_nhwc, GDRNPP_EINVAL, "gdrnpp_conv3x3_f32_split2: null pointer");
  GDRNPP_REQUIRE(n_img > 0 && H > 0 && W > 0 && H < 32768 && W < 32768 && Cin > 0 && Cout > 0, GDRNPP_EINVAL,
                 "gdrnpp_conv3x3_f32_split2: bad shape");
  const long M = (long)n_img * H * W;
  GDRNPP_REQUIRE(M < (1l << 31) && Cout % BN == 0 && Cin % 32 == 0, GDRNPP_ELIMIT,
                 "gdrnpp_conv3x3_f32_split2: Cout=%d Cin=%d must be multiples of %d/32", Cout, Cin, BN);
  GDRNPP_REQUIRE(epilogue == EPI_BIAS || epilogue == EPI_GELU, GDRNPP_EINVAL, "gdrnpp_conv3x3_f32_split2: epilogue=%d", epilogue);
  const ConvGeom cg{H, W, Cin, H, W, 3, 1, 1, 0};
  hipStream_t st = (hipStream_t)stream;
  const uint4* Wp = (const uint4*)W_packed;
  const char* what = "gdrnpp_conv3x3_f32_split2";
  if (gn_partials) {
    GDRNPP_REQUIRE(epilogue == EPI_BIAS && (H * W) % 256 == 0 && groups > 0 && Cout == 8 * groups, GDRNPP_ELIMIT,
                   "gdrnpp_conv3x3_f32_split2: GroupNorm statistics need the plain epilogue, H*W %% 256 == 0 and 8 channels per group "
                   "(H*W=%d Cout=%d groups=%d)", H * W, Cout, groups);
    return launch_one<EPI_BIAS, 1, true>(x_nhwc, Wp, bias, nullptr, nullptr, y_nhwc, (int)M, Cout, 9 * Cin, cg, 0,
                                         GnStats2{gn_partials, groups, (H * W) / 256}, range_flag, st, what);
  }
  const GnStats2 gn{nullptr, 0, 0};
  if (epilogue == EPI_GELU)
    return launch_one<EPI_GELU, 1, false>(x_nhwc, Wp, bias, nullptr, nullptr, y_nhwc, (int)M, Cout, 9 * Cin, cg, 0, gn, range_flag, st, what);
  return launch_one<EPI_BIAS, 1, false>(x_nhwc, Wp, bias, nullptr, nullptr, y_nhwc, (int)M, Cout, 9 * Cin, cg, 0, gn, range_flag, st, what);
}

extern "C" int gdrnpp_conv2d_f32_split2(const float* x_nhwc, const void* W_packed, const float* bias, float* y_nhwc, int n_img,
                                        int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int epilogue,
                                        int* range_flag, void* stream) {
  GDRNPP_REQUIRE(x_nhwc && W_packed && y_nhwc, GDRNPP_EINVAL, "gdrnpp_conv2d_f32_split2: null pointer");
  GDRNPP_REQUIRE(n_img > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && H < 32768 && W < 32768 && KH > 0 && KW > 0 && KH * KW <= 32 &&
                     stride > 0 && pad >= 0 && pad < KH && pad < KW,
                 GDRNPP_EINVAL, "gdrnpp_conv2d_f32_split2: bad shape (at most 32 taps)");
  if (KH == 3 && KW == 3 && stride == 1 && pad == 1)
    return gdrnpp_conv3x3_f32_split2(x_nhwc, W_packed, bias, y_nhwc, nullptr, n_img, H, W, Cin, Cout, 0, epilogue, range_flag, stream);
  const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
  GDRNPP_REQUIRE(OH > 0 && OW > 0 && (OH - 1) * stride < H && (OW - 1) * stride < W, GDRNPP_EINVAL,
                 "gdrnpp_conv2d_f32_split2: empty output or anchor pixel outside the image");
  const long M = (long)n_img * OH * OW;
  GDRNPP_REQUIRE(M < (1l << 31) && Cout % BN == 0 && Cin % 32 == 0, GDRNPP_ELIMIT,
                 "gdrnpp_conv2d_f32_split2: Cout=%d Cin=%d must be multiples of %d/32", Cout, Cin, BN);
  GDRNPP_REQUIRE(epilogue == EPI_BIAS || epilogue == EPI_GELU, GDRNPP_EINVAL, "gdrnpp_conv2d_f32_split2: epilogue=%d", epilogue);
  const ConvGeom cg{H, W, Cin, OH, OW, KW, stride, pad, 0};
  const GnStats2 gn{nullptr, 0, 0};
  hipStream_t st = (hipStream_t)stream;
  const uint4* Wp = (const uint4*)W_packed;
  const char* what = "gdrnpp_conv2d_f32_split2";
  if (epilogue == EPI_GELU)
    return launch_nj<EPI_GELU, 2, false, 4>(x_nhwc, Wp, bias, nullptr, nullptr, y_nhwc, (int)M, Cout, KH * KW * Cin, cg, 0, gn, range_flag, st, what);
  return launch_nj<EPI_BIAS, 2, false, 4>(x_nhwc, Wp, bias, nullptr, nullptr, y_nhwc, (int)M, Cout, KH * KW * Cin, cg, 0, gn, range_flag, st, what);
}

// The library's own sticky range word (launches with range_flag == NULL): *word = the bits raised since the last reset
// (synchronises the stream).  Callers running several streams / host threads / layers pass words of their own per launch.
extern "C" int gdrnpp_split2_range_word(int* flag, int reset, void* stream) {
  GDRNPP_REQUIRE(flag, GDRNPP_EINVAL, "gdrnpp_split2_range_word: null pointer");
  hipStream_t st = (hipStream_t)stream;
  int v = 0;
  GDRNPP_HIP_TRY(hipMemcpyFromSymbolAsync(&v, HIP_SYMBOL(g_split2_range_word), sizeof(int), 0, hipMemcpyDeviceToHost, st));
  GDRNPP_HIP_TRY(hipStreamSynchronize(st));
  if (reset && v) {
    const int zero = 0;
    GDRNPP_HIP_TRY(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_split2_range_word), &zero, sizeof(int), 0, hipMemcpyHostToDevice, st));
    GDRNPP_HIP_TRY(hipStreamSynchronize(st));
  }
  *flag = v;
  return 0;
}
