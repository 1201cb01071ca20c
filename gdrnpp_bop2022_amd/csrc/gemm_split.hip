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
#include "gemm_split.hpp"

namespace {

using namespace gdrnpp::splitgemm;

#ifndef SPLIT_OCC
#define SPLIT_OCC 2
#endif

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
  const int ntaps = CONV ? nk_total / cpt : 1, cps = (cpt & 1) ? 1 : 2;  // conv k-tile order: gemm_split.hpp
  const uint4* Wg = Wp + ((size_t)tile_n * nk_total + kt0) * W_TILE_SLOTS + tid;
  struct Stage { float4 a[MI]; uint4 b0, b1, b2; };
  auto gload = [&](int kt) {
    Stage r;
    int wkt = kt;
    if (CONV) {
      // the loads are unconditional (address clamped to the centre pixel, value zeroed afterwards): a predicated load
      // would make the outstanding-load count unknown to the compiler and collapse the software pipeline
      const int ka = kt0 + kt;   // absolute k-tile (split-K chunks of a convolution start mid-sequence)
      const int sup = ka / (ntaps * cps), rem = ka - sup * (ntaps * cps), tap = rem / cps;
      const int chunk = sup * cps + (rem - tap * cps), c0 = chunk * BK;
      wkt = tap * cpt + chunk - kt0;   // Wg already points at weight tile kt0
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
    const uint4* w = Wg + (long)wkt * W_TILE_SLOTS;
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
    // product order shared by all split-GEMM kernels (gemm_split.hpp); the accumulators rotate so no MFMA waits on
    // its predecessor
    GDRNPP_SPLIT_PRODUCT_ORDER
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
// LDS-DMA form of the same GEMM (gdrnpp_set_option("split_gemm_glds", 1)): 256x128x16 block tile, 4 waves stacked along M
// (each wave 64 rows x 128 columns = 2 x 4 MFMA tiles), NO register staging and NO ds_write in the k-loop:
//   A  stays fp32 in HBM and goes HBM -> LDS by global_load_lds_dwordx4 (16 B per lane, four lanes cover one 64-byte row
//      segment, so the loads stay coalesced).  The LDS image is lane-linear (slot = 4*row + p); which 16-byte chunk q of
//      the row segment a lane fetches is swizzled, q = p ^ ((row >> 2) & 3), so that the 16 lanes of a ds_read_b128 lane
//      group hit 16 different bank quads when a fragment (32 rows x 8 k) is read back.  The exact 3-way bf16 split is
//      done at fragment-read time in registers; with the waves stacked along M every A row is read and split by exactly
//      one wave (same VALU work as splitting before the LDS store), and each wave stages precisely the rows it consumes;
//   W  packed image (already the LDS image) by three global_load_lds_dwordx4 per wave and k-tile;
//   two LDS stages of 28 KB: the DMA of k-tile t+1 is issued before the MFMAs of k-tile t and waited for (vmcnt(0), the
//      only VMEM traffic of the loop) in front of the one barrier per k-tile; two workgroups per CU.
//   CONV taps outside the image fetch from a zero page instead of being predicated.
// The DMA is issued from inline asm (the compiler would otherwise drain it in front of every LDS read it cannot prove
// disjoint); M0 is set and restored inside the statement (cdna_hip_programming.md §5.7).
// ----------------------------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(64))) float g_zero_page[16];

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}

__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p;
}

constexpr int GA_SLOTS = 256 * 4;   // fp32 A image of one stage: 256 rows x 4 chunks of 16 B
constexpr int GB_SLOTS = 3 * KB * BN;  // packed weight tile image

// GroupNorm statistics of the result, taken in the epilogue of the convolution that produces it (GNS): every wave writes
// the fp64 (sum, sum of squares) of its 64 rows x 8-channel groups to part[image][P][G][2], P = 4 * (256-row tiles per
// image), slot 4 * tile + wave — the layout gn_apply_kernel (net_kernels.hip) reduces, so the separate statistics pass over
// the stored tensor is not needed.  Requires 8 channels per group and images of a multiple of 256 pixels.
struct GnStats { double* part; int G; int tiles_per_img; };

template <int EPI, int CONV, bool GNS = false>
__global__ __launch_bounds__(256, 2) void gemm_split_glds_kernel(const float* __restrict__ A, const uint4* __restrict__ Wp,
                                                                 const float* __restrict__ bias,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ resid, float* __restrict__ C,
                                                                 int M, int N, int K, ConvGeom cg, GnStats gn) {
  __shared__ uint4 sA[2 * GA_SLOTS];
  __shared__ uint4 sB[2 * GB_SLOTS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntn = N / BN;
  const int nwg = gridDim.x, xcd = blockIdx.x & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
  const int tile_m = tile / ntn, tile_n = tile % ntn;
  const int m0 = tile_m * 256, n0 = tile_n * BN;
  const int nk = K / BK;

  // A staging: DMA c (0..3) of this wave fills slots (wave*4 + c)*64 + lane = rows wave*64 + c*16 + lane/4
  const int prow = lane >> 2, pq = lane & 3;
  const float* ap[4];
  int pyx[4], cpt = 1;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int lrow = wave * 64 + c * 16 + prow;
    const int q = pq ^ ((lrow >> 2) & 3);
    const int arow = min(m0 + lrow, M - 1);
    if (CONV) {
      const int OH = CONV == 1 ? cg.H : cg.OH, OW = CONV == 1 ? cg.W : cg.OW, stride = CONV == 1 ? 1 : cg.stride;
      const int img = arow / (OH * OW), pp = arow - img * (OH * OW);
      const int iy = (pp / OW) * stride, ix = (pp % OW) * stride;
      pyx[c] = (iy << 16) | ix;
      ap[c] = A + (((size_t)img * cg.H + iy) * cg.W + ix) * cg.C + q * 4;
    } else {
      pyx[c] = 0;
      ap[c] = A + (size_t)arow * K + q * 4;
    }
  }
  if (CONV) cpt = cg.C / BK;
  const int ntaps = CONV ? nk / cpt : 1, cps = (cpt & 1) ? 1 : 2;  // conv k-tile order: gemm_split.hpp
  const uint4* Wg = Wp + (size_t)tile_n * nk * W_TILE_SLOTS + (wave * 3) * 64 + lane;
  const unsigned ldsA = lds_addr(sA) + (unsigned)(wave * 4) * 1024u;
  const unsigned ldsB = lds_addr(sB) + (unsigned)(wave * 3) * 1024u;

  auto issue = [&](int kt, int stage) {
    const unsigned da = ldsA + (unsigned)stage * (GA_SLOTS * 16u), db = ldsB + (unsigned)stage * (GB_SLOTS * 16u);
    int wkt = kt;
    if (CONV) {
      const int sup = kt / (ntaps * cps), rem = kt - sup * (ntaps * cps), tap = rem / cps;
      const int chunk = sup * cps + (rem - tap * cps), c0 = chunk * BK;
      wkt = tap * cpt + chunk;
      const int KW = CONV == 1 ? 3 : cg.KW, pad = CONV == 1 ? 1 : cg.pad;
      const int ky = tap / KW, dy = ky - pad, dx = tap - ky * KW - pad;
      const int off = (dy * cg.W + dx) * cg.C + c0;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const bool ok = (unsigned)((pyx[c] >> 16) + dy) < (unsigned)cg.H && (unsigned)((pyx[c] & 0xffff) + dx) < (unsigned)cg.W;
        glds16(ok ? (const void*)(ap[c] + off) : (const void*)g_zero_page, da + c * 1024u);
      }
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) glds16(ap[c] + kt * BK, da + c * 1024u);
    }
    const uint4* w = Wg + (size_t)wkt * W_TILE_SLOTS;
#pragma unroll
    for (int c = 0; c < 3; ++c) glds16(w + c * 64, db + c * 1024u);
  };

  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frow = lane & 31, fk = lane >> 5;
  // fragment slots of this lane in the A image: row wave*64 + i*32 + frow, chunks 2*fk and 2*fk + 1 (swizzled)
  int aslot[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int lrow = wave * 64 + i * 32 + frow, g = (lrow >> 2) & 3;
    aslot[i][0] = 4 * lrow + ((2 * fk) ^ g);
    aslot[i][1] = 4 * lrow + ((2 * fk + 1) ^ g);
  }
  auto compute = [&](int stage) {
    const uint4* a = sA + stage * GA_SLOTS;
    const uint4* b = sB + stage * GB_SLOTS + fk * BN + frow;
    bf16x8 fb[3][4];
    float4 ra[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      ra[i][0] = __builtin_bit_cast(float4, a[aslot[i][0]]);
      ra[i][1] = __builtin_bit_cast(float4, a[aslot[i][1]]);
    }
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[s][j] = __builtin_bit_cast(bf16x8, b[s * KB * BN + j * 32]);
    GDRNPP_SPLIT_PRODUCT_ORDER
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const Split3 p0 = split_pair(ra[i][0].x, ra[i][0].y), p1 = split_pair(ra[i][0].z, ra[i][0].w);
      const Split3 p2 = split_pair(ra[i][1].x, ra[i][1].y), p3 = split_pair(ra[i][1].z, ra[i][1].w);
      bf16x8 fa[3];
      fa[0] = __builtin_bit_cast(bf16x8, make_uint4(p0.h, p1.h, p2.h, p3.h));
      fa[1] = __builtin_bit_cast(bf16x8, make_uint4(p0.m, p1.m, p2.m, p3.m));
      fa[2] = __builtin_bit_cast(bf16x8, make_uint4(p0.l, p1.l, p2.l, p3.l));
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[TA[t]], fb[TB[t]][j], acc[i][j], 0, 0, 0);
    }
  };

  issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = 0; kt < nk; kt += 2) {   // nk is even (K % 32 == 0); the tail re-stages the last tile into the idle buffer
    issue(kt + 1, 1);
    compute(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    issue(min(kt + 2, nk - 1), 0);
    compute(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // epilogue: as in gemm_split_kernel, per wave one 16x64 slice at a time through LDS (the A images are dead)
  static_assert(2 * GA_SLOTS * sizeof(uint4) >= 4 * 16 * 65 * sizeof(float), "epilogue staging fits the A images");
  float* T = reinterpret_cast<float*>(sA) + wave * 16 * 65;
  const int c4 = (lane & 15) * 4;
#pragma unroll
  for (int jh = 0; jh < 2; ++jh) {
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
        float4 v = make_float4(t[0] + bv.x, t[1] + bv.y, t[2] + bv.z, t[3] + bv.w);
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
        { const f32x4v t4 = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(t4, reinterpret_cast<f32x4v*>(C + off)); }
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
  const bool fast3x3 = CONV && cg.KW == 3 && cg.stride == 1 && cg.pad == 1 && K == 9 * cg.C;
  const long tiles256 = (long)((M + 255) / 256) * (N / BN);
  GDRNPP_REQUIRE(tiles256 < (1l << 30), GDRNPP_ELIMIT, "%s: grid too large", what);
  // 256-row tiles when they still give every CU a workgroup (option split_gemm_big_tiles, 256; measured +3 % / +7 % on the
  // stage-2 MLP shapes over 128x128 tiles at three workgroups per CU; end to end against a threshold of 512: +0.4 % at 128
  // ROIs, +2.2 % at 64, +0.1 % at 32; 128: -0.5 % at 32, -0.9 % at 16); gdrnpp_set_option("split_gemm_mi4", 0/1) forces the choice.
  const int force = gdrnpp::option_split_gemm_mi4();
  const bool big = force >= 0 ? force == 1 : tiles256 >= gdrnpp::option_split_gemm_big_tiles();
  if (big && gdrnpp::option_split_gemm_pipe() && (!CONV || gdrnpp::option_split_gemm_pipe_conv())) {   // software-pipelined LDS-DMA kernel
    const int rc = launch_split_pipe(A, Wp, bias, gamma, resid, C, M, N, K, EPI, CONV, cg, gdrnpp::option_split_gemm_pipe(), st, what);
    if (rc >= 0) return rc;
  }
  if (big && gdrnpp::option_split_gemm_glds()) {   // LDS-DMA kernel: any M, any of the three A forms
    if (CONV && fast3x3) hipLaunchKernelGGL((gemm_split_glds_kernel<EPI, CONV ? 1 : 0>), dim3((unsigned)tiles256), dim3(256), 0, st, A, Wp, bias, gamma, resid, C, M, N, K, cg, GnStats{nullptr, 0, 0});
    else hipLaunchKernelGGL((gemm_split_glds_kernel<EPI, CONV ? 2 : 0>), dim3((unsigned)tiles256), dim3(256), 0, st, A, Wp, bias, gamma, resid, C, M, N, K, cg, GnStats{nullptr, 0, 0});
    return gdrnpp::check_launch(what);
  }
  // (the general convolution form needs a few more registers than 256x128 tiles leave: it stays on 128x128 tiles)
  if (M % 256 == 0 && big && !(CONV && !fast3x3)) {
    if (CONV && fast3x3) hipLaunchKernelGGL((gemm_split_kernel<EPI, CONV ? 1 : 0, 4>), dim3((unsigned)tiles256), dim3(256), 0, st, A, Wp, bias, gamma, resid, C, M, N, K, cg);
    else hipLaunchKernelGGL((gemm_split_kernel<EPI, CONV ? 2 : 0, 4>), dim3((unsigned)tiles256), dim3(256), 0, st, A, Wp, bias, gamma, resid, C, M, N, K, cg);
    return gdrnpp::check_launch(what);
  }
  if (!CONV && !big && gdrnpp::option_split_gemm_pipe() && M >= 96) {   // few tiles: the 128-row form of the pipelined kernel
    const int rc = launch_split_pipe128(A, Wp, bias, gamma, resid, C, M, N, K, EPI, 0, st, what);
    if (rc >= 0) return rc;
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
  const float4* pp = reinterpret_cast<const float4*>(part) + i;
  int s = 0;
  for (; s + 4 <= splits; s += 4) {   // four partials in flight, added in split order (the sum stays in fixed order)
    const float4 v0 = pp[(size_t)s * mn4], v1 = pp[(size_t)(s + 1) * mn4], v2 = pp[(size_t)(s + 2) * mn4], v3 = pp[(size_t)(s + 3) * mn4];
    acc.x += v0.x; acc.y += v0.y; acc.z += v0.z; acc.w += v0.w;
    acc.x += v1.x; acc.y += v1.y; acc.z += v1.z; acc.w += v1.w;
    acc.x += v2.x; acc.y += v2.y; acc.z += v2.z; acc.w += v2.w;
    acc.x += v3.x; acc.y += v3.y; acc.z += v3.z; acc.w += v3.w;
  }
  for (; s < splits; ++s) {
    const float4 v = pp[(size_t)s * mn4];
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

// Plan of gdrnpp_linear_f32_splitk.  From 96 rows on the pipelined kernel (gemm_split_pipe.hip: 256x128 tiles, or its 128-row
// form below 256 such tiles) does the work: at the reference's own batch sizes (one image = a few to ~30 ROIs per forward, data_loader.py:901) the deep ConvNeXt
// stages have 16-128 such tiles for 256 CUs.  The number of K chunks minimises a two-term model measured on the stage-2 /
// stage-3 MLP shapes: a workgroup alone on its CU takes ~0.73 us per k-tile (48 MFMAs per wave at 32 cycles), rounds of 256
// workgroups run back to back, and a split costs the partials' trip through memory (written once, read once, the result
// written once) plus one kernel boundary.  splits == 1 means: no workspace, one launch with the fused epilogue.
struct SplitKPlan { bool pipe; int rows; int nkc; int splits; };   // rows: 256 / 128 = tile height of the pipelined kernel
inline SplitKPlan splitk_plan(int M, int N, int K) {
  const int nk = K / BK;
  if (M < 96 || !gdrnpp::option_split_gemm_pipe() || (unsigned long long)M * (unsigned long long)K * 4ull >= (1ull << 32)) {
    const int c = splitk_chunk(M, N, K);
    return SplitKPlan{false, 128, c, nk / c};
  }
  // fewer than 256 tiles of 256 x 128: 128-row tiles (twice the workgroups; a k-tile of 24 MFMAs per wave takes 0.65 us there, not
  // half of 0.73: five LDS-DMA pieces per wave and k-tile for half the MFMAs)
  const double tiles256 = (double)((M + 255) / 256) * (N / BN);
  const bool small = tiles256 < 256.0 && gdrnpp::option_splitk_small_tiles();
  const double tiles = small ? (double)((M + 127) / 128) * (N / BN) : tiles256;
  const double t_k = small ? 0.65 : 0.73, t_boundary = 3.0, bytes_per_us = 3.0e6;   // measured us per k-tile, one workgroup per CU
  int best = 1;
  double best_t = 1e30;
  for (int s = 1; s <= 64; s *= 2) {
    if (nk % s || (nk / s) < 4 || (nk / s) % 2) break;
    const double rounds = tiles * s <= 256.0 ? 1.0 : tiles * s / 256.0;
    double t = rounds * (nk / s) * t_k + 3.0;
    if (s > 1) t += t_boundary + (double)(s + 2) * M * (double)N * 4.0 / bytes_per_us;
    if (t < best_t) { best_t = t; best = s; }
  }
  return SplitKPlan{true, small ? 128 : 256, nk / best, best};
}

}  // namespace

extern "C" size_t gdrnpp_linear_f32_splitk_workspace_bytes(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0 || K % 32 || N % BN) return 0;
  const SplitKPlan p = splitk_plan(M, N, K);
  return (p.splits > 1 || !p.pipe) ? (size_t)p.splits * M * N * sizeof(float) : 16;   // the 128-row kernel always goes through partials
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
  const SplitKPlan plan = splitk_plan(M, N, K);
  const int nkc = plan.nkc, splits = plan.splits;
  const long tiles = (long)((M + BM - 1) / BM) * (N / BN);
  GDRNPP_REQUIRE(tiles < (1l << 31) && splits < 65536, GDRNPP_ELIMIT, "gdrnpp_linear_f32_splitk: grid too large");
  hipStream_t st = (hipStream_t)stream;
  const int a_stages = gdrnpp::option_split_gemm_pipe() == 2 ? 2 : 3;
  if (plan.pipe && splits == 1 && plan.rows == 128) {   // few tiles, K too short to cut: one launch of the 128-row form, fused epilogue
    const int rc = launch_split_pipe128(A, (const uint4*)W_packed, bias, gamma, resid, C, M, N, K, epilogue, 0, st, "gdrnpp_linear_f32_splitk");
    if (rc >= 0) return rc;
  }
  if (plan.pipe && splits == 1) {   // enough tiles for the chip: one launch, fused epilogue
    const int rc = launch_split_pipe(A, (const uint4*)W_packed, bias, gamma, resid, C, M, N, K, epilogue, false,
                                     ConvGeom{0, 0, 0, 0, 0, 0, 0, 0, 0}, a_stages, st, "gdrnpp_linear_f32_splitk");
    if (rc >= 0) return rc;
  }
  int rc = -1;
  if (plan.pipe && splits > 1 && plan.rows == 128)
    rc = launch_split_pipe128(A, (const uint4*)W_packed, nullptr, nullptr, nullptr, (float*)workspace, M, N, K, EPI_BIAS, nkc, st, "gdrnpp_linear_f32_splitk");
  else if (plan.pipe && splits > 1)
    rc = launch_split_pipe_splitk(A, (const uint4*)W_packed, (float*)workspace, M, N, K, nkc, a_stages, st, "gdrnpp_linear_f32_splitk");
  if (rc > 0) return rc;
  if (rc < 0) {
    GDRNPP_REQUIRE(workspace_bytes >= (size_t)splits * M * N * sizeof(float), GDRNPP_EINVAL, "gdrnpp_linear_f32_splitk: workspace too small");
    hipLaunchKernelGGL((gemm_split_kernel<EPI_BIAS, 0, 2>), dim3((unsigned)tiles, (unsigned)splits), dim3(256), 0, st, A,
                       (const uint4*)W_packed, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                       (float*)workspace, M, N, K, ConvGeom{0, 0, 0, 0, 0, 0, 0, 0, nkc});
  }
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

// Split-K form of the convolution for launches with too few output tiles for the chip (the head's 16x16 / 32x32 convolutions and
// Patch-PnP's at the reference's own batch sizes: K = 9 * Cin is 54-144 k-tiles long while 8 ROIs give 32-128 tiles of 128x128):
// grid (tiles, K chunks), partial sums to the workspace, fixed-order reduction with bias / GELU.  The number of chunks divides
// the k-tile count into even pieces of at least 6 k-tiles and aims at >= 256 workgroups.
namespace {
inline int conv_splitk_chunks(long M, int N, int K) {
  const long tiles = ((M + BM - 1) / BM) * (N / BN);
  const int nk = K / BK;
  int best = 1;
  for (int sp = 2; sp <= 16; ++sp) {
    if (nk % sp || (nk / sp) % 2 || nk / sp < 6) continue;
    best = sp;
    if (tiles * sp >= 256) break;
  }
  return tiles >= 192 ? 1 : best;
}
}  // namespace

extern "C" size_t gdrnpp_conv2d_f32_splitk_workspace_bytes(int n_img, int OH, int OW, int Cin, int Cout, int KH, int KW) {
  if (n_img <= 0 || OH <= 0 || OW <= 0 || Cin <= 0 || Cout <= 0 || Cout % BN || Cin % 32) return 0;
  const long M = (long)n_img * OH * OW;
  const int sp = conv_splitk_chunks(M, Cout, KH * KW * Cin);
  return sp > 1 ? (size_t)sp * M * Cout * sizeof(float) : 0;   // 0: this shape runs as one launch, call gdrnpp_conv2d_f32_split
}

extern "C" int gdrnpp_conv2d_f32_splitk(const float* x_nhwc, const void* W_packed, const float* bias, float* y_nhwc, int n_img,
                                        int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int epilogue,
                                        void* workspace, size_t workspace_bytes, void* stream) {
  GDRNPP_REQUIRE(x_nhwc && W_packed && y_nhwc, GDRNPP_EINVAL, "gdrnpp_conv2d_f32_splitk: null pointer");
  GDRNPP_REQUIRE(n_img > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && H < 32768 && W < 32768 && KH > 0 && KW > 0 &&
                     stride > 0 && pad >= 0 && pad < KH && pad < KW,
                 GDRNPP_EINVAL, "gdrnpp_conv2d_f32_splitk: bad shape");
  const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
  GDRNPP_REQUIRE(OH > 0 && OW > 0 && (OH - 1) * stride < H && (OW - 1) * stride < W, GDRNPP_EINVAL,
                 "gdrnpp_conv2d_f32_splitk: empty output or anchor pixel outside the image");
  const long M = (long)n_img * OH * OW;
  GDRNPP_REQUIRE(M < (1l << 31) && Cout % BN == 0 && Cin % 32 == 0, GDRNPP_ELIMIT,
                 "gdrnpp_conv2d_f32_splitk: Cout=%d Cin=%d must be multiples of %d/32", Cout, Cin, BN);
  GDRNPP_REQUIRE(epilogue == EPI_BIAS || epilogue == EPI_GELU, GDRNPP_EINVAL, "gdrnpp_conv2d_f32_splitk: epilogue=%d", epilogue);
  const int K = KH * KW * Cin, splits = conv_splitk_chunks(M, Cout, K);
  if (splits == 1)
    return gdrnpp_conv2d_f32_split(x_nhwc, W_packed, bias, y_nhwc, n_img, H, W, Cin, Cout, KH, KW, stride, pad, epilogue, stream);
  GDRNPP_REQUIRE(workspace && workspace_bytes >= (size_t)splits * M * Cout * sizeof(float), GDRNPP_EINVAL,
                 "gdrnpp_conv2d_f32_splitk: workspace too small");
  const long tiles = ((M + BM - 1) / BM) * (Cout / BN);
  const ConvGeom cg{H, W, Cin, OH, OW, KW, stride, pad, (K / BK) / splits};
  hipStream_t st = (hipStream_t)stream;
  const bool fast3x3 = KH == 3 && KW == 3 && stride == 1 && pad == 1;
  if (fast3x3)
    hipLaunchKernelGGL((gemm_split_kernel<EPI_BIAS, 1, 2>), dim3((unsigned)tiles, (unsigned)splits), dim3(256), 0, st, x_nhwc,
                       (const uint4*)W_packed, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                       (float*)workspace, (int)M, Cout, K, cg);
  else
    hipLaunchKernelGGL((gemm_split_kernel<EPI_BIAS, 2, 2>), dim3((unsigned)tiles, (unsigned)splits), dim3(256), 0, st, x_nhwc,
                       (const uint4*)W_packed, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                       (float*)workspace, (int)M, Cout, K, cg);
  const long mn4 = M * Cout / 4;
  const dim3 grid((unsigned)((mn4 + 255) / 256));
  if (epilogue == EPI_BIAS) hipLaunchKernelGGL(splitk_reduce_kernel<EPI_BIAS>, grid, dim3(256), 0, st, (const float*)workspace, bias, (const float*)nullptr, (const float*)nullptr, y_nhwc, mn4, Cout, splits);
  else hipLaunchKernelGGL(splitk_reduce_kernel<EPI_GELU>, grid, dim3(256), 0, st, (const float*)workspace, bias, (const float*)nullptr, (const float*)nullptr, y_nhwc, mn4, Cout, splits);
  return gdrnpp::check_launch("gdrnpp_conv2d_f32_splitk");
}

extern "C" int gdrnpp_conv3x3_f32_split(const float* x_nhwc, const void* W_packed, const float* bias, float* y_nhwc,
                                        int n_img, int H, int W, int Cin, int Cout, int epilogue, void* stream) {
  return gdrnpp_conv2d_f32_split(x_nhwc, W_packed, bias, y_nhwc, n_img, H, W, Cin, Cout, 3, 3, 1, 1, epilogue, stream);
}

extern "C" int gdrnpp_conv3x3_gnstats_partials(int H, int W) { return (H * W) % 256 == 0 ? 4 * ((H * W) / 256) : 0; }

extern "C" int gdrnpp_conv3x3_f32_split_gnstats(const float* x_nhwc, const void* W_packed, const float* bias, float* y_nhwc,
                                                double* gn_partials, int n_img, int H, int W, int Cin, int Cout, int groups,
                                                void* stream) {
  GDRNPP_REQUIRE(x_nhwc && W_packed && y_nhwc && gn_partials, GDRNPP_EINVAL, "gdrnpp_conv3x3_f32_split_gnstats: null pointer");
  GDRNPP_REQUIRE(n_img > 0 && H > 0 && W > 0 && H < 32768 && W < 32768 && Cin > 0 && Cout > 0 && groups > 0, GDRNPP_EINVAL,
                 "gdrnpp_conv3x3_f32_split_gnstats: bad shape");
  const long M = (long)n_img * H * W;
  GDRNPP_REQUIRE(M < (1l << 31) && Cout % BN == 0 && Cin % 32 == 0, GDRNPP_ELIMIT,
                 "gdrnpp_conv3x3_f32_split_gnstats: Cout=%d Cin=%d must be multiples of %d/32", Cout, Cin, BN);
  GDRNPP_REQUIRE((H * W) % 256 == 0 && Cout == 8 * groups, GDRNPP_ELIMIT,
                 "gdrnpp_conv3x3_f32_split_gnstats: needs H*W %% 256 == 0 and 8 channels per group (H*W=%d Cout=%d groups=%d)",
                 H * W, Cout, groups);
  const long tiles = (M / 256) * (Cout / BN);
  GDRNPP_REQUIRE(tiles < (1l << 30), GDRNPP_ELIMIT, "gdrnpp_conv3x3_f32_split_gnstats: grid too large");
  hipLaunchKernelGGL((gemm_split_glds_kernel<EPI_BIAS, 1, true>), dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, x_nhwc,
                     (const uint4*)W_packed, bias, (const float*)nullptr, (const float*)nullptr, y_nhwc, (int)M, Cout, 9 * Cin,
                     ConvGeom{H, W, Cin, H, W, 3, 1, 1, 0}, GnStats{gn_partials, groups, (H * W) / 256});
  return gdrnpp::check_launch("gdrnpp_conv3x3_f32_split_gnstats");
}
