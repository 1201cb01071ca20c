// Network-side HIP kernels for the GDRN_Net forward on gfx950.
//
// The dense layers of ConvNeXt-B + head stay on MIOpen / hipBLASLt (already ~140 TFLOP/s fp32).  rocprofv3
// (profiles/r01_steady_state_step_breakdown.md) showed that three memory-bound layer types were far off
// their HBM roofline in PyTorch-ROCm at B=128, fp32, channels-last:
//   * depthwise 7x7 conv (timm ConvNeXtBlock.conv_dw) ran as a CK grouped implicit-GEMM: 32 ms / step,
//     HBM floor ≈ 1 ms;  + the LayerNorm that always follows it (4.4 ms)
//   * nn.UpsamplingBilinear2d in the head (top_down_doublemask_xyz_region_head.py:80): 24.6 ms, floor 0.15 ms
//   * GroupNorm(32) + GELU of lib/torch_utils ConvModule (conv_module.py:222-236): NHWC->NCHW copy, moments,
//     apply, copy back, separate GELU: ≈ 3 ms per 64x64 layer, floor 0.3 ms
// They are re-written here for NHWC fp32 with 16-byte lanes (one float4 of channels per lane, so a wave
// touches 1 KiB contiguous per pixel), fused where the reference graph allows it.  Arithmetic is the same
// fp32 math as the PyTorch operators (exact erf GELU, biased variance, eps inside the sqrt); summation
// order differs, so parity is a tolerance (1e-5 rel, tests/test_gpu_net_kernels.py), not bit equality.
#include "common.hpp"
#include "split2_common.hpp"
#include <mutex>

namespace {

using f4 = __attribute__((ext_vector_type(4))) float;
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ f4 ld4v(const float* p) { return *reinterpret_cast<const f4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// streaming store: the big activation tensors (hundreds of MB at the headline batch) are written once and read back by the next
// layer after everything else in between.  Round 3 measured it: dwconv+LN alone 2.63 -> 2.46 ms per forward, the whole step
// unchanged (36.30 vs 36.25 ms, three same-box pairs, gpurun_out/r03k) — the consumers pay it back; off by default
#ifndef NK_NT_STORE
#define NK_NT_STORE 0
#endif
__device__ __forceinline__ void st4s(float* p, float4 v) {
#if NK_NT_STORE
  const f4 t = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(t, reinterpret_cast<f4*>(p));
#else
  *reinterpret_cast<float4*>(p) = v;
#endif
}
__device__ __forceinline__ float4 fma4(float4 a, float4 b, float4 c) {
  return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}
using gdrnpp::gelu_erf;  // common.hpp

// --------------------------------------------------------------------------------------------------
// depthwise 7x7 (pad 3, stride 1) + bias [+ LayerNorm over C], NHWC.
// One thread: one channel quad x (TH x TW) output pixels; the lanes of a workgroup cover the C/4 quads of
// NP pixel tiles, so every global access is a contiguous 16 B x (C/4) run.  Input rows stream through
// registers (10 float4 per row), weights are read tap-major [49][C] (L1/L2 resident, 49*C*4 B).
// --------------------------------------------------------------------------------------------------
// Tile per thread (TH x TW output pixels): 2 x 8 at the headline batch; 2 x 4 and 1 x 4 when a launch has too few tiles to
// give every SIMD two waves (the reference's own batches: one image = a few to ~30 ROIs) — see gdrnpp_dwconv7x7_ln_nhwc.

// LDS_W: the [49][C] weights live in LDS (loaded once per workgroup) and the workgroup is persistent, walking tile
// groups with a grid stride.  Without it every thread re-reads its 49 weight quads per output row from L1/L2 — at
// C >= 256 the weight set (49*C*4 B) exceeds the 32 KB L1, and rocprof showed the kernel bound by L2->L1 traffic
// (~16 TB/s of requests, 98 of the 178 loads per thread were weights), not by HBM (1.5 TB/s) or VALU (18 TFLOP/s).
// Sum over each aligned group of `width` lanes (16, 32 or 64), returned in every lane of the group.  DPP row shifts
// and row broadcasts run in the VALU; the ds_bpermute tree that __shfl_xor expands to went through the LDS crossbar
// six times per value and was 36 % of the dwconv+LN kernel (tools/microbench_dwconv.py, LN on/off).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  const int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false);
  return v + __builtin_bit_cast(float, t);
}
__device__ __forceinline__ float group_sum_dpp(float s, int width, int lane) {
  s = dpp_add<0x111, 0xf>(s);  // row_shr:1
  s = dpp_add<0x112, 0xf>(s);  // row_shr:2
  s = dpp_add<0x114, 0xf>(s);  // row_shr:4
  s = dpp_add<0x118, 0xf>(s);  // row_shr:8   -> lane 15 of every 16-lane row holds the row total
  if (width >= 32) s = dpp_add<0x142, 0xa>(s);  // row_bcast:15 into rows 1 and 3
  if (width == 64) {
    s = dpp_add<0x143, 0xc>(s);                 // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave total
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s), 63));
  }
  return __shfl(s, lane | (width - 1), 64);
}

#ifndef DW_WPE
#define DW_WPE 2
#endif
#ifndef DW_PK
#define DW_PK 1
#endif
#ifndef DW_NT_STORE
#define DW_NT_STORE 0
#endif
__device__ __forceinline__ f4 fma4s(f4 a, f4 b, f4 c) {
  return f4{fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w)};
}
#if DW_PK
#define DW_FMA(a, b, c) __builtin_elementwise_fma(a, b, c)
#else
#define DW_FMA(a, b, c) fma4s(a, b, c)
#endif
template <bool FUSE_LN, bool LDS_W, int TH, int TW>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(DW_WPE))) void dwconv7_ln_kernel(const float* __restrict__ x, const float* __restrict__ w49c,
                                                         const float* __restrict__ bias,
                                                         const float* __restrict__ ln_w,
                                                         const float* __restrict__ ln_b, float* __restrict__ y, int N,
                                                         int H, int W, int C, float eps, int buf_rows, int y_rows) {
  __shared__ float red[TH * TW * 8];  // FUSE_LN: per-wave partials when a pixel spans several waves
  extern __shared__ float4 wlds[];    // LDS_W: [49][C/4]
  const f4* wldsv = reinterpret_cast<const f4*>(wlds);
  const int Q = C >> 2;                       // channel quads per pixel
  const int tiles_per_block = blockDim.x / Q; // >= 1 (host guarantees Q <= 256 and blockDim % Q == 0)
  const int q = threadIdx.x % Q;
  const int tl = threadIdx.x / Q;
  const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
  const long n_tiles = (long)N * tiles_y * tiles_x;
  const long n_groups = (n_tiles + tiles_per_block - 1) / tiles_per_block;
  if (LDS_W) {
    // all loads of the weight image in flight before the first LDS write (a load -> ds_write loop paid the memory latency per
    // trip: ~12 us per workgroup, the whole launch at the reference's batch sizes)
    constexpr int WMAX = 13;                  // 49 * C / 4 slots over >= 256 threads: <= 12.25 per thread for C <= 512
    const int n_slots = 49 * Q;
    if (n_slots <= WMAX * (int)blockDim.x) {
      float4 wtmp[WMAX];
#pragma unroll
      for (int k = 0; k < WMAX; ++k) {
        const int i = threadIdx.x + k * blockDim.x;
        if (i < n_slots) wtmp[k] = ld4(w49c + 4 * (size_t)i);
      }
#pragma unroll
      for (int k = 0; k < WMAX; ++k) {
        const int i = threadIdx.x + k * blockDim.x;
        if (i < n_slots) wlds[i] = wtmp[k];
      }
    } else {
      for (int i = threadIdx.x; i < n_slots; i += blockDim.x) wlds[i] = ld4(w49c + 4 * (size_t)i);
    }
    __syncthreads();
  }
  const f4 b4 = ld4v(bias + 4 * q);
  // XCD-aware walk: consecutive workgroup ids are dealt round-robin to the 8 XCDs, each with its own L2.  A tile needs
  // a (TH+6)x(TW+6) input halo, 7x its own size, shared with its neighbours — so neighbouring tiles must run on the
  // SAME XCD or every L2 fetches its own copy of the halo from HBM/MALL.  XCD k gets the contiguous group range
  // [k*n/8, (k+1)*n/8) and its workgroups stride through it.
  const bool by_xcd = gridDim.x >= 8;  // every XCD owns at least one workgroup
  const int xcd = blockIdx.x & 7;
  const long g_lo = by_xcd ? n_groups * xcd / 8 : 0, g_hi = by_xcd ? n_groups * (xcd + 1) / 8 : n_groups;
  const int wg_stride = by_xcd ? (int)((gridDim.x - xcd + 7) >> 3) : (int)gridDim.x;
  for (long group = g_lo + (by_xcd ? blockIdx.x >> 3 : blockIdx.x); group < g_hi; group += wg_stride) {
  const long tile = group * tiles_per_block + tl;
  const bool active = tile < n_tiles;
  int n = 0, ty0 = 0, tx0 = 0;
  if (active) {
    n = (int)(tile / (tiles_y * tiles_x));
    const int r = (int)(tile - (long)n * tiles_y * tiles_x);
    ty0 = (r / tiles_x) * TH;
    tx0 = (r % tiles_x) * TW;
  }
  // accumulators as 4-wide vectors: __builtin_elementwise_fma lowers to two v_pk_fma_f32 (the plain v_fma_f32 rate is
  // half the fp32 vector peak on CDNA4; same IEEE fma per channel)
  f4 acc[TH][TW];
#pragma unroll
  for (int i = 0; i < TH; ++i)
#pragma unroll
    for (int j = 0; j < TW; ++j) acc[i][j] = b4;

  if (active && buf_rows) {
    // Input rows through buffer loads (rocprofv3 SQ counters: the kernel is VALU-bound and only 37 % of its VALU instructions
    // were the pk_fma's — the rest was 64-bit address arithmetic per load and the compare + 4 v_cndmask that zero the columns
    // outside the image).  One buffer descriptor per image row (base uniform in the wave, num_records = the row's bytes): a
    // column offset outside [0, W*C*4) — negative ones wrap to huge unsigned values — makes the hardware return 0, and the
    // TW+6 column offsets are constants of the tile, so a row costs its loads and nothing else.
    using u4 = __attribute__((ext_vector_type(4))) unsigned;
    const int n_u = __builtin_amdgcn_readfirstlane(n), ty_u = __builtin_amdgcn_readfirstlane(ty0);
    int coff[TW + 6];
#pragma unroll
    for (int c = 0; c < TW + 6; ++c) coff[c] = ((tx0 + c - 3) * C + 4 * q) * 4;
    const int row_bytes = W * C * 4;
#pragma unroll 1
    for (int r = 0; r < TH + 6; ++r) {
      const int iy = ty_u + r - 3;
      if (iy < 0 || iy >= H) continue;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(x) + ((size_t)n_u * H + iy) * W * C, 0, row_bytes, 0x00020000);
      f4 row[TW + 6];
#ifdef DW_TIMING_NO_LOAD   // timing-only builds (results invalid): the kernel without its input loads / without its weight reads
#pragma unroll
      for (int c = 0; c < TW + 6; ++c) row[c] = b4 * (float)(c + r);
      (void)rs;
#else
#pragma unroll
      for (int c = 0; c < TW + 6; ++c) row[c] = __builtin_bit_cast(f4, (u4)__builtin_amdgcn_raw_buffer_load_b128(rs, coff[c], 0, 0));
#endif
#pragma unroll
      for (int i = 0; i < TH; ++i) {
        const int ky = r - i;
        if (ky < 0 || ky > 6) continue;
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) {
#ifdef DW_TIMING_NO_WLDS
          const f4 wv = b4 * (float)(ky * 7 + kx);
#else
          const f4 wv = LDS_W ? wldsv[(ky * 7 + kx) * Q + q] : ld4v(w49c + (size_t)(ky * 7 + kx) * C + 4 * q);
#endif
#pragma unroll
          for (int j = 0; j < TW; ++j) acc[i][j] = DW_FMA(row[j + kx], wv, acc[i][j]);
        }
      }
    }
  } else if (active) {
    const float* xn = x + (size_t)n * H * W * C + 4 * q;
#pragma unroll 1
    for (int r = 0; r < TH + 6; ++r) {
      const int iy = ty0 + r - 3;
      if (iy < 0 || iy >= H) continue;
      f4 row[TW + 6];
#pragma unroll
      for (int c = 0; c < TW + 6; ++c) {
        const int ix = tx0 + c - 3;
        row[c] = (ix >= 0 && ix < W) ? ld4v(xn + ((size_t)iy * W + ix) * C) : f4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int i = 0; i < TH; ++i) {
        const int ky = r - i;
        if (ky < 0 || ky > 6) continue;
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) {
          const f4 wv = LDS_W ? wldsv[(ky * 7 + kx) * Q + q] : ld4v(w49c + (size_t)(ky * 7 + kx) * C + 4 * q);
#pragma unroll
          for (int j = 0; j < TW; ++j) acc[i][j] = DW_FMA(row[j + kx], wv, acc[i][j]);
        }
      }
    }
  }

  if (FUSE_LN) {
    // LayerNorm over the C channels of each pixel.  The Q = C/4 lanes holding one pixel are Q consecutive lanes:
    // xor-shuffle tree inside the wave (width min(Q,64)), then <= 4 per-wave partials through LDS when Q > 64.
    // Two rounds — mean, then centred variance (biased), as ATen's layer norm defines them.
    constexpr int P = TH * TW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int width = Q < 64 ? Q : 64;
    const int wpt = Q > 64 ? Q / 64 : 1;  // waves per pixel tile
    const int w0 = (wave / wpt) * wpt;
    float mean[P], var[P];
#pragma unroll
    for (int round = 0; round < 2; ++round) {
      float part[P];
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const f4 v = acc[p / TW][p % TW];
        float s;
        if (round == 0) {
          s = (v.x + v.y) + (v.z + v.w);
        } else {
          const float dx = v.x - mean[p], dy = v.y - mean[p], dz = v.z - mean[p], dw = v.w - mean[p];
          s = (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
        if (width >= 16) {
          s = group_sum_dpp(s, width, lane);
        } else {
          for (int off = width >> 1; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
        }
        part[p] = s;
      }
      if (wpt > 1) {
        __syncthreads();  // red[] of the previous round / previous tile group has been consumed by every wave
        if (lane == 0) {
#pragma unroll
          for (int p = 0; p < P; ++p) red[p * 8 + wave] = part[p];
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < P; ++p) {
          float s = 0.f;
          for (int k = 0; k < wpt; ++k) s += red[p * 8 + w0 + k];
          part[p] = s;
        }
      }
#pragma unroll
      for (int p = 0; p < P; ++p) {
        if (round == 0) mean[p] = part[p] / (float)C;
        else var[p] = part[p] / (float)C;
      }
    }
    const float4 g4 = ld4(ln_w + 4 * q), be4 = ld4(ln_b + 4 * q);
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const float rstd = rsqrtf(var[p] + eps);
      const float m = mean[p];
      f4 v = acc[p / TW][p % TW];
      v.x = (v.x - m) * rstd * g4.x + be4.x;
      v.y = (v.y - m) * rstd * g4.y + be4.y;
      v.z = (v.z - m) * rstd * g4.z + be4.z;
      v.w = (v.w - m) * rstd * g4.w + be4.w;
      acc[p / TW][p % TW] = v;
    }
  }
  if (active) {
    float* yn = y + (size_t)n * H * W * C + 4 * q;
#pragma unroll
    for (int i = 0; i < TH; ++i)
#pragma unroll
      for (int j = 0; j < TW; ++j) {
        const int oy = ty0 + i, ox = tx0 + j;
        f4 o = acc[i][j];
        if (y_rows) {   // "f16x2 rows" output (split2_common.hpp): threads q = 2k / 2k + 1 of a pixel are neighbouring lanes
          const uint4 u = gdrnpp::split2::f16x2_rows_quad(o.x, o.y, o.z, o.w, q & 1);
          o = f4{__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w)};
        }
#if DW_NT_STORE
        if (oy < H && ox < W) __builtin_nontemporal_store(o, reinterpret_cast<f4*>(yn + ((size_t)oy * W + ox) * C));
#else
        if (oy < H && ox < W) *reinterpret_cast<f4*>(yn + ((size_t)oy * W + ox) * C) = o;
#endif
      }
  }
  }  // tile-group loop
}

// --------------------------------------------------------------------------------------------------
// bilinear x2 upsample, align_corners=True (nn.UpsamplingBilinear2d), NHWC.  Index/lambda arithmetic follows
// ATen's area_pixel_compute_source_index (scale = (in-1)/(out-1) in fp32, src = scale*dst).
// --------------------------------------------------------------------------------------------------
// One thread = one channel quad of a 2 x 2 OUTPUT block (the four outputs above input pixel (i, j)): with scale (H-1)/(2H-1) the
// source rows of output rows 2i / 2i+1 are (ya, ya+1) and (ya+1, ya+2) — the same for columns — except in the first block row /
// column, so nine 16-byte loads serve four outputs instead of sixteen.  The kernel was bound by those loads (L2 -> L1 traffic 4x
// the output bytes), not by HBM.  Each output is the expression of the one-output form below, bit for bit (generic path = that
// form, taken where the row / column pattern differs).
__device__ __forceinline__ float4 bilerp4(float ly0, float ly1, float lx0, float lx1, float4 v00, float4 v01, float4 v10, float4 v11) {
  float4 o;
  o.x = ly0 * (lx0 * v00.x + lx1 * v01.x) + ly1 * (lx0 * v10.x + lx1 * v11.x);
  o.y = ly0 * (lx0 * v00.y + lx1 * v01.y) + ly1 * (lx0 * v10.y + lx1 * v11.y);
  o.z = ly0 * (lx0 * v00.z + lx1 * v01.z) + ly1 * (lx0 * v10.z + lx1 * v11.z);
  o.w = ly0 * (lx0 * v00.w + lx1 * v01.w) + ly1 * (lx0 * v10.w + lx1 * v11.w);
  return o;
}
__global__ __launch_bounds__(256) void upsample2x_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C) {
  const int Q = C >> 2;
  const int OH = 2 * H, OW = 2 * W;
  const long total = (long)N * H * W * Q;
  const float sh = (OH > 1) ? (float)(H - 1) / (float)(OH - 1) : 0.f;
  const float sw = (OW > 1) ? (float)(W - 1) / (float)(OW - 1) : 0.f;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int q = (int)(t % Q);
    long p = t / Q;
    const int j = (int)(p % W); p /= W;
    const int i = (int)(p % H);
    const int n = (int)(p / H);
    float fy[2], fx[2];
    int y0[2], x0[2];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      fy[d] = sh * (float)(2 * i + d); y0[d] = (int)fy[d];
      fx[d] = sw * (float)(2 * j + d); x0[d] = (int)fx[d];
    }
    const float* b = x + ((size_t)n * H * W) * C + 4 * q;
    float* yo = y + (((size_t)n * OH + 2 * i) * OW + 2 * j) * C + 4 * q;
    if (y0[1] == y0[0] + 1 && x0[1] == x0[0] + 1) {
      const int ya = y0[0], xa = x0[0];
      const int r2 = min(ya + 2, H - 1), c2 = min(xa + 2, W - 1);
      const int rr[3] = {ya, ya + 1, r2}, cc[3] = {xa, xa + 1, c2};
      float4 v[3][3];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) v[r][c] = ld4(b + ((size_t)rr[r] * W + cc[c]) * C);
#pragma unroll
      for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          const float ly1 = fy[dy] - (float)y0[dy], ly0 = 1.f - ly1, lx1 = fx[dx] - (float)x0[dx], lx0 = 1.f - lx1;
          st4s(yo + ((size_t)dy * OW + dx) * C, bilerp4(ly0, ly1, lx0, lx1, v[dy][dx], v[dy][dx + 1], v[dy + 1][dx], v[dy + 1][dx + 1]));
        }
    } else {
#pragma unroll
      for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          const int yy = y0[dy], xx = x0[dx];
          const int yp = (yy < H - 1) ? 1 : 0, xp = (xx < W - 1) ? 1 : 0;
          const float ly1 = fy[dy] - (float)yy, ly0 = 1.f - ly1, lx1 = fx[dx] - (float)xx, lx0 = 1.f - lx1;
          const float4 v00 = ld4(b + ((size_t)yy * W + xx) * C), v01 = ld4(b + ((size_t)yy * W + xx + xp) * C);
          const float4 v10 = ld4(b + ((size_t)(yy + yp) * W + xx) * C), v11 = ld4(b + ((size_t)(yy + yp) * W + xx + xp) * C);
          st4s(yo + ((size_t)dy * OW + dx) * C, bilerp4(ly0, ly1, lx0, lx1, v00, v01, v10, v11));
        }
    }
  }
}

// --------------------------------------------------------------------------------------------------
// y = act(x + bias[c] (+ resid)), NHWC float4 lanes: what is left of [BatchNorm2d, (residual add,) ReLU] once the
// inference-mode BatchNorm is folded into the preceding convolution's weights (ResNet BasicBlock, timm resnet.py).
// In place when y == x (the only way the ResNet path calls it): x and y therefore carry no __restrict__.
// --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bias_act_kernel(const float* x, const float* __restrict__ bias,
                                                       const float* resid, float* y, long total4,
                                                       int Q, int relu) {
  // four elements per trip, loads first (one 16-byte load in flight per thread caps a streaming pass at ~4 TB/s)
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += 4 * stride) {
    float4 v[4], r[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i + u * stride < total4) {
        v[u] = ld4(x + 4 * (i + u * stride));
        if (resid) r[u] = ld4(resid + 4 * (i + u * stride));
      }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long k = i + u * stride;
      if (k >= total4) break;
      const float4 b = ld4(bias + 4 * (int)(k % Q));
      float4 w = v[u];
      w.x += b.x; w.y += b.y; w.z += b.z; w.w += b.w;
      if (resid) { w.x += r[u].x; w.y += r[u].y; w.z += r[u].z; w.w += r[u].w; }
      if (relu) { w.x = fmaxf(w.x, 0.f); w.y = fmaxf(w.y, 0.f); w.z = fmaxf(w.z, 0.f); w.w = fmaxf(w.w, 0.f); }
      st4(y + 4 * k, w);
    }
  }
}

// --------------------------------------------------------------------------------------------------
// Transposed convolution (nn.ConvTranspose2d, square kernel / stride / padding) as a GEMM plus this gather: the GEMM
// cols[n, iy, ix, (ky, kx, co)] = sum_ci x[n, iy, ix, ci] * w[ci, co, ky, kx] does exactly the useful multiplies (a
// zero-stuffed convolution would do stride^2 times as many); output pixel (oy, ox) then sums the taps whose input
// position (oy + pad - ky) / stride, (ox + pad - kx) / stride is integral and inside the image — at most
// ceil(KS / stride)^2 of them, added in (ky, kx) order, so the result is deterministic.
// --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void deconv_col2im_kernel(const float* __restrict__ cols, const float* __restrict__ bias,
                                                            float* __restrict__ y, int H, int W, int C, int KS, int stride,
                                                            int pad, int OH, int OW, long total) {
  const int Q = C >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int q = (int)(i % Q);
    long p = i / Q;
    const int ox = (int)(p % OW);
    p /= OW;
    const int oy = (int)(p % OH);
    const long n = p / OH;
    float4 a = bias ? ld4(bias + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int ky = 0; ky < KS; ++ky) {
      const int ty = oy + pad - ky;
      if (ty < 0 || ty % stride) continue;
      const int iy = ty / stride;
      if (iy >= H) continue;
      for (int kx = 0; kx < KS; ++kx) {
        const int tx = ox + pad - kx;
        if (tx < 0 || tx % stride) continue;
        const int ix = tx / stride;
        if (ix >= W) continue;
        const float4 v = ld4(cols + (((size_t)n * H + iy) * W + ix) * ((size_t)KS * KS * C) + (size_t)(ky * KS + kx) * C + 4 * q);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
      }
    }
    st4(y + (((size_t)n * OH + oy) * OW + ox) * C + 4 * q, a);
  }
}

// The same gather with the GroupNorm statistics of its result taken on the way (the head's ConvTranspose2d is followed by
// GroupNorm + GELU, top_down_doublemask_xyz_region_head.py:53-75): launch geometry, pixel partition and summation order of
// gn_stats_kernel, so the partials [N, P, G, 2] — and with them the normalised tensor — are bitwise those of the two-pass path.
__global__ __launch_bounds__(256) void deconv_col2im_gn_kernel(const float* __restrict__ cols, const float* __restrict__ bias,
                                                               float* __restrict__ y, double* __restrict__ part, int H, int W,
                                                               int C, int KS, int stride, int pad, int OH, int OW, int G, int P) {
  extern __shared__ double sred[];  // [2][256]
  const int Q = C >> 2, cpg = C / G, HW = OH * OW;
  const int n = blockIdx.y, pc = blockIdx.x;
  const int q = threadIdx.x % Q, row = threadIdx.x / Q, rows = blockDim.x / Q;
  const int per = (HW + P - 1) / P;
  const int p0 = pc * per, p1 = min(HW, p0 + per);
  const float4 b4 = bias ? ld4(bias + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
  double s = 0.0, ss = 0.0;
  for (int p = p0 + row; p < p1; p += rows) {
    const int oy = p / OW, ox = p - oy * OW;
    float4 a = b4;
    for (int ky = 0; ky < KS; ++ky) {
      const int ty = oy + pad - ky;
      if (ty < 0 || ty % stride) continue;
      const int iy = ty / stride;
      if (iy >= H) continue;
      for (int kx = 0; kx < KS; ++kx) {
        const int tx = ox + pad - kx;
        if (tx < 0 || tx % stride) continue;
        const int ix = tx / stride;
        if (ix >= W) continue;
        const float4 v = ld4(cols + (((size_t)n * H + iy) * W + ix) * ((size_t)KS * KS * C) + (size_t)(ky * KS + kx) * C + 4 * q);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
      }
    }
    st4(y + ((size_t)n * HW + p) * C + 4 * q, a);
    s += ((double)a.x + (double)a.y) + ((double)a.z + (double)a.w);
    ss += ((double)a.x * a.x + (double)a.y * a.y) + ((double)a.z * a.z + (double)a.w * a.w);
  }
  sred[threadIdx.x] = s;
  sred[256 + threadIdx.x] = ss;
  __syncthreads();
  if ((int)threadIdx.x < G) {
    const int g = threadIdx.x, qpg = cpg >> 2;
    double a = 0.0, c2 = 0.0;
    for (int r = 0; r < rows; ++r)
      for (int k = 0; k < qpg; ++k) {
        const int t = r * Q + g * qpg + k;
        a += sred[t];
        c2 += sred[256 + t];
      }
    double* o = part + (((size_t)n * P + pc) * G + g) * 2;
    o[0] = a;
    o[1] = c2;
  }
}

// --------------------------------------------------------------------------------------------------
// GroupNorm (+ optional exact GELU), NHWC.  Pass A: per (sample, pixel chunk) fp64 partial sums per group,
// written to a workspace [N, P, G, 2] (no atomics -> deterministic).  Pass B: every thread rebuilds mean/rstd of
// its quad's group from the P partials and applies (x-mean)*rstd*gamma+beta, then GELU.
// --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x, double* __restrict__ part, int HW,
                                                       int C, int G, int P) {
  extern __shared__ double sred[];  // [2][256]
  const int Q = C >> 2, cpg = C / G;
  const int n = blockIdx.y, pc = blockIdx.x;
  const int q = threadIdx.x % Q, row = threadIdx.x / Q, rows = blockDim.x / Q;
  const int per = (HW + P - 1) / P;
  const int p0 = pc * per, p1 = min(HW, p0 + per);
  double s = 0.0, ss = 0.0;
  const float* b = x + (size_t)n * HW * C + 4 * q;
  for (int p = p0 + row; p < p1; p += rows) {
    const float4 v = ld4(b + (size_t)p * C);
    s += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
    ss += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
  }
  sred[threadIdx.x] = s;
  sred[256 + threadIdx.x] = ss;
  __syncthreads();
  // one thread per group sums the (rows x quads-per-group) partials in a fixed order
  if ((int)threadIdx.x < G) {
    const int g = threadIdx.x, qpg = cpg >> 2;
    double a = 0.0, c2 = 0.0;
    for (int r = 0; r < rows; ++r)
      for (int k = 0; k < qpg; ++k) {
        const int t = r * Q + g * qpg + k;
        a += sred[t];
        c2 += sred[256 + t];
      }
    double* o = part + (((size_t)n * P + pc) * G + g) * 2;
    o[0] = a;
    o[1] = c2;
  }
}

template <bool GELU>
__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, const double* __restrict__ part,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float* __restrict__ y, int HW, int C, int G, int P, float eps) {
  __shared__ float s_mean[64], s_rstd[64];
  const int Q = C >> 2, cpg = C / G;
  const int n = blockIdx.y;
  // Group statistics from the P x G fp64 partials.  Every workgroup needs them before it can stream: 8 lanes per group share the P
  // partials (8 independent loads in flight per lane instead of a chain of 64 per group thread: the serial form held each of the
  // 8192 workgroups of a 64x64 launch ~6 us before its first store — a quarter of its lifetime), then a 3-step shuffle tree.
  if (G * 8 <= (int)blockDim.x) {
    const int g8 = threadIdx.x >> 3, sub = threadIdx.x & 7;
    double a = 0.0, c2 = 0.0;
    if (g8 < G) {
      for (int p = sub; p < P; p += 8) {
        const double* o = part + (((size_t)n * P + p) * G + g8) * 2;
        a += o[0];
        c2 += o[1];
      }
    }
#pragma unroll
    for (int off = 4; off >= 1; off >>= 1) {
      a += __shfl_xor(a, off, 64);
      c2 += __shfl_xor(c2, off, 64);
    }
    if (g8 < G && sub == 0) {
      const double cnt = (double)HW * cpg;
      const double m = a / cnt;
      double var = c2 / cnt - m * m;
      if (var < 0.0) var = 0.0;
      s_mean[g8] = (float)m;
      s_rstd[g8] = (float)(1.0 / sqrt(var + (double)eps));
    }
  } else if ((int)threadIdx.x < G) {
    double a = 0.0, c2 = 0.0;
    for (int p = 0; p < P; ++p) {
      const double* o = part + (((size_t)n * P + p) * G + threadIdx.x) * 2;
      a += o[0];
      c2 += o[1];
    }
    const double cnt = (double)HW * cpg;
    const double m = a / cnt;
    double var = c2 / cnt - m * m;
    if (var < 0.0) var = 0.0;
    s_mean[threadIdx.x] = (float)m;
    s_rstd[threadIdx.x] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  const long total = (long)HW * Q;
  // Four elements per trip with their loads issued together: one 16-byte load in flight per thread held the pass at ~4 TB/s
  // (2048 threads per CU x 16 B per ~2 us of latency).  The grid stride is a multiple of Q (host), so a thread keeps its
  // channel quad — gamma / beta / mean / rstd are loop constants.
  const long stride = (long)gridDim.x * blockDim.x;
  const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int q = (int)(i0 % Q);
  const int g = (4 * q) / cpg;
  const float m = s_mean[g], r = s_rstd[g];
  const float4 ga = ld4(gamma + 4 * q), be = ld4(beta + 4 * q);
  const float* xn = x + (size_t)n * HW * C;
  float* yn = y + (size_t)n * HW * C;
  for (long i = i0; i < total; i += 4 * stride) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i + u * stride < total) v[u] = ld4(xn + 4 * (i + u * stride));
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (i + u * stride >= total) break;
      float4 w = v[u];
      w.x = (w.x - m) * r * ga.x + be.x;
      w.y = (w.y - m) * r * ga.y + be.y;
      w.z = (w.z - m) * r * ga.z + be.z;
      w.w = (w.w - m) * r * ga.w + be.w;
      if (GELU) { w.x = gelu_erf(w.x); w.y = gelu_erf(w.y); w.z = gelu_erf(w.z); w.w = gelu_erf(w.w); }
      st4s(yn + 4 * (i + u * stride), w);
    }
  }
}

// GroupNorm (+ GELU) of SMALL images in one launch (Patch-PnP's 32x32 / 16x16 / 8x8 x 128-channel maps: the two-pass form spent a
// launch on ~8 us of statistics): one 1024-thread workgroup per image, pass 1 = fp64 sums per thread and an LDS gather per group,
// pass 2 = normalise the same elements again (the image, <= 1 MB, comes back from L2).
template <bool GELU>
__global__ __launch_bounds__(1024) void gn_small_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ y, int HW, int C, int G,
                                                        float eps) {
  extern __shared__ double sred[];  // [2][1024]
  __shared__ float s_mean[64], s_rstd[64];
  const int Q = C >> 2, cpg = C / G;
  const int n = blockIdx.x;
  const int q = threadIdx.x % Q, row = threadIdx.x / Q, rows = blockDim.x / Q;
  const float* xb = x + (size_t)n * HW * C + 4 * q;
  float* yb = y + (size_t)n * HW * C + 4 * q;
  double s = 0.0, ss = 0.0;
  for (int p = row; p < HW; p += rows) {
    const float4 v = ld4(xb + (size_t)p * C);
    s += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
    ss += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
  }
  sred[threadIdx.x] = s;
  sred[1024 + threadIdx.x] = ss;
  __syncthreads();
  if ((int)threadIdx.x < G) {
    const int g = threadIdx.x, qpg = cpg >> 2;
    double a = 0.0, c2 = 0.0;
    for (int r = 0; r < rows; ++r)
      for (int k = 0; k < qpg; ++k) {
        const int t = r * Q + g * qpg + k;
        a += sred[t];
        c2 += sred[1024 + t];
      }
    const double cnt = (double)HW * cpg, m = a / cnt;
    double var = c2 / cnt - m * m;
    if (var < 0.0) var = 0.0;
    s_mean[g] = (float)m;
    s_rstd[g] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  const int g = (4 * q) / cpg;
  const float m = s_mean[g], r = s_rstd[g];
  const float4 ga = ld4(gamma + 4 * q), be = ld4(beta + 4 * q);
  for (int p = row; p < HW; p += rows) {
    float4 w = ld4(xb + (size_t)p * C);
    w.x = (w.x - m) * r * ga.x + be.x;
    w.y = (w.y - m) * r * ga.y + be.y;
    w.z = (w.z - m) * r * ga.z + be.z;
    w.w = (w.w - m) * r * ga.w + be.w;
    if (GELU) { w.x = gelu_erf(w.x); w.y = gelu_erf(w.y); w.z = gelu_erf(w.z); w.w = gelu_erf(w.w); }
    st4(yb + (size_t)p * C, w);
  }
}


// --------------------------------------------------------------------------------------------------
// LayerNorm over C of an NHWC tensor (timm LayerNorm2d of the ConvNeXt stem / downsample layers).  One thread holds one
// channel quad of LN_PIX pixels (all loads issued before the first reduction); the C/4 lanes of a pixel are
// consecutive, sums go through DPP inside the wave and, when a pixel spans several waves, through LDS.
// Two passes over registers (mean, then centred variance), biased variance, eps inside the sqrt.
// --------------------------------------------------------------------------------------------------
constexpr int LN_PIX = 4;
__global__ __launch_bounds__(256) void layernorm_nhwc_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ b, float* __restrict__ y,
                                                             long n_pix, int C, float eps) {
  __shared__ float red[LN_PIX * 4];
  const int Q = C >> 2, ppb = blockDim.x / Q, q = threadIdx.x % Q, pl = threadIdx.x / Q;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int width = Q < 64 ? Q : 64, wpt = Q > 64 ? Q / 64 : 1, w0 = (wave / wpt) * wpt;
  const f4 g4 = ld4v(w + 4 * q), b4 = ld4v(b + 4 * q);
  const long n_groups = (n_pix + (long)ppb * LN_PIX - 1) / ((long)ppb * LN_PIX);
  const float inv_c = 1.f / (float)C;
  for (long group = blockIdx.x; group < n_groups; group += gridDim.x) {
    f4 v[LN_PIX];
    long pix[LN_PIX];
#pragma unroll
    for (int u = 0; u < LN_PIX; ++u) {
      pix[u] = (group * LN_PIX + u) * ppb + pl;
      v[u] = pix[u] < n_pix ? ld4v(x + pix[u] * C + 4 * q) : f4{0.f, 0.f, 0.f, 0.f};
    }
    float mean[LN_PIX], rstd[LN_PIX];
#pragma unroll
    for (int round = 0; round < 2; ++round) {
      float part[LN_PIX];
#pragma unroll
      for (int u = 0; u < LN_PIX; ++u) {
        float s;
        if (round == 0) {
          s = (v[u].x + v[u].y) + (v[u].z + v[u].w);
        } else {
          v[u] -= mean[u];
          s = (v[u].x * v[u].x + v[u].y * v[u].y) + (v[u].z * v[u].z + v[u].w * v[u].w);
        }
        if (width >= 16) {
          s = group_sum_dpp(s, width, lane);
        } else {
          for (int off = width >> 1; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
        }
        part[u] = s;
      }
      if (wpt > 1) {
        __syncthreads();
        if (lane == 0) {
#pragma unroll
          for (int u = 0; u < LN_PIX; ++u) red[u * 4 + wave] = part[u];
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < LN_PIX; ++u) {
          float s = 0.f;
          for (int k = 0; k < wpt; ++k) s += red[u * 4 + w0 + k];
          part[u] = s;
        }
      }
#pragma unroll
      for (int u = 0; u < LN_PIX; ++u) {
        if (round == 0) mean[u] = part[u] * inv_c;
        else rstd[u] = rsqrtf(part[u] * inv_c + eps);
      }
    }
#pragma unroll
    for (int u = 0; u < LN_PIX; ++u)
      if (pix[u] < n_pix) *reinterpret_cast<f4*>(y + pix[u] * C + 4 * q) = v[u] * rstd[u] * g4 + b4;
  }
}

// --------------------------------------------------------------------------------------------------
// ConvNeXt stem: Conv2d(3 -> 128, 4x4, stride 4) + bias + LayerNorm2d in one pass, NCHW image in, NHWC features out
// (timm convnext stem_0 / stem_1).  MIOpen ran the convolution (0.11-0.23 ms at 128 ROIs), ATen added the bias (0.08 ms),
// a layout copy and a LayerNorm launch followed.  One wave = the 64 channel pairs of a pixel: the 2 x 48 weights of a lane
// live in registers for the whole kernel, the 4 x 256 x 3 input floats of an output row are staged in LDS and read back as
// broadcast float4s, LayerNorm over the 128 channels is a wave reduction (DPP).  fp32 fma chain in (ci, ky, kx) order.
// (Round 5: the lane's two channels as one v_pk_fma_f32 accumulator — 48 instead of 96 instructions per pixel — ran 194 instead of
// 164 us in the step, and with two pixels per trip (two independent packed chains) 211 instead of 191 us isolated: the packed form
// with a broadcast operand issues slower than two scalar chains (profiles/r04_dwconv_valu.txt, pk_bcast); not kept.)
// --------------------------------------------------------------------------------------------------
constexpr int kStemC = 128, kStemK = 48;
__global__ __launch_bounds__(256) void stem_conv_ln_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, const float* __restrict__ ln_w,
                                                           const float* __restrict__ ln_b, float* __restrict__ y, int N, int H,
                                                           int W, float eps) {
  extern __shared__ float4 stem_rows[];   // [3][4][W/4] float4: the four image rows of one output row, per input channel
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int OH = H >> 2, OW = W >> 2, W4 = W >> 2;
  float wr[2][kStemK];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int k4 = 0; k4 < kStemK / 4; ++k4) {
      const float4 v = ld4(w + (size_t)(2 * lane + c) * kStemK + 4 * k4);
      wr[c][4 * k4] = v.x; wr[c][4 * k4 + 1] = v.y; wr[c][4 * k4 + 2] = v.z; wr[c][4 * k4 + 3] = v.w;
    }
  const float b0 = bias ? bias[2 * lane] : 0.f, b1 = bias ? bias[2 * lane + 1] : 0.f;
  const float g0 = ln_w[2 * lane], g1 = ln_w[2 * lane + 1], be0 = ln_b[2 * lane], be1 = ln_b[2 * lane + 1];
  const long n_rows = (long)N * OH;
  for (long row = blockIdx.x; row < n_rows; row += gridDim.x) {
    const int n = (int)(row / OH), oy = (int)(row - (long)n * OH);
    __syncthreads();   // the previous row's reads are done
    for (int i = tid; i < 12 * W4; i += 256) {
      const int r = i / W4, c4 = i - r * W4, ci = r >> 2, ky = r & 3;
      stem_rows[i] = ld4(x + (((size_t)n * 3 + ci) * H + 4 * oy + ky) * W + 4 * c4);
    }
    __syncthreads();
    for (int ox = wave; ox < OW; ox += 4) {
      float a0 = b0, a1 = b1;
#pragma unroll
      for (int r = 0; r < 12; ++r) {
        const float4 v = stem_rows[r * W4 + ox];   // same address in every lane: LDS broadcast
        a0 = fmaf(v.x, wr[0][4 * r], a0); a1 = fmaf(v.x, wr[1][4 * r], a1);
        a0 = fmaf(v.y, wr[0][4 * r + 1], a0); a1 = fmaf(v.y, wr[1][4 * r + 1], a1);
        a0 = fmaf(v.z, wr[0][4 * r + 2], a0); a1 = fmaf(v.z, wr[1][4 * r + 2], a1);
        a0 = fmaf(v.w, wr[0][4 * r + 3], a0); a1 = fmaf(v.w, wr[1][4 * r + 3], a1);
      }
      // LayerNorm2d over the 128 channels of the pixel (two passes, biased variance, eps inside the sqrt: as layernorm_nhwc_kernel)
      const float mean = group_sum_dpp(a0 + a1, 64, lane) * (1.f / (float)kStemC);
      a0 -= mean; a1 -= mean;
      const float rstd = rsqrtf(group_sum_dpp(a0 * a0 + a1 * a1, 64, lane) * (1.f / (float)kStemC) + eps);
      float2 o;
      o.x = a0 * rstd * g0 + be0;
      o.y = a1 * rstd * g1 + be1;
      *reinterpret_cast<float2*>(y + (((size_t)n * OH + oy) * OW + ox) * kStemC + 2 * lane) = o;
    }
  }
}

// --------------------------------------------------------------------------------------------------
// Tail of the geometry head (GDRN_double_mask.py:128-160 + conv_pnp_net.py:120-134) on the NHWC result of the class-sliced
// output layer.  in: f32[B*HW][pitch], channels [vis | full (double mask) | x | y | z | region 0..R] (R = 64 regions + bg).
//   pnp_in f32[B*HW][96] = [(x-0.5) ex, (y-0.5) ey, (z-0.5) ez | coord2d u, v | softmax(region[1..64]) | 27 zeros]
//       (ConvPnPNet's in-place de-normalisation by the object extent, its torch.cat with the region softmax, Cin padded to the
//        next multiple of 32 for the implicit-GEMM convolution)
//   planes f32[n_planes][B*HW]: vis, (full,) x, y, z as they appear in out_dict (normalised xyz)
// One workgroup of 256 threads = 64 pixels; 4 lanes per pixel share the 64-way softmax (16 logits each, quad shuffles).
// --------------------------------------------------------------------------------------------------
constexpr int kTailPix = 64, kTailIn = 72, kTailOut = 96;
__global__ __launch_bounds__(256) void head_tail_kernel(const float* __restrict__ in, int pitch, const float* __restrict__ coord2d,
                                                        const float* __restrict__ extents, float* __restrict__ pnp_in,
                                                        float* __restrict__ planes, int hw, long n_pix, int double_mask) {
  __shared__ float T[kTailPix][kTailIn + 1];
  __shared__ float O[kTailPix][kTailOut + 4];
  const int tid = threadIdx.x;
  const long p0 = (long)blockIdx.x * kTailPix;
  for (int idx = tid; idx < kTailPix * (kTailIn / 4); idx += 256) {
    const int row = idx / (kTailIn / 4), c4 = idx - row * (kTailIn / 4);
    const float4 v = ld4(in + (size_t)(p0 + row) * pitch + 4 * c4);
    T[row][4 * c4] = v.x; T[row][4 * c4 + 1] = v.y; T[row][4 * c4 + 2] = v.z; T[row][4 * c4 + 3] = v.w;
  }
  __syncthreads();
  const int p = tid >> 2, part = tid & 3;
  const long pix = p0 + p;
  const int b = (int)(pix / hw), q = (int)(pix - (long)b * hw);
  const int kx = double_mask ? 2 : 1, kr = kx + 3;   // first xyz channel, region background channel
  // softmax over region[1..64] (the background channel kr is dropped, GDRN_double_mask.py:148)
  float e[16];
  float m = -3.402823466e38f;
#pragma unroll
  for (int i = 0; i < 16; ++i) { e[i] = T[p][kr + 1 + part * 16 + i]; m = fmaxf(m, e[i]); }
  m = fmaxf(m, __shfl_xor(m, 1, 64));
  m = fmaxf(m, __shfl_xor(m, 2, 64));
  float ssum = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) { e[i] = expf(e[i] - m); ssum += e[i]; }
  ssum += __shfl_xor(ssum, 1, 64);
  ssum += __shfl_xor(ssum, 2, 64);
#pragma unroll
  for (int i = 0; i < 16; ++i) O[p][5 + part * 16 + i] = e[i] / ssum;
  if (part == 0) {
#pragma unroll
    for (int c = 0; c < 3; ++c) O[p][c] = (T[p][kx + c] - 0.5f) * extents[3 * (size_t)b + c];
    O[p][3] = coord2d[((size_t)b * 2) * hw + q];
    O[p][4] = coord2d[((size_t)b * 2 + 1) * hw + q];
    const int n_planes = kx + 3;
    for (int k = 0; k < n_planes; ++k) planes[(size_t)k * n_pix + pix] = T[p][k];
  } else {
    // 27 padding channels, 9 per lane
#pragma unroll
    for (int i = 0; i < 9; ++i) O[p][69 + (part - 1) * 9 + i] = 0.f;
  }
  __syncthreads();
  for (int idx = tid; idx < kTailPix * (kTailOut / 4); idx += 256) {
    const int row = idx / (kTailOut / 4), c4 = idx - row * (kTailOut / 4);
    st4(pnp_in + (size_t)(p0 + row) * kTailOut + 4 * c4, make_float4(O[row][4 * c4], O[row][4 * c4 + 1], O[row][4 * c4 + 2], O[row][4 * c4 + 3]));
  }
}

template <int TH, int TW>
int launch_dwconv(const float* x, const float* w49c, const float* bias, const float* ln_w, const float* ln_b, float* y, int N,
                  int H, int W, int C, float eps, int y_rows, hipStream_t st) {
  const int Q = C / 4;
  const long n_tiles = (long)N * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
  // buffer-load rows need the image (n) and tile row of a wave to be uniform: one tile per wave or more (Q >= 64), or the
  // 64 / Q tiles of a wave side by side in one tile row; the row must fit a 32-bit buffer range
  const int tiles_x = (W + TW - 1) / TW;
  const int buf_rows = ((Q >= 64) || (tiles_x % (64 / Q) == 0)) && (long)W * C * 4 < (1l << 31) ? 1 : 0;
  const size_t wbytes = (size_t)49 * C * sizeof(float);
  const bool fuse = ln_w && ln_b;
  if (wbytes <= 112 * 1024 && gdrnpp::option_dwconv_lds_w()) {
    // weights in LDS, persistent workgroups: 512 threads when the weight set allows only one workgroup per CU
    const int threads = wbytes > 52 * 1024 ? 512 : 256;
    const long n_groups = (n_tiles + threads / Q - 1) / (threads / Q);
    // CU count, LDS attribute and occupancy are per (device, kernel, launch shape): looked up once, not per launch
    struct Cfg { int dev; bool fuse; int threads; size_t wbytes; int n_cu, per_cu; };
    static std::mutex mu;   // one cache per (TH, TW) instantiation
    static Cfg cache[16];
    static int n_cached = 0;
    int dev = 0, n_cu = 0, per_cu = 1;
    GDRNPP_HIP_TRY(hipGetDevice(&dev));
    {
      std::lock_guard<std::mutex> lock(mu);
      const Cfg* hit = nullptr;
      for (int i = 0; i < n_cached; ++i)
        if (cache[i].dev == dev && cache[i].fuse == fuse && cache[i].threads == threads && cache[i].wbytes == wbytes) hit = &cache[i];
      if (!hit) {
        Cfg c{dev, fuse, threads, wbytes, 0, 1};
        GDRNPP_HIP_TRY(hipDeviceGetAttribute(&c.n_cu, hipDeviceAttributeMultiprocessorCount, dev));
        const void* fn = fuse ? (const void*)dwconv7_ln_kernel<true, true, TH, TW> : (const void*)dwconv7_ln_kernel<false, true, TH, TW>;
        if (int rc = gdrnpp::ensure_dynamic_lds(fn, 112 * 1024)) return rc;
        GDRNPP_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&c.per_cu, fn, threads, wbytes));
        if (c.per_cu < 1) c.per_cu = 1;
        cache[n_cached < 16 ? n_cached++ : 15] = c;
        hit = &cache[n_cached - 1 < 15 ? n_cached - 1 : 15];
      }
      n_cu = hit->n_cu;
      per_cu = hit->per_cu;
    }
    long blocks = (long)n_cu * per_cu;
    if (blocks > n_groups) blocks = n_groups;
    if (fuse) {
      hipLaunchKernelGGL((dwconv7_ln_kernel<true, true, TH, TW>), dim3((unsigned)blocks), dim3(threads), wbytes, st, x, w49c, bias,
                         ln_w, ln_b, y, N, H, W, C, eps, buf_rows, y_rows);
    } else {
      hipLaunchKernelGGL((dwconv7_ln_kernel<false, true, TH, TW>), dim3((unsigned)blocks), dim3(threads), wbytes, st, x, w49c, bias,
                         nullptr, nullptr, y, N, H, W, C, eps, buf_rows, y_rows);
    }
  } else {
    const int tiles_per_block = 256 / Q;
    const long blocks = (n_tiles + tiles_per_block - 1) / tiles_per_block;
    GDRNPP_REQUIRE(blocks < (1l << 31), GDRNPP_ELIMIT, "gdrnpp_dwconv7x7_ln_nhwc: grid too large");
    const size_t pad = (size_t)gdrnpp::option_dwconv_lds_pad();      // experiment only (0 in the product): LDS the workgroup holds without using it
    if (pad > 64 * 1024) {
      const void* fn = fuse ? (const void*)dwconv7_ln_kernel<true, false, TH, TW> : (const void*)dwconv7_ln_kernel<false, false, TH, TW>;
      if (int rc = gdrnpp::ensure_dynamic_lds(fn, 112 * 1024)) return rc;
    }
    if (fuse) {
      hipLaunchKernelGGL((dwconv7_ln_kernel<true, false, TH, TW>), dim3((unsigned)blocks), dim3(256), pad, st, x, w49c, bias, ln_w,
                         ln_b, y, N, H, W, C, eps, buf_rows, y_rows);
    } else {
      hipLaunchKernelGGL((dwconv7_ln_kernel<false, false, TH, TW>), dim3((unsigned)blocks), dim3(256), pad, st, x, w49c, bias,
                         nullptr, nullptr, y, N, H, W, C, eps, buf_rows, y_rows);
    }
  }
  return gdrnpp::check_launch("gdrnpp_dwconv7x7_ln_nhwc");
}


}  // namespace


extern "C" {

int gdrnpp_dwconv7x7_ln_nhwc(const float* x, const float* w49c, const float* bias, const float* ln_w,
                             const float* ln_b, float* y, int N, int H, int W, int C, float eps, void* stream) {
  return gdrnpp_dwconv7x7_ln_nhwc_rows(x, w49c, bias, ln_w, ln_b, y, N, H, W, C, eps, 0, stream);
}

int gdrnpp_dwconv7x7_ln_nhwc_rows(const float* x, const float* w49c, const float* bias, const float* ln_w,
                                  const float* ln_b, float* y, int N, int H, int W, int C, float eps, int y_rows, void* stream) {
  GDRNPP_REQUIRE(x && w49c && bias && y, GDRNPP_EINVAL, "gdrnpp_dwconv7x7_ln_nhwc: null pointer");
  GDRNPP_REQUIRE(!y_rows || C % 8 == 0, GDRNPP_ELIMIT, "gdrnpp_dwconv7x7_ln_nhwc_rows: the f16x2-rows output needs C %% 8 == 0 (C=%d)", C);
  GDRNPP_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0, GDRNPP_EINVAL, "gdrnpp_dwconv7x7_ln_nhwc: N=%d H=%d W=%d C=%d", N,
                 H, W, C);
  GDRNPP_REQUIRE(C % 4 == 0 && C / 4 <= 256 && 256 % (C / 4) == 0, GDRNPP_ELIMIT,
                 "gdrnpp_dwconv7x7_ln_nhwc: C=%d must give a power-of-two quad count <= 256", C);
  const int Q = C / 4;
  // tile per thread: the largest of 2x8 / 2x4 / 1x4 that still gives the chip ~1.5 waves per SIMD (1024 SIMDs); the small
  // tiles re-read more halo and weights per output, which only pays when the launch is latency-bound (few ROIs)
  auto waves_of = [&](int th, int tw) { return (double)N * ((H + th - 1) / th) * ((W + tw - 1) / tw) * Q / 64.0; };
  const int force = gdrnpp::option_dwconv_tile();
  int cfg = waves_of(2, 8) >= 1536.0 ? 0 : (waves_of(2, 4) >= 1536.0 ? 1 : 2);
  if (force >= 0 && force <= 2) cfg = force;
  if (cfg == 0) return launch_dwconv<2, 8>(x, w49c, bias, ln_w, ln_b, y, N, H, W, C, eps, y_rows ? 1 : 0, (hipStream_t)stream);
  if (cfg == 1) return launch_dwconv<2, 4>(x, w49c, bias, ln_w, ln_b, y, N, H, W, C, eps, y_rows ? 1 : 0, (hipStream_t)stream);
  return launch_dwconv<1, 4>(x, w49c, bias, ln_w, ln_b, y, N, H, W, C, eps, y_rows ? 1 : 0, (hipStream_t)stream);
}

int gdrnpp_layernorm_nhwc(const float* x, const float* weight, const float* bias, float* y, long n_pix, int C,
                          float eps, void* stream) {
  GDRNPP_REQUIRE(x && weight && bias && y, GDRNPP_EINVAL, "gdrnpp_layernorm_nhwc: null pointer");
  GDRNPP_REQUIRE(n_pix > 0 && C > 0, GDRNPP_EINVAL, "gdrnpp_layernorm_nhwc: n_pix=%ld C=%d", n_pix, C);
  GDRNPP_REQUIRE(C % 4 == 0 && C / 4 <= 256 && 256 % (C / 4) == 0, GDRNPP_ELIMIT,
                 "gdrnpp_layernorm_nhwc: C=%d must give a power-of-two quad count <= 256", C);
  const int ppb = 256 / (C / 4);
  const long n_groups = (n_pix + (long)ppb * LN_PIX - 1) / ((long)ppb * LN_PIX);
  const long blocks = n_groups < 256 * 16 ? n_groups : 256 * 16;  // grid-stride beyond 16 workgroups per CU
  hipLaunchKernelGGL(layernorm_nhwc_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, weight, bias,
                     y, n_pix, C, eps);
  return gdrnpp::check_launch("gdrnpp_layernorm_nhwc");
}

int gdrnpp_upsample_bilinear2x_nhwc(const float* x, float* y, int N, int H, int W, int C, void* stream) {
  GDRNPP_REQUIRE(x && y, GDRNPP_EINVAL, "gdrnpp_upsample_bilinear2x_nhwc: null pointer");
  GDRNPP_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, GDRNPP_EINVAL,
                 "gdrnpp_upsample_bilinear2x_nhwc: N=%d H=%d W=%d C=%d (C %% 4 == 0 required)", N, H, W, C);
  const long total = (long)N * H * W * (C / 4);          // one thread per channel quad of a 2 x 2 output block
  const long blocks = (total + 255) / 256;
  hipLaunchKernelGGL(upsample2x_kernel, dim3((unsigned)(blocks < 65536 * 8 ? blocks : 65536 * 8)), dim3(256), 0,
                     (hipStream_t)stream, x, y, N, H, W, C);
  return gdrnpp::check_launch("gdrnpp_upsample_bilinear2x_nhwc");
}

size_t gdrnpp_groupnorm_workspace_bytes(int N, int HW, int G) {
  const int P = HW >= 1024 ? 64 : (HW >= 64 ? 8 : 1);
  return sizeof(double) * 2 * (size_t)N * P * G;
}

int gdrnpp_groupnorm_act_nhwc(const float* x, const float* gamma, const float* beta, float* y, void* workspace,
                              int N, int HW, int C, int G, float eps, int act_gelu, void* stream) {
  GDRNPP_REQUIRE(x && gamma && beta && y && workspace, GDRNPP_EINVAL, "gdrnpp_groupnorm_act_nhwc: null pointer");
  GDRNPP_REQUIRE(N > 0 && HW > 0 && C > 0 && G > 0 && C % G == 0, GDRNPP_EINVAL,
                 "gdrnpp_groupnorm_act_nhwc: N=%d HW=%d C=%d G=%d", N, HW, C, G);
  const int cpg = C / G, Q = C / 4;
  GDRNPP_REQUIRE(C % 4 == 0 && cpg % 4 == 0 && Q <= 256 && 256 % Q == 0 && G <= 64 && N <= 65535, GDRNPP_ELIMIT,
                 "gdrnpp_groupnorm_act_nhwc: unsupported shape C=%d G=%d N=%d", C, G, N);
  const int P = HW >= 1024 ? 64 : (HW >= 64 ? 8 : 1);
  hipStream_t st = (hipStream_t)stream;
  if ((size_t)HW * C * 4 <= (1u << 20) && x != y) {     // small images: statistics + normalisation in one launch
    if (act_gelu)
      hipLaunchKernelGGL(gn_small_kernel<true>, dim3(N), dim3(1024), sizeof(double) * 2048, st, x, gamma, beta, y, HW, C, G, eps);
    else
      hipLaunchKernelGGL(gn_small_kernel<false>, dim3(N), dim3(1024), sizeof(double) * 2048, st, x, gamma, beta, y, HW, C, G, eps);
    return gdrnpp::check_launch("gdrnpp_groupnorm_act_nhwc");
  }
  hipLaunchKernelGGL(gn_stats_kernel, dim3(P, N), dim3(256), sizeof(double) * 512, st, x, (double*)workspace, HW, C, G,
                     P);
  const long total = (long)HW * Q;
  // every block first rebuilds mean/rstd from the P partials (a ~2 us prologue): give it >= 16 float4 per thread
  long bx = (total + 256 * 16 - 1) / (256 * 16);
  if (bx > 64) bx = 64;
  if (bx < 1) bx = 1;
  if (act_gelu)
    hipLaunchKernelGGL(gn_apply_kernel<true>, dim3((unsigned)bx, N), dim3(256), 0, st, x, (const double*)workspace,
                       gamma, beta, y, HW, C, G, P, eps);
  else
    hipLaunchKernelGGL(gn_apply_kernel<false>, dim3((unsigned)bx, N), dim3(256), 0, st, x, (const double*)workspace,
                       gamma, beta, y, HW, C, G, P, eps);
  return gdrnpp::check_launch("gdrnpp_groupnorm_act_nhwc");
}

int gdrnpp_bias_act_nhwc(const float* x, const float* bias, const float* resid, float* y, long n_pix, int C, int relu,
                         void* stream) {
  if (n_pix == 0) return 0;
  GDRNPP_REQUIRE(x && bias && y, GDRNPP_EINVAL, "gdrnpp_bias_act_nhwc: null pointer");
  GDRNPP_REQUIRE(n_pix > 0 && C > 0 && C % 4 == 0, GDRNPP_EINVAL, "gdrnpp_bias_act_nhwc: n_pix=%ld C=%d (C %% 4 == 0)", n_pix, C);
  const long total4 = n_pix * (C / 4);
  const long blocks = (total4 + 255) / 256;
  hipLaunchKernelGGL(bias_act_kernel, dim3((unsigned)(blocks < 65536 * 8 ? blocks : 65536 * 8)), dim3(256), 0,
                     (hipStream_t)stream, x, bias, resid, y, total4, C / 4, relu);
  return gdrnpp::check_launch("gdrnpp_bias_act_nhwc");
}

int gdrnpp_deconv_col2im_nhwc(const float* cols, const float* bias, float* y, int N, int H, int W, int C, int KS, int stride,
                              int pad, int out_pad, void* stream) {
  GDRNPP_REQUIRE(cols && y, GDRNPP_EINVAL, "gdrnpp_deconv_col2im_nhwc: null pointer");
  GDRNPP_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && KS > 0 && stride > 0 && pad >= 0 && out_pad >= 0 &&
                     out_pad < stride,
                 GDRNPP_EINVAL, "gdrnpp_deconv_col2im_nhwc: bad shape");
  const int OH = (H - 1) * stride - 2 * pad + KS + out_pad, OW = (W - 1) * stride - 2 * pad + KS + out_pad;
  GDRNPP_REQUIRE(OH > 0 && OW > 0, GDRNPP_EINVAL, "gdrnpp_deconv_col2im_nhwc: empty output");
  const long total = (long)N * OH * OW * (C / 4);
  const long blocks = (total + 255) / 256;
  hipLaunchKernelGGL(deconv_col2im_kernel, dim3((unsigned)(blocks < 65536 * 8 ? blocks : 65536 * 8)), dim3(256), 0,
                     (hipStream_t)stream, cols, bias, y, H, W, C, KS, stride, pad, OH, OW, total);
  return gdrnpp::check_launch("gdrnpp_deconv_col2im_nhwc");
}

int gdrnpp_deconv_col2im_gn_nhwc(const float* cols, const float* bias, float* y, double* gn_partials, int N, int H, int W, int C,
                                 int KS, int stride, int pad, int out_pad, int G, void* stream) {
  GDRNPP_REQUIRE(cols && y && gn_partials, GDRNPP_EINVAL, "gdrnpp_deconv_col2im_gn_nhwc: null pointer");
  GDRNPP_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && KS > 0 && stride > 0 && pad >= 0 && out_pad >= 0 &&
                     out_pad < stride && G > 0 && C % G == 0,
                 GDRNPP_EINVAL, "gdrnpp_deconv_col2im_gn_nhwc: bad shape");
  const int OH = (H - 1) * stride - 2 * pad + KS + out_pad, OW = (W - 1) * stride - 2 * pad + KS + out_pad;
  GDRNPP_REQUIRE(OH > 0 && OW > 0, GDRNPP_EINVAL, "gdrnpp_deconv_col2im_gn_nhwc: empty output");
  const int cpg = C / G, Q = C / 4, HW = OH * OW;
  GDRNPP_REQUIRE(cpg % 4 == 0 && Q <= 256 && 256 % Q == 0 && G <= 64 && N <= 65535, GDRNPP_ELIMIT,
                 "gdrnpp_deconv_col2im_gn_nhwc: unsupported shape C=%d G=%d N=%d", C, G, N);
  const int P = HW >= 1024 ? 64 : (HW >= 64 ? 8 : 1);      // = gdrnpp_groupnorm_workspace_bytes / gdrnpp_groupnorm_act_nhwc
  hipLaunchKernelGGL(deconv_col2im_gn_kernel, dim3(P, N), dim3(256), sizeof(double) * 512, (hipStream_t)stream, cols, bias, y,
                     gn_partials, H, W, C, KS, stride, pad, OH, OW, G, P);
  return gdrnpp::check_launch("gdrnpp_deconv_col2im_gn_nhwc");
}

int gdrnpp_groupnorm_apply_nhwc(const float* x, const double* partials, int P, const float* gamma, const float* beta,
                                float* y, int N, int HW, int C, int G, float eps, int act_gelu, void* stream) {
  GDRNPP_REQUIRE(x && partials && gamma && beta && y, GDRNPP_EINVAL, "gdrnpp_groupnorm_apply_nhwc: null pointer");
  GDRNPP_REQUIRE(N > 0 && HW > 0 && C > 0 && G > 0 && P > 0 && C % G == 0, GDRNPP_EINVAL,
                 "gdrnpp_groupnorm_apply_nhwc: N=%d HW=%d C=%d G=%d P=%d", N, HW, C, G, P);
  const int cpg = C / G, Q = C / 4;
  GDRNPP_REQUIRE(C % 4 == 0 && cpg % 4 == 0 && Q <= 256 && 256 % Q == 0 && G <= 64 && N <= 65535, GDRNPP_ELIMIT,
                 "gdrnpp_groupnorm_apply_nhwc: unsupported shape C=%d G=%d N=%d", C, G, N);
  const long total = (long)HW * Q;
  long bx = (total + 256 * 16 - 1) / (256 * 16);
  if (bx > 64) bx = 64;
  if (bx < 1) bx = 1;
  hipStream_t st = (hipStream_t)stream;
  if (act_gelu)
    hipLaunchKernelGGL(gn_apply_kernel<true>, dim3((unsigned)bx, N), dim3(256), 0, st, x, partials, gamma, beta, y, HW, C, G, P, eps);
  else
    hipLaunchKernelGGL(gn_apply_kernel<false>, dim3((unsigned)bx, N), dim3(256), 0, st, x, partials, gamma, beta, y, HW, C, G, P, eps);
  return gdrnpp::check_launch("gdrnpp_groupnorm_apply_nhwc");
}

int gdrnpp_head_tail_nhwc(const float* out_nhwc, int pitch, const float* coord2d, const float* extents, float* pnp_in,
                          float* planes, int b, int hw, int double_mask, void* stream) {
  if (b == 0) return 0;
  GDRNPP_REQUIRE(out_nhwc && coord2d && extents && pnp_in && planes, GDRNPP_EINVAL, "gdrnpp_head_tail_nhwc: null pointer");
  GDRNPP_REQUIRE(b > 0 && hw > 0 && hw % kTailPix == 0 && pitch >= kTailIn && pitch % 4 == 0, GDRNPP_EINVAL,
                 "gdrnpp_head_tail_nhwc: b=%d hw=%d (multiple of %d) pitch=%d (>= %d, multiple of 4)", b, hw, kTailPix, pitch, kTailIn);
  const long n_pix = (long)b * hw;
  hipLaunchKernelGGL(head_tail_kernel, dim3((unsigned)(n_pix / kTailPix)), dim3(256), 0, (hipStream_t)stream, out_nhwc, pitch,
                     coord2d, extents, pnp_in, planes, hw, n_pix, double_mask ? 1 : 0);
  return gdrnpp::check_launch("gdrnpp_head_tail_nhwc");
}

int gdrnpp_stem_conv4x4_ln(const float* x_nchw, const float* weight, const float* bias, const float* ln_weight,
                           const float* ln_bias, float* y_nhwc, int N, int H, int W, int Cout, float eps, void* stream) {
  if (N == 0) return 0;
  GDRNPP_REQUIRE(x_nchw && weight && ln_weight && ln_bias && y_nhwc, GDRNPP_EINVAL, "gdrnpp_stem_conv4x4_ln: null pointer");
  GDRNPP_REQUIRE(N > 0 && H > 0 && W > 0 && H % 4 == 0 && W % 16 == 0 && W <= 1024 && Cout == kStemC, GDRNPP_ELIMIT,
                 "gdrnpp_stem_conv4x4_ln: N=%d H=%d W=%d Cout=%d (H %% 4, W %% 16, W <= 1024, Cout = %d)", N, H, W, Cout, kStemC);
  const long n_rows = (long)N * (H / 4);
  const long blocks = n_rows < 256 * 8 ? n_rows : 256 * 8;
  hipLaunchKernelGGL(stem_conv_ln_kernel, dim3((unsigned)blocks), dim3(256), 12 * (W / 4) * sizeof(float4), (hipStream_t)stream, x_nchw,
                     weight, bias, ln_weight, ln_bias, y_nhwc, N, H, W, eps);
  return gdrnpp::check_launch("gdrnpp_stem_conv4x4_ln");
}

}  // extern "C"
