// Fast depth refinement (render-and-compare) + stand-alone depth render on gfx950.
// SURVEY.md §8 rows a8, a8.1, a14.
//
// Behavioural spec: core/gdrn_modeling/engine/gdrn_evaluator.py:461-573 (process_depth_refine;
// runnable twin demo/predictor_gdrn.py:195-286), get_out_mask engine_utils.py:315-333,
// cv2.resize 256->64 (INTER_LINEAR at exact scale 4 = mean of the 2x2 centre of each 4x4 block),
// vispy render lib/render_vispy/renderer.py (see raster.hpp).
//
// Design: ONE workgroup of 512 threads (8 waves) per ROI runs every refinement iteration on chip — only the refined
// translation / the finished pose record goes back to HBM:
//   prologue  (optional) K_crop from the camera, ROI centre and scale (get_K_crop_resize fused); per-ROI mask min/max
//             (shuffle + LDS reduce); the per-pixel query base ||xyz||*mask and the 64x64 sensor-depth crop go to LDS;
//   stage     per iteration the V model points are transformed ONCE into homogeneous pixel space (fp64) and staged together
//             with their projections u = h0/h2, v = h1/h2 (fp32 {u, v, z} for the candidate test, once per vertex instead of fp64 divisions per
//             corner) — in LDS (40 B/vertex, meshes up to 2688 vertices) or, for larger meshes, in a per-ROI slice of a global
//             workspace (same kernel, template parameter);
//   render    triangles -> threads; candidate pixels from the staged projections first (most sub-pixel triangles have none and
//             stop there), corners gathered from the stage, fp64 edge functions, ds_min_u32 on a 16 KiB LDS
//             z-buffer of float-Z bits (rounding to float is monotonic, so min-of-rounded == rounded-min); triangles with a
//             large pixel bbox are queued in LDS and rasterised by a whole wave each;
//   compare   q-map, fp64 block sum, fp32 normalise, block max, threshold, np.median of the selected depth differences by
//             a two-rank radix select (4 x 8-bit passes, LDS histograms), fp64 weighted centroid, ray through K_crop^-1,
//             t += ray * median;
//   epilogue  (optional) the f32[16] pose record R | t | score | obj | roi_id | valid (pack_pose_records fused).
// HBM traffic per ROI is the algorithmic minimum except for the mesh, which is re-read per iteration from L2 (all ROIs of
// a class share it): 4 maps x 16 KiB + the used quarter of the 256x256 depth crop + I x (12 V + 12 F) mesh bytes.
#include "common.hpp"
#include "raster.hpp"
#include <cfloat>

namespace {

using namespace gdrnpp;

constexpr int kT = 512;           // threads of a render workgroup
constexpr int kMaxPix = 4096;     // res*res limit for the refine kernel (res <= 64)
constexpr int kLargeArea = 32;    // bbox pixel centres above which a triangle is rasterised by its whole wave
constexpr unsigned kInfBits = 0x7f800000u;

struct MeshView {
  const float* verts;
  const int* faces;
  int nfaces;
};

__device__ __forceinline__ MeshView mesh_of(const float* verts, const int* faces, const int* vert_off,
                                            const int* face_off, int obj) {
  MeshView m;
  m.verts = verts + 3 * (size_t)vert_off[obj];
  m.faces = faces + 3 * (size_t)face_off[obj];
  m.nfaces = face_off[obj + 1] - face_off[obj];
  return m;
}

__device__ __forceinline__ void face_setup(const MeshView& m, int f, const double* K, const double* R,
                                           const double* t, int res, double z_near, double z_far, TriSetup& s) {
  const int i0 = m.faces[3 * f], i1 = m.faces[3 * f + 1], i2 = m.faces[3 * f + 2];
  double h0[3], h1[3], h2[3];
  project_vertex(m.verts + 3 * (size_t)i0, K, R, t, h0);
  project_vertex(m.verts + 3 * (size_t)i1, K, R, t, h1);
  project_vertex(m.verts + 3 * (size_t)i2, K, R, t, h2);
  setup_triangle(h0, h1, h2, res, res, z_near, z_far, s);
}

// phase stamps of workgroup 0 (s_memtime cycles): [0] start [1] prologue done, then per iteration
// [2+5i] staged [3+5i] rastered [4+5i] reduced [5+5i] median [6+5i] updated; read by gdrnpp_debug_refine_profile
__device__ long long g_refine_prof[16];
#define PROF_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0 && (k) < 16) g_refine_prof[(k)] = (long long)__builtin_readcyclecounter(); } while (0)

#ifndef GDRNPP_REFINE_THREADS
#define GDRNPP_REFINE_THREADS 512
#endif
constexpr int kTS = GDRNPP_REFINE_THREADS;
constexpr int kWavesS = kTS / 64;
constexpr int kPPTS = kMaxPix / kTS;
constexpr int kStageBytes = 40;        // staged per vertex: homogeneous pixel-space h[3] (fp64) + {u = h0/h2, v = h1/h2, z} as one fp32 float4
constexpr int kMaxStagedVerts = 2688;  // 105 KiB of LDS (with the 54 KiB of static LDS: the whole 160 KiB of a CU)
constexpr int kMaxLargeS = 1024;       // queue of wave-rasterised triangles (overflow falls back to per-lane)

__device__ __forceinline__ double block_sum_s(double v, double* s_red) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) s_red[wave] = v;
  __syncthreads();
  double r = 0.0;
  for (int w = 0; w < kWavesS; ++w) r += s_red[w];
  return r;
}
__device__ __forceinline__ float block_max_s(float v, float* s_red) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) s_red[wave] = v;
  __syncthreads();
  float r = -FLT_MAX;
  for (int w = 0; w < kWavesS; ++w) r = fmaxf(r, s_red[w]);
  return r;
}

__device__ __forceinline__ unsigned f2key(float f) {  // order-preserving float -> uint
  const unsigned b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// keys[k] valid where sel[k]; returns the elements of 0-based ranks r0 and r1 (r0 <= r1) among the selected keys.
__device__ void radix_select2(const unsigned* keys, const bool* sel, int nkeys, int r0, int r1, unsigned (*hist)[256],
                              unsigned* s_pref, unsigned& out0, unsigned& out1) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned pref0 = 0, pref1 = 0;
  int rank0 = r0, rank1 = r1;
  for (int pass = 3; pass >= 0; --pass) {
    const int shift = 8 * pass;
    const unsigned himask = (pass == 3) ? 0u : (0xffffffffu << (shift + 8));
    if (threadIdx.x < 512) hist[threadIdx.x >> 8][threadIdx.x & 255] = 0;
    __syncthreads();
    for (int k = 0; k < nkeys; ++k)
      if (sel[k]) {
        const unsigned key = keys[k], byte = (key >> shift) & 255u;
        if ((key & himask) == (pref0 & himask)) atomicAdd(&hist[0][byte], 1u);
        if ((key & himask) == (pref1 & himask)) atomicAdd(&hist[1][byte], 1u);
      }
    __syncthreads();
    if (wave < 2) {  // wave 0 resolves rank0, wave 1 resolves rank1: 4 bins per lane, shuffle prefix scan
      const unsigned* h = hist[wave];
      const int want = wave == 0 ? rank0 : rank1;
      const unsigned c0 = h[4 * lane], c1 = h[4 * lane + 1], c2 = h[4 * lane + 2], c3 = h[4 * lane + 3];
      const unsigned tot = c0 + c1 + c2 + c3;
      unsigned incl = tot;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
      }
      const unsigned excl = incl - tot;
      const bool mine = (unsigned)want >= excl && (unsigned)want < incl;
      if (mine) {
        unsigned r = (unsigned)want - excl, b;
        if (r < c0) b = 0;
        else if (r < c0 + c1) { b = 1; r -= c0; }
        else if (r < c0 + c1 + c2) { b = 2; r -= c0 + c1; }
        else { b = 3; r -= c0 + c1 + c2; }
        s_pref[2 * wave] = (unsigned)(4 * lane) + b;  // winning byte
        s_pref[2 * wave + 1] = r;                     // rank inside that bin
      }
    }
    __syncthreads();
    pref0 |= s_pref[0] << shift; rank0 = (int)s_pref[1];
    pref1 |= s_pref[2] << shift; rank1 = (int)s_pref[3];
    __syncthreads();
  }
  out0 = pref0;
  out1 = pref1;
}

struct RefineArgs {
  const float* verts; const int* faces; const int* vert_off; const int* face_off; int n_obj;
  const int* obj;
  const float *coor_x, *coor_y, *coor_z, *mask_raw, *roi_depth;
  const float* K_crop;                                 // f32[b,9], or null: computed from cam / center / scale / out_res
  const float *cam, *center, *scale; float out_res;    // get_K_crop_resize (camera_geometry.py:6-21) inputs
  const float *R, *t_in;
  double* t_out;                                       // f64[b,3] or null
  float* debug_depth;                                  // f32[b,iters,res,res] or null
  float* rec; const float* score; const int* roi_id;   // f32[b,16] pose records or null (score / roi_id nullable)
  double* hv_global; int hv_stride;                    // vertex stage of meshes too large for LDS: [b] x (hv_stride x 40 bytes)
  int max_verts;                                       // of the mesh set (LDS stage layout)
  int res, iters; float threshold; int mask_type, use_coor_z; float z_near, z_far;
};

// STAGED: the transformed vertices of the current iteration live in LDS; otherwise in this ROI's slice of a.hv_global
template <bool STAGED>
__global__ __launch_bounds__(kTS) void depth_refine_kernel(const RefineArgs a) {
  const float* __restrict__ coor_x = a.coor_x; const float* __restrict__ coor_y = a.coor_y; const float* __restrict__ coor_z = a.coor_z;
  const float* __restrict__ mask_raw = a.mask_raw; const float* __restrict__ roi_depth = a.roi_depth;
  const float* __restrict__ Rin = a.R; const float* __restrict__ t_in = a.t_in;
  float* __restrict__ debug_depth = a.debug_depth;
  const int res = a.res, iters = a.iters, mask_type = a.mask_type, use_coor_z = a.use_coor_z;
  const float threshold = a.threshold, z_near = a.z_near, z_far = a.z_far;
  extern __shared__ double hv_lds[];  // [V][3] homogeneous pixel-space vertices of the current iteration (STAGED)
  __shared__ unsigned zbuf[kMaxPix];
  __shared__ float s_qbase[kMaxPix], s_ds[kMaxPix];  // iteration-invariant per-pixel terms (kept out of the VGPRs)
  __shared__ unsigned hist[2][256];
  __shared__ unsigned s_pref[4];
  __shared__ int s_large[kMaxLargeS];
  __shared__ int s_nsel, s_nlarge;
  __shared__ double s_redd[kWavesS], s_redd2[kWavesS];
  __shared__ float s_redf[kWavesS];
  __shared__ double s_K[9], s_R[9], s_t[3];
  __shared__ float s_Rf[3];

  const int bi = blockIdx.x, tid = threadIdx.x;
  const int hw = res * res;
  const int ob = a.obj[bi];
  // an object id outside the mesh set (a detector with another class map): the ROI keeps the network translation and
  // its record is marked invalid — no out-of-bounds mesh offsets
  const bool known = (unsigned)ob < (unsigned)a.n_obj;
  const int obs = known ? ob : 0;
  const float* mverts = a.verts + 3 * (size_t)a.vert_off[obs];
  const int nverts = a.vert_off[obs + 1] - a.vert_off[obs];
  const int* mfaces = a.faces + 3 * (size_t)a.face_off[obs];
  const int nfaces = a.face_off[obs + 1] - a.face_off[obs];
  const double zn = (double)z_near, zf = (double)z_far;
  double* hv;
  if constexpr (STAGED) hv = hv_lds;
  else hv = a.hv_global + (size_t)bi * a.hv_stride * (kStageBytes / 8);
  // staged projections behind the fp64 vertices (vertex count rounded up to even keeps the float4s 16-byte aligned)
  const int vpad = STAGED ? ((a.max_verts + 1) & ~1) : a.hv_stride;
  float4* const uvz = reinterpret_cast<float4*>(hv + 3 * (size_t)vpad);
  const int n_it = known ? iters : 0;

  PROF_STAMP(0);
  if (tid < 9) {
    float kc;
    if (a.K_crop) {
      kc = a.K_crop[9 * (size_t)bi + tid];
    } else {  // get_K_crop_resize: K' = diag(r, r, 1) (K - [0 0 x0; 0 0 y0; 0 0 0]), float32 like the torch ops
      const float* k = a.cam + 9 * (size_t)bi;
      const float sc = a.scale[bi];
      const float x0 = a.center[2 * bi] - sc / 2, y0 = a.center[2 * bi + 1] - sc / 2;
      const float r = a.out_res / sc;
      kc = tid == 2 ? (k[2] - x0) * r : tid == 5 ? (k[5] - y0) * r : tid < 6 ? k[tid] * r : k[tid];
    }
    s_K[tid] = (double)kc;
    s_R[tid] = (double)Rin[9 * (size_t)bi + tid];
  }
  if (tid < 3) { s_t[tid] = (double)t_in[3 * (size_t)bi + tid]; s_Rf[tid] = Rin[9 * (size_t)bi + 6 + tid]; }

  // ---- prologue: mask normalisation, query base, sensor depth crop ------------------------------------------
  // every global load of the prologue is issued before the first barrier (the mask min / max reduction): loads do not move
  // across __syncthreads, and 8 pixels x 8 loads in flight per thread hide the HBM latency that 24 k cycles were spent on
  const float* mk = mask_raw + (size_t)bi * hw;
  const int in_w = 4 * res;
  const float* dep = roi_depth + (size_t)bi * in_w * in_w;
  float mraw[kPPTS], cx[kPPTS], cy[kPPTS], cz[kPPTS], d00[kPPTS], d01[kPPTS], d10[kPPTS], d11[kPPTS];
  float lo = FLT_MAX, hi = -FLT_MAX;
#pragma unroll
  for (int k = 0; k < kPPTS; ++k) {
    const int p = k * kTS + tid;
    mraw[k] = 0.f; cx[k] = cy[k] = cz[k] = 0.f; d00[k] = d01[k] = d10[k] = d11[k] = 0.f;
    if (p < hw) {
      mraw[k] = mk[p];
      cx[k] = coor_x[(size_t)bi * hw + p]; cy[k] = coor_y[(size_t)bi * hw + p]; cz[k] = coor_z[(size_t)bi * hw + p];
      const int yy = p / res, xx = p - yy * res;
      const float* r0 = dep + (size_t)(4 * yy + 1) * in_w + 4 * xx + 1;
      const float* r1 = r0 + in_w;
      d00[k] = r0[0]; d01[k] = r0[1]; d10[k] = r1[0]; d11[k] = r1[1];
    }
  }
#pragma unroll
  for (int k = 0; k < kPPTS; ++k)
    if (k * kTS + tid < hw) { lo = fminf(lo, mraw[k]); hi = fmaxf(hi, mraw[k]); }
  float mmin = 0.f, mden = 1.f;
  if (mask_type == 0) {
    mmin = -block_max_s(-lo, s_redf);
    const float mmax = block_max_s(hi, s_redf);
    mden = mmax - mmin;
  } else {
    __syncthreads();
  }
  const float r20 = s_Rf[0], r21 = s_Rf[1], r22 = s_Rf[2];
#pragma unroll
  for (int k = 0; k < kPPTS; ++k) {
    const int p = k * kTS + tid;
    if (p < hw) {
      float m = mraw[k];
      if (mask_type == 0) m = (m - mmin) / mden;
      else if (mask_type == 1) m = 1.f / (1.f + expf(-m));  // 2: already a probability / label (CE argmax)
      const float x = cx[k], y = cy[k], z = cz[k];
      float qv;
      if (use_coor_z) qv = (r20 * x + r21 * y) + r22 * z;
      else qv = sqrtf((x * x + y * y) + z * z);
      s_qbase[p] = qv * m;
      const float h0 = d00[k] * 0.5f + d01[k] * 0.5f;
      const float h1 = d10[k] * 0.5f + d11[k] * 0.5f;
      s_ds[p] = h0 * 0.5f + h1 * 0.5f;
    }
  }

  PROF_STAMP(1);
  for (int it = 0; it < n_it; ++it) {
    // ---- stage the transformed model points, clear the z-buffer ------------------------------------------------
    {
      double K[9], R[9], tr[3];
#pragma unroll
      for (int k = 0; k < 9; ++k) { K[k] = s_K[k]; R[k] = s_R[k]; }
#pragma unroll
      for (int k = 0; k < 3; ++k) tr[k] = (double)(float)s_t[k];  // GL receives float32 uniforms
      for (int v = tid; v < nverts; v += kTS) {
        double h[3];
        project_vertex(mverts + 3 * (size_t)v, K, R, tr, h);
        double* o = hv + 3 * (size_t)v;
        o[0] = h[0]; o[1] = h[1]; o[2] = h[2];
        // projections for the candidate test, once per vertex, in single precision (the test carries a 1e-3 px margin): one
        // 16-byte gather per corner instead of three 8-byte ones — the rasteriser is bound by these LDS gathers
        const float hz = (float)h[2];
        uvz[v] = make_float4((float)h[0] / hz, (float)h[1] / hz, hz, 0.f);
      }
    }
#pragma unroll
    for (int k = 0; k < kPPTS; ++k) {
      const int p = k * kTS + tid;
      if (p < hw) zbuf[p] = kInfBits;
    }
    if (tid == 0) { s_nsel = 0; s_nlarge = 0; }
    __syncthreads();
    PROF_STAMP(2 + 5 * it);

    // ---- rasterise: triangles -> threads, corners gathered from LDS.  A triangle whose bbox holds more than
    //      kLargeArea pixel centres is queued in LDS and rasterised by a whole wave (16 waves drain the queue in
    //      parallel), so one lane never serialises a big triangle --------------------------------------------------
    // candidates first (three z, then u, v of the staged vertices), edge planes only for triangles that have any: same
    // TriSetup as setup_triangle
    auto staged_setup = [&](int f, TriSetup& s) -> bool {
      const int i0 = mfaces[3 * f], i1 = mfaces[3 * f + 1], i2 = mfaces[3 * f + 2];
      s.D = 0.0;
      const int cand = triangle_bbox_f32(uvz[i0], uvz[i1], uvz[i2], res, res, z_near, z_far, s);
      if (cand == 0) return false;
      const double* p0 = hv + 3 * (size_t)i0;
      const double* p1 = hv + 3 * (size_t)i1;
      const double* p2 = hv + 3 * (size_t)i2;
      const double h0[3] = {p0[0], p0[1], p0[2]}, h1[3] = {p1[0], p1[1], p1[2]}, h2[3] = {p2[0], p2[1], p2[2]};
      if (cand == 2) {   // near-plane clipping / non-finite projections: the exact fp64 candidate box
        double uv0[2] = {0, 0}, uv1[2] = {0, 0}, uv2[2] = {0, 0};
        if (fmin(h0[2], fmin(h1[2], h2[2])) >= zn) {
          uv0[0] = h0[0] / h0[2]; uv0[1] = h0[1] / h0[2]; uv1[0] = h1[0] / h1[2]; uv1[1] = h1[1] / h1[2];
          uv2[0] = h2[0] / h2[2]; uv2[1] = h2[1] / h2[2];
        }
        if (!triangle_bbox(h0, h1, h2, uv0, uv1, uv2, res, res, zn, zf, s)) return false;
      }
      return triangle_edges(h0, h1, h2, s);
    };
    for (int f = tid; f < nfaces; f += kTS) {
      TriSetup s;
      if (!staged_setup(f, s)) continue;
      if ((s.i_hi - s.i_lo + 1) * (s.j_hi - s.j_lo + 1) > kLargeArea) {
        const int slot = atomicAdd(&s_nlarge, 1);
        if (slot < kMaxLargeS) { s_large[slot] = f; continue; }
      }
      for (int j = s.j_lo; j <= s.j_hi; ++j)
        for (int i = s.i_lo; i <= s.i_hi; ++i) {
          double Z;
          if (sample_triangle(s, i, j, zn, zf, Z, nullptr)) atomicMin(&zbuf[j * res + i], __float_as_uint((float)Z));
        }
    }
    __syncthreads();
    {
      const int nl = min(s_nlarge, kMaxLargeS);
      const int lane = tid & 63;
      for (int qd = tid >> 6; qd < nl; qd += kWavesS) {
        const int f = s_large[qd];
        TriSetup s;
        staged_setup(f, s);
        const int bw = s.i_hi - s.i_lo + 1, bh = s.j_hi - s.j_lo + 1;
        for (int p = lane; p < bw * bh; p += 64) {
          const int j = s.j_lo + p / bw, i = s.i_lo + p % bw;
          double Z;
          if (sample_triangle(s, i, j, zn, zf, Z, nullptr)) atomicMin(&zbuf[j * res + i], __float_as_uint((float)Z));
        }
      }
    }
    __syncthreads();
    PROF_STAMP(3 + 5 * it);

    // ---- query map -----------------------------------------------------------------------------------------------
    float ren[kPPTS], q[kPPTS], ds[kPPTS];
    double part = 0.0;
#pragma unroll
    for (int k = 0; k < kPPTS; ++k) {
      const int p = k * kTS + tid;
      ren[k] = 0.f; q[k] = 0.f; ds[k] = 0.f;
      if (p < hw) {
        ds[k] = s_ds[p];
        const unsigned zb = zbuf[p];
        ren[k] = (zb == kInfBits) ? 0.f : __uint_as_float(zb);
        if (debug_depth) debug_depth[((size_t)bi * iters + it) * hw + p] = ren[k];
        const float rm = ren[k] > 0.f ? 1.f : 0.f, dm = ds[k] > 0.f ? 1.f : 0.f;
        q[k] = (s_qbase[p] * rm) * dm;
        part += (double)q[k];
      }
    }
    const float norm_sum = (float)block_sum_s(part, s_redd);
    if (norm_sum == 0.f) continue;

    float qm = -FLT_MAX;
    double sy = 0.0, sx = 0.0;
#pragma unroll
    for (int k = 0; k < kPPTS; ++k) {
      const int p = k * kTS + tid;
      if (p < hw) {
        q[k] = q[k] / norm_sum;
        qm = fmaxf(qm, q[k]);
        const int yy = p / res, xx = p - yy * res;
        sy += (double)yy * (double)q[k];
        sx += (double)xx * (double)q[k];
      }
    }
    // qmax, sum y q, sum x q in ONE block reduction (wave shuffles, one LDS exchange, two barriers instead of six)
    float qmax;
    {
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        qm = fmaxf(qm, __shfl_xor(qm, off, 64));
        sy += __shfl_xor(sy, off, 64);
        sx += __shfl_xor(sx, off, 64);
      }
      const int lane = tid & 63, wave = tid >> 6;
      __syncthreads();
      if (lane == 0) { s_redf[wave] = qm; s_redd[wave] = sy; s_redd2[wave] = sx; }
      __syncthreads();
      qmax = -FLT_MAX; sy = 0.0; sx = 0.0;
      for (int w = 0; w < kWavesS; ++w) { qmax = fmaxf(qmax, s_redf[w]); sy += s_redd[w]; sx += s_redd2[w]; }
    }
    const float thr = qmax * threshold;

    PROF_STAMP(4 + 5 * it);
    // ---- median of the selected depth differences: two-rank radix select --------------------------------------------
    unsigned keys[kPPTS];
    bool sel[kPPTS];
    int mine = 0;
#pragma unroll
    for (int k = 0; k < kPPTS; ++k) {
      const int p = k * kTS + tid;
      sel[k] = (p < hw) && (q[k] > thr);
      keys[k] = f2key(ds[k] - ren[k]);
      mine += sel[k] ? 1 : 0;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mine += __shfl_xor(mine, off, 64);
    if ((tid & 63) == 0 && mine) atomicAdd(&s_nsel, mine);
    __syncthreads();
    const int nsel = s_nsel;
    if (nsel == 0) continue;
    unsigned k0, k1;
    radix_select2(keys, sel, kPPTS, (nsel - 1) >> 1, nsel >> 1, hist, s_pref, k0, k1);
    PROF_STAMP(5 + 5 * it);
    if (tid == 0) {
      const float a0 = key2f(k0), a1 = key2f(k1);
      const float med = (nsel & 1) ? a1 : (a0 + a1) / 2.f;  // np.median
      const double a = s_K[0], b = s_K[1], c = s_K[2], d = s_K[3], e = s_K[4], f = s_K[5], g = s_K[6], h = s_K[7],
                   i9 = s_K[8];
      const double det = a * (e * i9 - f * h) - b * (d * i9 - f * g) + c * (d * h - e * g);
      double Ki[9] = {(e * i9 - f * h) / det, (c * h - b * i9) / det, (b * f - c * e) / det,
                      (f * g - d * i9) / det, (a * i9 - c * g) / det, (c * d - a * f) / det,
                      (d * h - e * g) / det, (b * g - a * h) / det, (a * e - b * d) / det};
      for (int k = 0; k < 9; ++k) Ki[k] = (double)(float)Ki[k];
      double ray[3];
      for (int r = 0; r < 3; ++r) ray[r] = (Ki[3 * r] * sx + Ki[3 * r + 1] * sy) + Ki[3 * r + 2] * 1.0;
      const double rz = ray[2];
      for (int r = 0; r < 3; ++r) s_t[r] = s_t[r] + (ray[r] / rz) * (double)med;
    }
    __syncthreads();
    PROF_STAMP(6 + 5 * it);
  }
  __syncthreads();
  if (tid < 3 && a.t_out) a.t_out[3 * (size_t)bi + tid] = s_t[tid];
  if (a.rec && tid < 16) {  // pose_prediction_to_json's fields as one f32[16] record (gdrn_evaluator.py:636-665)
    float v;
    if (tid < 9) v = Rin[9 * (size_t)bi + tid];
    else if (tid < 12) v = (float)s_t[tid - 9];
    else if (tid == 12) v = a.score ? a.score[bi] : 1.f;
    else if (tid == 13) v = (float)ob;
    else if (tid == 14) v = a.roi_id ? (float)a.roi_id[bi] : (float)bi;
    else v = known ? 1.f : 0.f;
    a.rec[16 * (size_t)bi + tid] = v;
  }
}

// ---- stand-alone render: depth (+ optional object-space xyz) -------------------------------
__global__ __launch_bounds__(kT) void render_depth_kernel(const float* __restrict__ verts,
                                                          const int* __restrict__ faces,
                                                          const int* __restrict__ vert_off,
                                                          const int* __restrict__ face_off,
                                                          const int* __restrict__ obj, const float* __restrict__ Kin,
                                                          const float* __restrict__ Rin, const float* __restrict__ tin,
                                                          float* __restrict__ depth, float* __restrict__ xyz, int res,
                                                          float z_near, float z_far, int n_obj) {
  extern __shared__ unsigned long long zkey[];  // (float Z bits << 32) | face id
  const int bi = blockIdx.x, tid = threadIdx.x, hw = res * res;
  if ((unsigned)obj[bi] >= (unsigned)n_obj) {  // unknown object id: empty render
    for (int p = tid; p < hw; p += kT) {
      depth[(size_t)bi * hw + p] = 0.f;
      if (xyz) { float* dst = xyz + ((size_t)bi * hw + p) * 3; dst[0] = 0.f; dst[1] = 0.f; dst[2] = 0.f; }
    }
    return;
  }
  const MeshView mesh = mesh_of(verts, faces, vert_off, face_off, obj[bi]);
  double K[9], R[9], t[3];
  for (int k = 0; k < 9; ++k) { K[k] = (double)Kin[9 * (size_t)bi + k]; R[k] = (double)Rin[9 * (size_t)bi + k]; }
  for (int k = 0; k < 3; ++k) t[k] = (double)tin[3 * (size_t)bi + k];
  const unsigned long long kEmpty = ((unsigned long long)kInfBits << 32) | 0xffffffffull;
  for (int p = tid; p < hw; p += kT) zkey[p] = kEmpty;
  __syncthreads();
  // triangle-parallel for small bboxes, then block-cooperative over triangles with large ones
  for (int f0 = 0; f0 < mesh.nfaces; f0 += kT) {
    const int f = f0 + tid;
    bool large = false;
    TriSetup s;
    if (f < mesh.nfaces) {
      face_setup(mesh, f, K, R, t, res, z_near, z_far, s);
      if (s.i_lo <= s.i_hi && s.j_lo <= s.j_hi) {
        const int area = (s.i_hi - s.i_lo + 1) * (s.j_hi - s.j_lo + 1);
        if (area > kLargeArea) large = true;
        else
          for (int j = s.j_lo; j <= s.j_hi; ++j)
            for (int i = s.i_lo; i <= s.i_hi; ++i) {
              double Z;
              if (sample_triangle(s, i, j, z_near, z_far, Z, nullptr))
                atomicMin(&zkey[j * res + i], ((unsigned long long)__float_as_uint((float)Z) << 32) | (unsigned)f);
            }
      }
    }
    // wave-cooperative handling of large triangles (ballot over the wave, all 64 lanes rasterise one)
    unsigned long long bal = __ballot(large);
    const int lane = tid & 63;
    while (bal) {
      const int src = __ffsll((long long)bal) - 1;
      bal &= bal - 1;
      const int ff = __shfl(f, src, 64);
      const TriSetup s2 = shfl_setup(s, src);
      const int bw = s2.i_hi - s2.i_lo + 1, bh = s2.j_hi - s2.j_lo + 1;
      for (int p = lane; p < bw * bh; p += 64) {
        const int j = s2.j_lo + p / bw, i = s2.i_lo + p % bw;
        double Z;
        if (sample_triangle(s2, i, j, z_near, z_far, Z, nullptr))
          atomicMin(&zkey[j * res + i], ((unsigned long long)__float_as_uint((float)Z) << 32) | (unsigned)ff);
      }
    }
  }
  __syncthreads();
  for (int p = tid; p < hw; p += kT) {
    const unsigned long long key = zkey[p];
    const unsigned zb = (unsigned)(key >> 32);
    const bool hit = zb != kInfBits;
    depth[(size_t)bi * hw + p] = hit ? __uint_as_float(zb) : 0.f;
    if (xyz) {
      float o[3] = {0.f, 0.f, 0.f};
      if (hit) {
        const int f = (int)(key & 0xffffffffu);
        TriSetup s;
        face_setup(mesh, f, K, R, t, res, z_near, z_far, s);
        double Z, lam[3];
        const int j = p / res, i = p - j * res;
        if (sample_triangle(s, i, j, z_near, z_far, Z, lam)) {
          const float* v0 = mesh.verts + 3 * (size_t)mesh.faces[3 * f];
          const float* v1 = mesh.verts + 3 * (size_t)mesh.faces[3 * f + 1];
          const float* v2 = mesh.verts + 3 * (size_t)mesh.faces[3 * f + 2];
          for (int c = 0; c < 3; ++c)
            o[c] = (float)((lam[0] * (double)v0[c] + lam[1] * (double)v1[c]) + lam[2] * (double)v2[c]);
        }
      }
      float* dst = xyz + ((size_t)bi * hw + p) * 3;
      dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2];
    }
  }
}

int check_meshes(const gdrnpp_meshes* m, const char* who) {
  GDRNPP_REQUIRE(m && m->verts && m->faces && m->vert_off && m->face_off && m->n_obj > 0, GDRNPP_EINVAL,
                 "%s: invalid mesh set", who);
  return 0;
}

// hipFuncSetAttribute is a per-device setting: done once per (kernel, device), not per launch
template <typename F>
int raise_dynamic_lds_once(F kernel, int bytes, bool* done /*[64]*/) {
  int dev = 0;
  GDRNPP_HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || done[dev]) return 0;
  GDRNPP_HIP_TRY(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  done[dev] = true;
  return 0;
}

int launch_refine(const gdrnpp_meshes* meshes, RefineArgs a, int b, void* workspace, size_t workspace_bytes, hipStream_t st,
                  const char* who) {
  a.verts = meshes->verts; a.faces = meshes->faces; a.vert_off = meshes->vert_off; a.face_off = meshes->face_off;
  a.n_obj = meshes->n_obj;
  if (meshes->max_verts > 0 && meshes->max_verts <= kMaxStagedVerts) {
    static bool done[64];
    if (int rc = raise_dynamic_lds_once(depth_refine_kernel<true>, kStageBytes * kMaxStagedVerts, done)) return rc;
    a.hv_global = nullptr; a.hv_stride = 0; a.max_verts = meshes->max_verts;
    hipLaunchKernelGGL(depth_refine_kernel<true>, dim3(b), dim3(kTS), kStageBytes * ((meshes->max_verts + 1) & ~1), st, a);
  } else {
    GDRNPP_REQUIRE(meshes->max_verts > 0, GDRNPP_EINVAL, "%s: gdrnpp_meshes.max_verts must be set", who);
    const int vp = (meshes->max_verts + 1) & ~1;
    const size_t need = (size_t)b * vp * kStageBytes;
    GDRNPP_REQUIRE(workspace && workspace_bytes >= need, GDRNPP_EINVAL,
                   "%s: meshes of up to %d vertices need a workspace of %zu bytes (gdrnpp_depth_refine_workspace_bytes)", who,
                   meshes->max_verts, need);
    a.hv_global = (double*)workspace; a.hv_stride = vp; a.max_verts = meshes->max_verts;
    hipLaunchKernelGGL(depth_refine_kernel<false>, dim3(b), dim3(kTS), 0, st, a);
  }
  return gdrnpp::check_launch(who);
}

}  // namespace

extern "C" {

int gdrnpp_render_depth(const gdrnpp_meshes* meshes, const int* obj, const float* K, const float* R, const float* t,
                        float* depth, float* xyz, int b, int res, float z_near, float z_far, void* stream) {
  if (int rc = check_meshes(meshes, "gdrnpp_render_depth")) return rc;
  GDRNPP_REQUIRE(obj && K && R && t && depth, GDRNPP_EINVAL, "gdrnpp_render_depth: null pointer");
  GDRNPP_REQUIRE(b > 0 && res > 0, GDRNPP_EINVAL, "gdrnpp_render_depth: b=%d res=%d", b, res);
  GDRNPP_REQUIRE(res <= 128, GDRNPP_ELIMIT, "gdrnpp_render_depth: res=%d > 128 (z-buffer lives in LDS)", res);
  const int lds = res * res * (int)sizeof(unsigned long long);
  if (lds > 48 * 1024) {
    static bool done[64];
    if (int rc = raise_dynamic_lds_once(render_depth_kernel, 128 * 128 * (int)sizeof(unsigned long long), done)) return rc;
  }
  hipLaunchKernelGGL(render_depth_kernel, dim3(b), dim3(kT), lds, (hipStream_t)stream, meshes->verts, meshes->faces,
                     meshes->vert_off, meshes->face_off, obj, K, R, t, depth, xyz, res, z_near, z_far, meshes->n_obj);
  return gdrnpp::check_launch("gdrnpp_render_depth");
}

size_t gdrnpp_depth_refine_workspace_bytes(const gdrnpp_meshes* meshes, int b) {
  if (!meshes || b <= 0 || meshes->max_verts <= kMaxStagedVerts) return 0;
  return (size_t)b * ((meshes->max_verts + 1) & ~1) * kStageBytes;
}

int gdrnpp_depth_refine(const gdrnpp_meshes* meshes, const int* obj, const float* coor_x, const float* coor_y,
                        const float* coor_z, const float* mask_raw, const float* roi_depth, const float* K_crop,
                        const float* R, const float* t_in, double* t_out, float* debug_depth, int b, int res, int in_res,
                        int iters, float threshold, int mask_type, int use_coor_z, float z_near, float z_far,
                        void* workspace, size_t workspace_bytes, void* stream) {
  if (b == 0) return 0;
  if (int rc = check_meshes(meshes, "gdrnpp_depth_refine")) return rc;
  GDRNPP_REQUIRE(obj && coor_x && coor_y && coor_z && mask_raw && roi_depth && K_crop && R && t_in && t_out,
                 GDRNPP_EINVAL, "gdrnpp_depth_refine: null pointer");
  GDRNPP_REQUIRE(b > 0 && res > 0 && iters >= 0, GDRNPP_EINVAL, "gdrnpp_depth_refine: b=%d res=%d iters=%d", b, res,
                 iters);
  GDRNPP_REQUIRE(res * res <= kMaxPix, GDRNPP_ELIMIT, "gdrnpp_depth_refine: res=%d > 64", res);
  GDRNPP_REQUIRE(in_res == 4 * res, GDRNPP_ELIMIT,
                 "gdrnpp_depth_refine: roi_depth is %d x %d, the kernel reads the INPUT_RES = 4 x OUTPUT_RES crop (%d) the way "
                 "cv2.resize does at scale 4", in_res, in_res, 4 * res);
  GDRNPP_REQUIRE(mask_type >= 0 && mask_type <= 2, GDRNPP_EINVAL, "gdrnpp_depth_refine: mask_type=%d", mask_type);
  RefineArgs a{};
  a.obj = obj; a.coor_x = coor_x; a.coor_y = coor_y; a.coor_z = coor_z; a.mask_raw = mask_raw; a.roi_depth = roi_depth;
  a.K_crop = K_crop; a.R = R; a.t_in = t_in; a.t_out = t_out; a.debug_depth = debug_depth;
  a.res = res; a.iters = iters; a.threshold = threshold; a.mask_type = mask_type; a.use_coor_z = use_coor_z;
  a.z_near = z_near; a.z_far = z_far;
  return launch_refine(meshes, a, b, workspace, workspace_bytes, (hipStream_t)stream, "gdrnpp_depth_refine");
}

int gdrnpp_refine_to_records(const gdrnpp_meshes* meshes, const int* obj, const float* coor_x, const float* coor_y,
                             const float* coor_z, const float* mask_raw, const float* roi_depth, const float* cam,
                             const float* center, const float* scale, const float* R, const float* t_in,
                             const float* score, const int* roi_id, float* rec, int b, int res, int in_res, int iters,
                             float threshold, int mask_type, int use_coor_z, float z_near, float z_far, void* workspace,
                             size_t workspace_bytes, void* stream) {
  if (b == 0) return 0;
  if (int rc = check_meshes(meshes, "gdrnpp_refine_to_records")) return rc;
  GDRNPP_REQUIRE(obj && coor_x && coor_y && coor_z && mask_raw && roi_depth && cam && center && scale && R && t_in && rec,
                 GDRNPP_EINVAL, "gdrnpp_refine_to_records: null pointer");
  GDRNPP_REQUIRE(b > 0 && res > 0 && iters >= 0, GDRNPP_EINVAL, "gdrnpp_refine_to_records: b=%d res=%d iters=%d", b, res, iters);
  GDRNPP_REQUIRE(res * res <= kMaxPix && in_res == 4 * res, GDRNPP_ELIMIT,
                 "gdrnpp_refine_to_records: res=%d (<= 64) and in_res=%d (= 4 x res) required", res, in_res);
  GDRNPP_REQUIRE(mask_type >= 0 && mask_type <= 2, GDRNPP_EINVAL, "gdrnpp_refine_to_records: mask_type=%d", mask_type);
  RefineArgs a{};
  a.obj = obj; a.coor_x = coor_x; a.coor_y = coor_y; a.coor_z = coor_z; a.mask_raw = mask_raw; a.roi_depth = roi_depth;
  a.K_crop = nullptr; a.cam = cam; a.center = center; a.scale = scale; a.out_res = (float)res;
  a.R = R; a.t_in = t_in; a.t_out = nullptr; a.debug_depth = nullptr; a.rec = rec; a.score = score; a.roi_id = roi_id;
  a.res = res; a.iters = iters; a.threshold = threshold; a.mask_type = mask_type; a.use_coor_z = use_coor_z;
  a.z_near = z_near; a.z_far = z_far;
  return launch_refine(meshes, a, b, workspace, workspace_bytes, (hipStream_t)stream, "gdrnpp_refine_to_records");
}

/* debug: s_memtime stamps of workgroup 0 of the last staged refine launch (see g_refine_prof) */
int gdrnpp_debug_refine_profile(long long* h_out16) {
  GDRNPP_REQUIRE(h_out16, GDRNPP_EINVAL, "gdrnpp_debug_refine_profile: null pointer");
  GDRNPP_HIP_TRY(hipMemcpyFromSymbol(h_out16, HIP_SYMBOL(g_refine_prof), sizeof(long long) * 16));
  return 0;
}

}  // extern "C"
