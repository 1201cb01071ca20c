// Fast depth refinement (render-and-compare) + stand-alone depth render on gfx950.
// SURVEY.md §8 rows a8, a8.1, a14.
//
// Behavioural spec: core/gdrn_modeling/engine/gdrn_evaluator.py:461-573 (process_depth_refine;
// runnable twin demo/predictor_gdrn.py:195-286), get_out_mask engine_utils.py:315-333,
// cv2.resize 256->64 (INTER_LINEAR at exact scale 4 = mean of the 2x2 centre of each 4x4 block),
// vispy render lib/render_vispy/renderer.py (see raster.hpp).
//
// Design: ONE workgroup of 1024 threads (16 waves) per ROI runs every refinement iteration
// on chip — nothing but the final translation goes back to HBM:
//   prologue  per-ROI mask min/max (shuffle + LDS reduce), per-pixel query base
//             ||xyz||*mask and the 64x64 sensor-depth crop, all held in registers
//             (4 pixels per thread, pixel p = k*1024 + tid -> coalesced 4 KiB rows);
//   render    triangles are distributed over threads, transformed and set up in fp64,
//             and resolved with ds_min_u32 on a 16 KiB LDS z-buffer holding float-Z bits
//             (rounding to float is monotonic, so min-of-rounded == rounded-min); triangles
//             with a large pixel bbox are queued in LDS and rasterised by the whole
//             workgroup so one thread never serialises a big triangle;
//   compare   q-map, fp64 block sum, fp32 normalise, block max, threshold, LDS compaction of
//             the selected depth differences, bitonic sort in LDS, np.median semantics,
//             fp64 weighted centroid, ray through K_crop^-1, t += ray * median.
// HBM traffic per ROI is the algorithmic minimum except for the mesh, which is re-read per
// iteration from L2 (all ROIs of a class share it): 4 maps x 16 KiB + the used quarter of the
// 256x256 depth crop + I x (12 V + 12 F) mesh bytes.
#include "common.hpp"
#include "raster.hpp"
#include <cfloat>

namespace {

using namespace gdrnpp;

constexpr int kT = 512;           // threads per ROI workgroup (8 waves; 1024 spills under the 128-VGPR cap)
constexpr int kWaves = kT / 64;
constexpr int kMaxPix = 4096;     // res*res limit for the refine kernel (res <= 64)
constexpr int kPPT = kMaxPix / kT;
constexpr int kLargeArea = 48;    // bbox pixels above which a triangle goes to the cooperative queue
constexpr int kMaxLarge = 512;
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

// rasterise the whole mesh into an LDS z-buffer of float-Z bits (must be pre-filled with kInfBits)
__device__ void raster_mesh_u32(const MeshView& m, const double* K, const double* R, const double* t, int res,
                                double z_near, double z_far, unsigned* zbuf, int* s_large, int* s_nlarge) {
  if (threadIdx.x == 0) *s_nlarge = 0;
  __syncthreads();
  for (int f = threadIdx.x; f < m.nfaces; f += kT) {
    TriSetup s;
    face_setup(m, f, K, R, t, res, z_near, z_far, s);
    if (s.i_lo > s.i_hi || s.j_lo > s.j_hi) continue;
    const int area = (s.i_hi - s.i_lo + 1) * (s.j_hi - s.j_lo + 1);
    if (area > kLargeArea) {
      const int slot = atomicAdd(s_nlarge, 1);
      if (slot < kMaxLarge) { s_large[slot] = f; continue; }
    }
    for (int j = s.j_lo; j <= s.j_hi; ++j)
      for (int i = s.i_lo; i <= s.i_hi; ++i) {
        double Z;
        if (sample_triangle(s, i, j, z_near, z_far, Z, nullptr))
          atomicMin(&zbuf[j * res + i], __float_as_uint((float)Z));
      }
  }
  __syncthreads();
  const int nl = min(*s_nlarge, kMaxLarge);
  for (int q = 0; q < nl; ++q) {
    TriSetup s;
    face_setup(m, s_large[q], K, R, t, res, z_near, z_far, s);
    const int bw = s.i_hi - s.i_lo + 1, bh = s.j_hi - s.j_lo + 1;
    for (int p = threadIdx.x; p < bw * bh; p += kT) {
      const int j = s.j_lo + p / bw, i = s.i_lo + p % bw;
      double Z;
      if (sample_triangle(s, i, j, z_near, z_far, Z, nullptr))
        atomicMin(&zbuf[j * res + i], __float_as_uint((float)Z));
    }
  }
  __syncthreads();
}

// ---- block reductions over kT threads (results broadcast to every thread) -------------
__device__ __forceinline__ double block_sum(double v, double* s_red) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) s_red[wave] = v;
  __syncthreads();
  double r = 0.0;
  for (int w = 0; w < kWaves; ++w) r += s_red[w];
  return r;
}
__device__ __forceinline__ float block_max(float v, float* s_red) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) s_red[wave] = v;
  __syncthreads();
  float r = -FLT_MAX;
  for (int w = 0; w < kWaves; ++w) r = fmaxf(r, s_red[w]);
  return r;
}
__device__ __forceinline__ float block_min(float v, float* s_red) { return -block_max(-v, s_red); }

__global__ __launch_bounds__(kT) void depth_refine_kernel(
    const float* __restrict__ verts, const int* __restrict__ faces, const int* __restrict__ vert_off,
    const int* __restrict__ face_off, const int* __restrict__ obj, const float* __restrict__ coor_x,
    const float* __restrict__ coor_y, const float* __restrict__ coor_z, const float* __restrict__ mask_raw,
    const float* __restrict__ roi_depth, const float* __restrict__ K_crop, const float* __restrict__ Rin,
    const float* __restrict__ t_in, double* __restrict__ t_out, float* __restrict__ debug_depth, int res, int iters,
    float threshold, int mask_type, int use_coor_z, float z_near, float z_far) {
  __shared__ unsigned zbuf[kMaxPix];
  __shared__ float sortbuf[kMaxPix];
  __shared__ int s_large[kMaxLarge];
  __shared__ int s_nlarge, s_nsel;
  __shared__ double s_redd[kWaves];
  __shared__ float s_redf[kWaves];
  __shared__ double s_t[3];

  const int bi = blockIdx.x, tid = threadIdx.x;
  const int hw = res * res;
  const MeshView mesh = mesh_of(verts, faces, vert_off, face_off, obj[bi]);

  double K[9], R[9], t[3];
  float Rf[9];
  for (int k = 0; k < 9; ++k) { K[k] = (double)K_crop[9 * (size_t)bi + k]; Rf[k] = Rin[9 * (size_t)bi + k]; R[k] = (double)Rf[k]; }
  for (int k = 0; k < 3; ++k) t[k] = (double)t_in[3 * (size_t)bi + k];

  // ---- prologue: mask normalisation, query base, sensor depth crop --------------------
  const float* mk = mask_raw + (size_t)bi * hw;
  float mraw[kPPT];
  float lo = FLT_MAX, hi = -FLT_MAX;
#pragma unroll
  for (int k = 0; k < kPPT; ++k) {
    const int p = k * kT + tid;
    mraw[k] = (p < hw) ? mk[p] : 0.f;
    if (p < hw) { lo = fminf(lo, mraw[k]); hi = fmaxf(hi, mraw[k]); }
  }
  float mmin = 0.f, mden = 1.f;
  if (mask_type == 0) {
    mmin = block_min(lo, s_redf);
    const float mmax = block_max(hi, s_redf);
    mden = mmax - mmin;  // no epsilon (engine_utils.py:325)
  }
  float qbase[kPPT], ds[kPPT];
  const int in_w = 4 * res;
  const float* dep = roi_depth + (size_t)bi * in_w * in_w;
#pragma unroll
  for (int k = 0; k < kPPT; ++k) {
    const int p = k * kT + tid;
    qbase[k] = 0.f; ds[k] = 0.f;
    if (p < hw) {
      float m = mraw[k];
      if (mask_type == 0) m = (m - mmin) / mden;
      else m = 1.f / (1.f + expf(-m));
      const float x = coor_x[(size_t)bi * hw + p], y = coor_y[(size_t)bi * hw + p], z = coor_z[(size_t)bi * hw + p];
      float qv;
      if (use_coor_z) qv = (Rf[6] * x + Rf[7] * y) + Rf[8] * z;  // z component of R @ xyz (evaluator :528-535)
      else qv = sqrtf((x * x + y * y) + z * z);                 // torch.norm(xyz, dim=-1) (:538-540)
      qbase[k] = qv * m;
      // cv2.resize(roi_depth, (res,res)) INTER_LINEAR, scale 4: source coordinate 4x+1.5
      const int yy = p / res, xx = p - yy * res;
      const float* r0 = dep + (size_t)(4 * yy + 1) * in_w + 4 * xx + 1;
      const float* r1 = r0 + in_w;
      const float h0 = r0[0] * 0.5f + r0[1] * 0.5f;
      const float h1 = r1[0] * 0.5f + r1[1] * 0.5f;
      ds[k] = h0 * 0.5f + h1 * 0.5f;
    }
  }

  for (int it = 0; it < iters; ++it) {
    // ---- render ------------------------------------------------------------------------
#pragma unroll
    for (int k = 0; k < kPPT; ++k) {
      const int p = k * kT + tid;
      if (p < hw) zbuf[p] = kInfBits;
    }
    if (tid == 0) s_nsel = 0;
    // the GL pipeline receives the pose as float32 uniforms
    const double tr[3] = {(double)(float)t[0], (double)(float)t[1], (double)(float)t[2]};
    raster_mesh_u32(mesh, K, R, tr, res, (double)z_near, (double)z_far, zbuf, s_large, &s_nlarge);

    // ---- query map -----------------------------------------------------------------------
    float ren[kPPT], q[kPPT];
    double part = 0.0;
#pragma unroll
    for (int k = 0; k < kPPT; ++k) {
      const int p = k * kT + tid;
      ren[k] = 0.f; q[k] = 0.f;
      if (p < hw) {
        const unsigned zb = zbuf[p];
        ren[k] = (zb == kInfBits) ? 0.f : __uint_as_float(zb);
        if (debug_depth) debug_depth[((size_t)bi * iters + it) * hw + p] = ren[k];
        const float rm = ren[k] > 0.f ? 1.f : 0.f, dm = ds[k] > 0.f ? 1.f : 0.f;
        q[k] = (qbase[k] * rm) * dm;
        part += (double)q[k];
      }
    }
    const float norm_sum = (float)block_sum(part, s_redd);
    if (norm_sum == 0.f) continue;  // evaluator :542-544 (uniform across the workgroup)

    float qm = -FLT_MAX;
    double sy = 0.0, sx = 0.0;
#pragma unroll
    for (int k = 0; k < kPPT; ++k) {
      const int p = k * kT + tid;
      if (p < hw) {
        q[k] = q[k] / norm_sum;
        qm = fmaxf(qm, q[k]);
        const int yy = p / res, xx = p - yy * res;
        sy += (double)yy * (double)q[k];  // int64 * float32 -> float64 (:553-555)
        sx += (double)xx * (double)q[k];
      }
    }
    const float qmax = block_max(qm, s_redf);
    const float thr = qmax * threshold;
    sy = block_sum(sy, s_redd);
    sx = block_sum(sx, s_redd);

    // ---- selected depth differences -> LDS, sort, median --------------------------------------
#pragma unroll
    for (int k = 0; k < kPPT; ++k) {
      const int p = k * kT + tid;
      if (p < hw && q[k] > thr) {
        const int slot = atomicAdd(&s_nsel, 1);
        sortbuf[slot] = ds[k] - ren[k];
      }
    }
    __syncthreads();
    const int nsel = s_nsel;
    if (nsel == 0) continue;
    int M = 1;
    while (M < nsel) M <<= 1;
    for (int i = nsel + tid; i < M; i += kT) sortbuf[i] = INFINITY;
    __syncthreads();
    for (int k2 = 2; k2 <= M; k2 <<= 1)
      for (int j = k2 >> 1; j > 0; j >>= 1) {
        for (int i = tid; i < M; i += kT) {
          const int ixj = i ^ j;
          if (ixj > i) {
            const float a = sortbuf[i], b = sortbuf[ixj];
            const bool asc = (i & k2) == 0;
            if ((a > b) == asc) { sortbuf[i] = b; sortbuf[ixj] = a; }
          }
        }
        __syncthreads();
      }
    if (tid == 0) {
      // np.median: odd -> middle; even -> float32 mean of the two middle elements
      const float med = (nsel & 1) ? sortbuf[nsel >> 1] : (sortbuf[(nsel >> 1) - 1] + sortbuf[nsel >> 1]) / 2.f;
      // ray = inv(K_crop) @ (x, y, 1); np.linalg.inv on float32 returns float32
      const double a = K[0], b = K[1], c = K[2], d = K[3], e = K[4], f = K[5], g = K[6], h = K[7], i9 = K[8];
      const double det = a * (e * i9 - f * h) - b * (d * i9 - f * g) + c * (d * h - e * g);
      double Ki[9] = {(e * i9 - f * h) / det, (c * h - b * i9) / det, (b * f - c * e) / det,
                      (f * g - d * i9) / det, (a * i9 - c * g) / det, (c * d - a * f) / det,
                      (d * h - e * g) / det, (b * g - a * h) / det, (a * e - b * d) / det};
      for (int k = 0; k < 9; ++k) Ki[k] = (double)(float)Ki[k];
      double ray[3];
      for (int r = 0; r < 3; ++r) ray[r] = (Ki[3 * r] * sx + Ki[3 * r + 1] * sy) + Ki[3 * r + 2] * 1.0;
      const double rz = ray[2];
      for (int r = 0; r < 3; ++r) s_t[r] = t[r] + (ray[r] / rz) * (double)med;
    }
    __syncthreads();
    for (int r = 0; r < 3; ++r) t[r] = s_t[r];
    __syncthreads();
  }
  if (tid == 0)
    for (int r = 0; r < 3; ++r) t_out[3 * (size_t)bi + r] = t[r];
}

// ---- stand-alone render: depth (+ optional object-space xyz) -------------------------------
__global__ __launch_bounds__(kT) void render_depth_kernel(const float* __restrict__ verts,
                                                          const int* __restrict__ faces,
                                                          const int* __restrict__ vert_off,
                                                          const int* __restrict__ face_off,
                                                          const int* __restrict__ obj, const float* __restrict__ Kin,
                                                          const float* __restrict__ Rin, const float* __restrict__ tin,
                                                          float* __restrict__ depth, float* __restrict__ xyz, int res,
                                                          float z_near, float z_far) {
  extern __shared__ unsigned long long zkey[];  // (float Z bits << 32) | face id
  const int bi = blockIdx.x, tid = threadIdx.x, hw = res * res;
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
      TriSetup s2;
      face_setup(mesh, ff, K, R, t, res, z_near, z_far, s2);
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
    GDRNPP_HIP_TRY(hipFuncSetAttribute((const void*)render_depth_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       lds));
  }
  hipLaunchKernelGGL(render_depth_kernel, dim3(b), dim3(kT), lds, (hipStream_t)stream, meshes->verts, meshes->faces,
                     meshes->vert_off, meshes->face_off, obj, K, R, t, depth, xyz, res, z_near, z_far);
  return gdrnpp::check_launch("gdrnpp_render_depth");
}

int gdrnpp_depth_refine(const gdrnpp_meshes* meshes, const int* obj, const float* coor_x, const float* coor_y,
                        const float* coor_z, const float* mask_raw, const float* roi_depth, const float* K_crop,
                        const float* R, const float* t_in, double* t_out, float* debug_depth, int b, int res,
                        int iters, float threshold, int mask_type, int use_coor_z, float z_near, float z_far,
                        void* stream) {
  if (int rc = check_meshes(meshes, "gdrnpp_depth_refine")) return rc;
  GDRNPP_REQUIRE(obj && coor_x && coor_y && coor_z && mask_raw && roi_depth && K_crop && R && t_in && t_out,
                 GDRNPP_EINVAL, "gdrnpp_depth_refine: null pointer");
  GDRNPP_REQUIRE(b > 0 && res > 0 && iters >= 0, GDRNPP_EINVAL, "gdrnpp_depth_refine: b=%d res=%d iters=%d", b, res,
                 iters);
  GDRNPP_REQUIRE(res * res <= kMaxPix, GDRNPP_ELIMIT, "gdrnpp_depth_refine: res=%d > 64", res);
  GDRNPP_REQUIRE(mask_type == 0 || mask_type == 1, GDRNPP_EINVAL, "gdrnpp_depth_refine: mask_type=%d", mask_type);
  hipLaunchKernelGGL(depth_refine_kernel, dim3(b), dim3(kT), 0, (hipStream_t)stream, meshes->verts, meshes->faces,
                     meshes->vert_off, meshes->face_off, obj, coor_x, coor_y, coor_z, mask_raw, roi_depth, K_crop, R,
                     t_in, t_out, debug_depth, res, iters, threshold, mask_type, use_coor_z, z_near, z_far);
  return gdrnpp::check_launch("gdrnpp_depth_refine");
}

}  // extern "C"
