// Shared host-side helpers for libgdrnpp_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdarg>
#include <cstdlib>
#include "../../include/gdrnpp_hip.h"

namespace gdrnpp {

// thread-local last-error text (gdrnpp_last_error)
void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

#define GDRNPP_REQUIRE(cond, code, ...) \
  do {                                  \
    if (!(cond)) {                      \
      gdrnpp::set_error(__VA_ARGS__);   \
      return (code);                    \
    }                                   \
  } while (0)

#define GDRNPP_HIP_TRY(expr)                                        \
  do {                                                              \
    hipError_t _e = (expr);                                         \
    if (_e != hipSuccess) {                                         \
      gdrnpp::set_error("%s: %s", #expr, hipGetErrorString(_e));    \
      return (int)_e;                                               \
    }                                                               \
  } while (0)

constexpr int kWave = 64;  // CDNA4 wavefront

// GELU(x) = x Phi(x), the erf form of torch.nn.GELU, branch-free:  x Phi(x) = max(x, 0) - u Phi(-u) with u = min(|x|, 5.9)
// and Phi(-u) = exp2(q(u)), q the degree-8 fit of log2 Phi(-u) on [0, 5.9] that minimises the error of Phi (tools/fit_gelu_poly.py).
// 13 instructions, one of them transcendental; ocml's erff is ~45 with divergent branches, which made the GELU epilogue a
// quarter of an fc1 tile of the split GEMM.  Error against fp64 over [-9, 9] (emulated fp32, v_exp_f32 at 1 ulp): 2.7e-7
// absolute, 8.9e-8 of max(|x|, 1) — below the 4.5e-7 / 1.0e-7 of the fp32 formula 0.5 x (1 + erf(x / sqrt 2)) evaluated with a
// correctly rounded erf.  Below -5.9 the result is -1e-8 instead of -0 .. -1e-8; NaN propagates.
__device__ __forceinline__ float gelu_erf(float x) {
  const float u = fminf(fabsf(x), 5.9f);
  float q = -2.7721912374545354e-06f;
  q = fmaf(q, u, 3.862218727590516e-05f);
  q = fmaf(q, u, -0.00018255332543049008f);
  q = fmaf(q, u, -0.000145858692121692f);
  q = fmaf(q, u, 0.007075459696352482f);
  q = fmaf(q, u, -0.052505023777484894f);
  q = fmaf(q, u, -0.45920491218566895f);
  q = fmaf(q, u, -1.1511057615280151f);
  q = fmaf(q, u, -1.0f);
  const float r = fmaf(-u, __builtin_amdgcn_exp2f(q), fmaxf(x, 0.f));
  return x != x ? x : r;
}

// Device scratch of the host-pointer shims of the reference ABI (farthest_point_sampling, uncertainty_pnp): grow-only, one
// block per host thread, re-allocated when the thread switches device; kept for the life of the process (no hipMalloc /
// hipFree per call).  Returns nullptr with the error text set when the allocation fails.
void* shim_scratch(size_t bytes);
// hipFuncAttributeMaxDynamicSharedMemorySize is a per-(kernel, device) setting: raised when `bytes` exceeds what this process
// has already set for the pair on the current device, not on every launch.  Returns 0 or a GDRNPP_* code (error text set).
int ensure_dynamic_lds(const void* kernel, int bytes);

// process-wide tuning switches (gdrnpp_set_option): read on the launch path instead of getenv
int option_split_gemm_glds();   // 1: 256-row split-GEMM tiles use the LDS-DMA kernel
int option_split_gemm_mi4();    // -1: by tile count, 0 / 1: force 128- / 256-row tiles (A/B measurements)
int option_splitk_small_tiles();   // 1 (default): unsplit problems of gdrnpp_linear_f32_splitk pick their tile height by tile count (128 rows below 256 tiles of 256x128: 8 ROIs 4.65 -> 4.41 ms per step), 0: always 256-row tiles
int option_split2_wide();        // 1: 256x256 block tiles of the three-product kernels when N % 256 == 0 (A/B; default off: measured slower)
int option_mlp_fused_pipe();    // 1 (default): software-pipelined tile loop of the fused MLP; 0: plain loop (A/B)
int option_dwconv_tile();       // -1 (default): by launch size; 0 / 1 / 2: force the 2x8 / 2x4 / 1x4 pixel tile of dwconv7x7+LN
int option_dwconv_lds_w();      // 1 (default): [49][C] weights in LDS + persistent workgroups where they fit; 0: weights through L1 / L2, no LDS (A/B:
                                // a workgroup that needs 98 KB of LDS cannot start on a CU that holds one 80 KB GEMM workgroup of another stream)
int option_dwconv_lds_pad();    // experiment only: dynamic LDS bytes the no-LDS form allocates all the same (profiles/r06_dwconv_shared.txt)

}  // namespace gdrnpp
