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

// process-wide tuning switches (gdrnpp_set_option): read on the launch path instead of getenv
int option_split_gemm_glds();   // 1: 256-row split-GEMM tiles use the LDS-DMA kernel
int option_split_gemm_mi4();    // -1: by tile count, 0 / 1: force 128- / 256-row tiles (A/B measurements)

}  // namespace gdrnpp
