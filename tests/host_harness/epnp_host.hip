// Test harness (not part of the product): the device EPnP math of gdrnpp_bop2022_amd/csrc/epnp_ransac.hip compiled for the
// HOST as well, so that the -m "not gpu" tests can compare it with the LAPACK-based oracle without a device.  Every
// __device__ function of the included source becomes __host__ __device__; the wave shuffles get inert host overloads
// (the serial instantiation epnp_solve<false> never calls them).
#include <hip/hip_runtime.h>
__host__ static inline double __shfl_xor(double v, int, int) { return v; }
__host__ static inline int __shfl_xor(int v, int, int) { return v; }
__host__ static inline int __shfl(int v, int, int) { return v; }
__host__ static inline double __shfl(double v, int, int) { return v; }
#undef __device__
#define __device__ __attribute__((host)) __attribute__((device))
#include EPNP_SOURCE
namespace gdrnpp { void set_error(const char*, ...) {} }
extern "C" int host_epnp(const float* uv, const float* pw, int n, const float* K9, double* R, double* t) {
  Cam cam = Cam{(double)K9[0], (double)K9[4], (double)K9[2], (double)K9[5]};
  PointSet ps{uv, pw, nullptr, nullptr, cam, 0.f, n, n};
  return epnp_solve<false>(ps, cam, 0, R, t) ? 1 : 0;
}
extern "C" int host_p3p(const float* uv, const float* pw, const float* K9, double* R, double* t) {
  Cam cam = Cam{(double)K9[0], (double)K9[4], (double)K9[2], (double)K9[5]};
  return p3p_4points(uv, pw, cam, R, t) ? 1 : 0;
}
