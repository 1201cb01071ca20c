// Diagnostic (not part of the product): packed-fp32 VALU results checked against the scalar instructions inside the kernel, to see
// whether a kernel sharing the chip changes them.  Built with -fno-slp-vectorize so the scalar side stays scalar.
#include <hip/hip_runtime.h>
#include <stdint.h>
using f2 = __attribute__((ext_vector_type(2))) float;
__device__ __forceinline__ f2 pk_mul(f2 a, f2 b) { f2 r; asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ f2 pk_add(f2 a, f2 b) { f2 r; asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ f2 pk_mul_bcast(f2 a, f2 b) { f2 r; asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ f2 pk_add_cross(f2 a, f2 b) { f2 r; asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float smul(float a, float b) { float r; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float sadd(float a, float b) { float r; asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
extern "C" __global__ __launch_bounds__(256) void pk_probe_kernel(unsigned* __restrict__ log, int max_log, int iters, float* __restrict__ sink) {
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  f2 a = {1.0f + (float)(t & 1023) * 0.001f, 2.0f - (float)(t & 511) * 0.002f};
  f2 b = {0.5f + (float)(threadIdx.x & 63) * 0.01f, 1.5f - (float)(threadIdx.x & 63) * 0.003f};
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    const f2 m = pk_mul(a, b), s = pk_add(a, b), mb = pk_mul_bcast(a, b), ac = pk_add_cross(a, b);
    const float m0 = smul(a.x, b.x), m1 = smul(a.y, b.y), s0 = sadd(a.x, b.x), s1 = sadd(a.y, b.y);
    const float mb0 = smul(a.x, b.x), mb1 = smul(a.x, b.y);          // op_sel_hi:[0,1]: src0's low half in both lanes of the pair
    const float ac0 = sadd(a.x, b.y), ac1 = sadd(a.y, b.x);          // op_sel:[0,1] op_sel_hi:[1,0]
    unsigned bad = 0;
    bad |= (__float_as_uint(m.x) != __float_as_uint(m0) || __float_as_uint(m.y) != __float_as_uint(m1)) ? 1u : 0u;
    bad |= (__float_as_uint(s.x) != __float_as_uint(s0) || __float_as_uint(s.y) != __float_as_uint(s1)) ? 2u : 0u;
    bad |= (__float_as_uint(mb.x) != __float_as_uint(mb0) || __float_as_uint(mb.y) != __float_as_uint(mb1)) ? 4u : 0u;
    bad |= (__float_as_uint(ac.x) != __float_as_uint(ac0) || __float_as_uint(ac.y) != __float_as_uint(ac1)) ? 8u : 0u;
    if (bad) {
      const unsigned k = atomicAdd(log, 1u);
      if ((int)k < max_log) {
        unsigned* o = log + 8 + 8 * (size_t)k;
        o[0] = t; o[1] = it; o[2] = threadIdx.x & 63; o[3] = bad; o[4] = __float_as_uint(m0); o[5] = __float_as_uint(m.x); o[6] = __float_as_uint(m.y);
        o[7] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
      }
    }
    acc += m.x + s.y + mb.y + ac.x;
    a.x = a.x * 1.0001f + 0.001f; a.y = a.y * 0.9999f + 0.002f; b.x += 0.0003f; b.y -= 0.0001f;
  }
  sink[t] = acc;
}
extern "C" int pk_probe_launch(void* log, int max_log, int iters, void* sink, long n_threads, void* stream) {
  hipLaunchKernelGGL(pk_probe_kernel, dim3((unsigned)(n_threads / 256)), dim3(256), 0, (hipStream_t)stream, (unsigned*)log, max_log, iters, (float*)sink);
  return (int)hipGetLastError();
}
