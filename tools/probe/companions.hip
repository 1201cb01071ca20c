// Diagnostic (not part of the product): kernels that loop over ONE instruction kind of the split GEMM, as companions of pk_probe_kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f2 = __attribute__((ext_vector_type(2))) float;
template <int KIND>
__global__ __launch_bounds__(256) void companion_kernel(float* __restrict__ sink, const float* __restrict__ src, int iters) {
  extern __shared__ uint4 smem[];
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  float x = 1.0f + (float)(t & 255) * 0.01f, y = 0.5f + (float)(threadIdx.x & 63) * 0.02f, acc = 0.f;
  if constexpr (KIND == 0) {                       // v_mfma_f32_32x32x16_f16
    f16x8 a, b; f32x16 c;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(x + i); b[i] = (_Float16)(y - i); }
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    for (int it = 0; it < iters; ++it) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    for (int i = 0; i < 16; ++i) acc += c[i];
  } else if constexpr (KIND == 1) {                // v_fma_mix_f32 (VOP3P, op_sel / op_sel_hi pick halves and widths)
    unsigned h = __builtin_bit_cast(unsigned, (__attribute__((ext_vector_type(2))) _Float16){(_Float16)x, (_Float16)y});
    for (int it = 0; it < iters; ++it) {
      asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,0]" : "+v"(acc) : "v"(h), "v"(h));
      asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,0]" : "+v"(acc) : "v"(h), "v"(h));
    }
  } else if constexpr (KIND == 2) {                // v_cvt_pk_f16_f32
    unsigned r = 0;
    for (int it = 0; it < iters; ++it) { unsigned q; asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(q) : "v"(x), "v"(y)); r ^= q; x += 0.001f; }
    acc = (float)r;
  } else if constexpr (KIND == 3) {                // v_dot2c_f32_f16
    unsigned h = __builtin_bit_cast(unsigned, (__attribute__((ext_vector_type(2))) _Float16){(_Float16)x, (_Float16)y});
    for (int it = 0; it < iters; ++it) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(acc) : "v"(h), "v"(h));
  } else if constexpr (KIND == 4) {                // global_load_lds_dwordx4
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    const float* p = src + (size_t)t * 4;
    for (int it = 0; it < iters; ++it) {
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(p + (size_t)(it & 15) * 1048576), "s"(__builtin_amdgcn_readfirstlane(lds0 + (threadIdx.x >> 6) * 1024u)) : "memory");
      if ((it & 7) == 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    acc = __builtin_bit_cast(float, smem[threadIdx.x].x);
  } else if constexpr (KIND == 5) {                // v_mov_b32_dpp (row_shr) + ds_read_b128 / ds_write_b32
    smem[threadIdx.x] = uint4{t, t, t, t};
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
      float q;
      asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "=v"(q) : "v"(x));
      const uint4 u = smem[(threadIdx.x + it) & 255];
      acc += q + __builtin_bit_cast(float, u.x & 0x3fffffffu);
    }
  } else if constexpr (KIND == 6) {                // v_accvgpr_write / read
    for (int it = 0; it < iters; ++it) {
      float q;
      asm volatile("v_accvgpr_write_b32 a0, %1\n\ts_nop 1\n\tv_accvgpr_read_b32 %0, a0" : "=v"(q) : "v"(x) : "a0");
      acc += q; x += 0.001f;
    }
  } else if constexpr (KIND == 7) {                // f64 VALU (GroupNorm statistics): v_add_f64 / v_mul_f64
    double d = x, e = y;
    for (int it = 0; it < iters; ++it) { d = d * 1.0000001 + e; }
    acc = (float)d;
  } else if constexpr (KIND == 8) {                // plain packed fp32 in the companion too
    f2 a = {x, y}, b = {y, x};
    for (int it = 0; it < iters; ++it) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(a) : "v"(b));
    acc = a.x + a.y;
  } else if constexpr (KIND >= 9) {               // other MFMA shapes
    using f16x4 = __attribute__((ext_vector_type(4))) _Float16;
    using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
    using f32x4 = __attribute__((ext_vector_type(4))) float;
    f16x8 a8, b8; f16x4 a4, b4; bf16x8 ab, bb; f32x16 c16; f32x4 c4;
    for (int i = 0; i < 8; ++i) { a8[i] = (_Float16)(x + i); b8[i] = (_Float16)(y - i); ab[i] = (__bf16)(x + i); bb[i] = (__bf16)(y - i); }
    for (int i = 0; i < 4; ++i) { a4[i] = (_Float16)(x + i); b4[i] = (_Float16)(y - i); c4[i] = 0.f; }
    for (int i = 0; i < 16; ++i) c16[i] = 0.f;
    long a64 = (long)t * 0x0101010101010101L, b64 = 0x3838383838383838L;
    for (int it = 0; it < iters; ++it) {
      if constexpr (KIND == 9) c4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c4, 0, 0, 0);
      if constexpr (KIND == 10) c16 = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, c16, 0, 0, 0);
      if constexpr (KIND == 11) c16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c16, 0, 0, 0);
      if constexpr (KIND == 12) c16 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, c16, 0, 0, 0);
      if constexpr (KIND == 13) c16 = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a64, b64, c16, 0, 0, 0);
      if constexpr (KIND == 14) c4 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c4, 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) acc += c16[i];
    for (int i = 0; i < 4; ++i) acc += c4[i];
  }
  sink[t] = acc;
}
extern "C" int companion_launch(int kind, void* sink, const void* src, int iters, int blocks, void* stream) {
  hipStream_t st = (hipStream_t)stream;
#define L(K) case K: hipLaunchKernelGGL(companion_kernel<K>, dim3(blocks), dim3(256), 8192, st, (float*)sink, (const float*)src, iters); break;
  switch (kind) { L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) L(10) L(11) L(12) L(13) L(14) default: return -1; }
  return (int)hipGetLastError();
}
