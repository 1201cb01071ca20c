// Device helpers shared by the three-product (fp16x2) GEMM kernels: gemm_split2_pipe.hip and gemm_mlp_fused.hip.
#pragma once
#include "gemm_split.hpp"

namespace gdrnpp {
namespace split2 {

using namespace gdrnpp::splitgemm;
using f16x2 = __attribute__((ext_vector_type(2))) _Float16;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

__device__ __forceinline__ void dma_s(unsigned voff, const void* sbase, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst)
               : "memory");
}
__device__ __forceinline__ void dma_v(const void* gsrc, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int I, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < E) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, E>(f);
  }
}

__device__ __forceinline__ unsigned cvt_pk_f16(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}
// x - f32(half HI of hpk), exact: one v_fma_mix_f32 (f16 source read in place, no v_cvt_f32_f16 in front of the subtraction)
template <int HI>
__device__ __forceinline__ float residual(float x, unsigned hpk) {
  float r;
  if constexpr (HI) asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hpk), "v"(x));
  else asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hpk), "v"(x));
  return r;
}

// s + (lo of hpk)^2 + (hi of hpk)^2: one v_dot2c_f32_f16 (row sums of squares of the range check)
__device__ __forceinline__ float sumsq2(unsigned hpk, float s) {
  const f16x2 v = __builtin_bit_cast(f16x2, hpk);
  return __builtin_amdgcn_fdot2(v, v, s, false);
}

// "f16x2 rows": an activation tensor f32[M][K] in which every aligned group of 8 consecutive elements of a row is replaced, in the
// same 32 bytes, by its 8 fp16 h halves followed by its 8 fp16 l halves — the split the k-loop of gemm_split2_pipe.hip otherwise
// repeats per k-tile AND per column tile (K = 512, N = 2048: sixteen times per element).  A producer whose lanes 2k / 2k + 1 hold
// values 0..3 / 4..7 of a group (float4 per lane, row-major) writes it with one quad exchange: each lane stores the returned 16
// bytes where its float4 would have gone.  Same cvt_pk / v_fma_mix / cvt_pk as HalfSplit2::step: bit-identical operands.
// Must be called by both lanes of a pair (DPP reads the neighbour's registers).
__device__ __forceinline__ uint4 f16x2_rows_quad(float x, float y, float z, float w, bool odd) {
  const unsigned h01 = cvt_pk_f16(x, y), h23 = cvt_pk_f16(z, w);
  const unsigned l01 = cvt_pk_f16(residual<0>(x, h01), residual<1>(y, h01)), l23 = cvt_pk_f16(residual<0>(z, h23), residual<1>(w, h23));
  const unsigned g0 = odd ? h01 : l01, g1 = odd ? h23 : l23;          // what the neighbour needs
  const unsigned r0 = (unsigned)__builtin_amdgcn_mov_dpp((int)g0, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]: lane ^ 1
  const unsigned r1 = (unsigned)__builtin_amdgcn_mov_dpp((int)g1, 0xB1, 0xF, 0xF, true);
  return odd ? make_uint4(r0, r1, l01, l23) : make_uint4(h01, h23, r0, r1);
}

// power-of-two scale of a weight tensor: max|w| * 2^e in [2^13, 2^14) (fp16: h = rn(x) cannot overflow, l of a typical weight
// stays normal); e clamped so that 2^e and 2^-e are normal fp32
__device__ __forceinline__ int weight_exp(unsigned amax_bits) {
  const int ex = (int)((amax_bits >> 23) & 0xffu) - 127;   // floor(log2 amax) for normal amax; zero / subnormal -> -127
  const int e = 13 - ex;
  return amax_bits == 0u ? 0 : max(-110, min(110, e));
}

// W f32[n] -> max |w| (bits of a non-negative float order like unsigned integers)
static __global__ void amax_kernel(const float* __restrict__ W, long n, unsigned* __restrict__ out) {
  float m = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float a = fabsf(W[i]);
    m = (a > m || a != a) ? a : m;   // a NaN weight poisons the maximum (and with it every product, as it must)
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float t = __shfl_xor(m, o, 64);
    m = (t > m || t != t) ? t : m;
  }
  if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

// one workgroup per weight row: a row that is not all zero whose SCALED rms is below 2^-4 (2^-17 of the tensor maximum: its low
// halves are fp16 subnormals), or that holds an inf / NaN, raises *flag — the per-row test of the kernels' A side, applied to the other operand
static __global__ void weight_rows_range_kernel(const float* __restrict__ W, const unsigned* __restrict__ amax_bits,
                                                unsigned* __restrict__ flag, int K) {
  const float sc = __builtin_ldexpf(1.f, weight_exp(*amax_bits));
  const float* row = W + (size_t)blockIdx.x * K;
  float s = 0.f;
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    const float v = row[k] * sc;
    s = fmaf(v, v, s);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    s = (part[0] + part[1]) + (part[2] + part[3]);
    if ((s > 0.f && s < (float)K * 0x1p-8f) || !(s < __builtin_inff())) atomicOr(flag, 1u);   // below range, or inf / NaN weights
  }
}

}  // namespace split2
}  // namespace gdrnpp
