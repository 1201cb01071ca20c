// What is the issue rate of the fp32 FMA forms on gfx950?  (profiles/r04_dwconv_valu.txt)
// dwconv7x7+LN spends its time in a v_pk_fma_f32 stream that reaches a third of the 157.3 TFLOP/s vector peak even with its loads
// and weight reads removed.  This probe times register-only FMA streams, 64 independent accumulators per lane like the kernel:
//   fma      v_fma_f32      acc, a, b, acc           (1 FMA per lane and instruction)
//   pk       v_pk_fma_f32   acc2, a2, b2, acc2       (2 FMAs per lane and instruction)
//   pk_bcast v_pk_fma_f32 with the b operand the same register pair in every instruction (a weight held in registers)
//   pk_alt2  ... the b operand alternating between two pairs (the kernel's pattern: wv.xy, wv.zw, wv.xy, ...)
//   pk_run8  ... the b operand changing every 8 instructions, the a operand sliding by one pair per instruction (a kx tap of
//            dwconv: acc[j] += row[j + kx] * w for j = 0..7)
// at 1, 2 and 4 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 tools/valu_rate_probe.hip -o _ab/valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

using f2 = __attribute__((ext_vector_type(2))) float;
constexpr int kIters = 4096;

template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, float seed) {
  f2 acc[32];
  f2 a[8];
#pragma unroll
  for (int i = 0; i < 32; ++i) acc[i] = f2{seed * i, seed + i};
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = f2{1.0f + seed * i, 1.0f - seed * i};
  f2 b = f2{seed, -seed}, b2 = f2{-seed, seed * 2};
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      if constexpr (MODE == 0) {          // scalar fma, two per accumulator pair
        asm volatile("v_fma_f32 %0, %2, %3, %0\n\tv_fma_f32 %1, %4, %5, %1"
                     : "+v"(acc[i].x), "+v"(acc[i].y) : "v"(a[i & 7].x), "v"(a[(i + 1) & 7].x), "v"(a[i & 7].y), "v"(a[(i + 1) & 7].y));
      } else if constexpr (MODE == 1) {   // packed, all three operands distinct register pairs
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i & 7]), "v"(a[(i + 3) & 7]));
      } else if constexpr (MODE == 2) {   // packed, one operand the same pair throughout
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i & 7]), "v"(b));
      } else if constexpr (MODE == 3) {   // packed, the shared operand alternates between two pairs
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i & 7]), "v"(i & 1 ? b : b2));
      } else {                            // runs of 8 with one weight pair, the other operand sliding
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[(i + (i >> 3)) & 7]), "v"((i >> 3) & 1 ? b : b2));
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) s += acc[i].x + acc[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, float* d_out, int cus) {
  for (int wps : {1, 2, 4}) {            // waves per SIMD = workgroups of 256 threads (4 waves) per CU
    const int blocks = cus * wps;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, 1e-9f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, 1e-9f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 5.0 * blocks * 256.0 * kIters * 32 * 2 * 2;   // 32 accumulator pairs x 2 FMAs x 2 flops
    const double insts = 5.0 * blocks * 4.0 * kIters * (MODE == 0 ? 64 : 32);
    printf("%-9s %d wave(s)/SIMD: %7.1f TFLOP/s  (%.2f cycles per wave-instruction per SIMD at 2.4 GHz)\n", name, wps, flops / (ms * 1e-3) / 1e12,
           (ms * 1e-3) * 2.4e9 / (insts / (cus * 4.0)));
  }
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  float* d_out;
  hipMalloc(&d_out, sizeof(float) * cus * 4 * 256);
  printf("# %s, %d CUs, clock %d MHz\n", p.gcnArchName, cus, p.clockRate / 1000);
  run<0>("fma", d_out, cus);
  run<1>("pk", d_out, cus);
  run<2>("pk_bcast", d_out, cus);
  run<3>("pk_alt2", d_out, cus);
  run<4>("pk_run8", d_out, cus);
  return 0;
}
