// Depth-to-flow between two rendered / observed depth maps — the `flow_cuda` torch extension of the reference
// (core/csrc/flow/src/flow_cuda.cpp:30-47, kernel flow_cuda_kernel.cu:33-64), boundary row of SURVEY.md §8(b).
//
// One thread per source pixel, x fastest (coalesced 4-byte reads/writes; the gather from depth_tgt is the only
// irregular access).  HBM-bound: 4 B read + 12 B written per pixel plus one 4 B gather.
// Arithmetic is kept exactly as the reference's C++ (float, left to right, no FMA: the library is built with
// -ffp-contract=off; the comparisons against 1E-3 / 3E-3 and the + 1E-15 run in double), so flow and valid are
// bit-identical to the oracle and to the reference's CPU kernel for one image.  Kinv / KT are indexed per image.
#include "common.hpp"

namespace {

__global__ __launch_bounds__(256) void flow_kernel(const float* __restrict__ depth_src, const float* __restrict__ depth_tgt,
                                                   const float* __restrict__ KT, const float* __restrict__ Kinv,
                                                   float* __restrict__ flow, float* __restrict__ valid, int B, int height,
                                                   int width) {
  const long n = (long)B * height * width;
  for (long index = (long)blockIdx.x * blockDim.x + threadIdx.x; index < n; index += (long)gridDim.x * blockDim.x) {
    const int w = (int)(index % width);
    const int h = (int)((index / width) % height);
    const int bi = (int)(index / width / height);
    const float* ki = Kinv + 9 * bi;
    const float* kt = KT + 12 * bi;
    const float zs = depth_src[index];
    const float x = (w * ki[0] + h * ki[1] + ki[2]) * zs;
    const float y = (w * ki[3] + h * ki[4] + ki[5]) * zs;
    const float z = zs;
    float f0 = 0.f, f1 = 0.f, v = 0.f;
    if ((double)zs > 1E-3) {
      const float hx = x * kt[0] + y * kt[1] + z * kt[2] + kt[3];
      const float hy = x * kt[4] + y * kt[5] + z * kt[6] + kt[7];
      const float hz = (float)((double)(x * kt[8] + y * kt[9] + z * kt[10] + kt[11]) + 1E-15);
      const float u = hx / hz;
      const float v_ = hy / hz;
      const int ui = (int)roundf(u);
      const int vi = (int)roundf(v_);
      if (u >= 0 && u <= width - 1 && v_ >= 0 && v_ <= height - 1) {
        const float zt = depth_tgt[((long)bi * height + vi) * width + ui];
        if ((double)fabsf(hz - zt) < 3E-3) {
          f0 = v_ - h;
          f1 = u - w;
          v = 1.f;
        }
      }
    }
    flow[(((long)bi * 2 + 0) * height + h) * width + w] = f0;
    flow[(((long)bi * 2 + 1) * height + h) * width + w] = f1;
    valid[index] = v;
  }
}

}  // namespace

extern "C" int gdrnpp_flow_forward(const float* depth_src, const float* depth_tgt, const float* KT, const float* Kinv,
                                   float* flow, float* valid, int B, int H, int W, void* stream) {
  GDRNPP_REQUIRE(depth_src && depth_tgt && KT && Kinv && flow && valid, GDRNPP_EINVAL, "gdrnpp_flow_forward: null pointer");
  GDRNPP_REQUIRE(B > 0 && H > 0 && W > 0, GDRNPP_EINVAL, "gdrnpp_flow_forward: B=%d H=%d W=%d", B, H, W);
  const long n = (long)B * H * W;
  long blocks = (n + 255) / 256;
  if (blocks > 256 * 64) blocks = 256 * 64;
  hipLaunchKernelGGL(flow_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, depth_src, depth_tgt, KT, Kinv,
                     flow, valid, B, H, W);
  return gdrnpp::check_launch("gdrnpp_flow_forward");
}
