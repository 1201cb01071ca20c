/* ORACLE (test infrastructure, not shipped): CPU restatement of the reference's depth-to-flow kernel
 * core/csrc/flow/src/flow_cuda_kernel.cu:33-64 (== flow_cpu.cpp:5-47 for one image).
 *
 * Per pixel (h, w) of image b: back-project with Kinv and the source depth, transform/project with KT = K [R|t],
 * round to the nearest target pixel, accept when the projected depth agrees with the target depth within 3 mm;
 * flow = (v_ - h, u - w) in channels (0, 1), valid = 1, else zeros.
 * Semantics kept: float arithmetic left to right without FMA (build with -ffp-contract=off), `zs > 1E-3` and
 * `|dz| < 3E-3` compared in double, `+ 1E-15` added in double before the store to float, round-half-away (round()),
 * bounds tested on the UNROUNDED projection.  Kinv / KT are indexed per image (the CUDA kernel's behaviour with one
 * pixel per thread); the cumulative pointer bump of the CPU file for batch > 1 is a reference bug and not copied. */
#include <math.h>

void oracle_flow_forward(const float* depth_src, const float* depth_tgt, const float* KT, const float* Kinv, float* flow,
                         float* valid, int b, int height, int width) {
  for (int bi = 0; bi < b; ++bi) {
    const float* ki = Kinv + 9 * bi;
    const float* kt = KT + 12 * bi;
    for (int h = 0; h < height; ++h)
      for (int w = 0; w < width; ++w) {
        const int index = (bi * height + h) * width + w;
        const float zs = depth_src[index];
        const float x = (w * ki[0] + h * ki[1] + ki[2]) * zs;
        const float y = (w * ki[3] + h * ki[4] + ki[5]) * zs;
        const float z = zs;
        int ok = 0;
        if (zs > 1E-3) {
          const float hx = x * kt[0] + y * kt[1] + z * kt[2] + kt[3];
          const float hy = x * kt[4] + y * kt[5] + z * kt[6] + kt[7];
          const float hz = (float)(x * kt[8] + y * kt[9] + z * kt[10] + kt[11] + 1E-15);
          const float u = hx / hz;
          const float v_ = hy / hz;
          const int ui = (int)roundf(u);
          const int vi = (int)roundf(v_);
          if (u >= 0 && u <= width - 1 && v_ >= 0 && v_ <= height - 1) {
            const float zt = depth_tgt[(bi * height + vi) * width + ui];
            if (fabsf(hz - zt) < 3E-3) {
              flow[((bi * 2 + 0) * height + h) * width + w] = v_ - h;
              flow[((bi * 2 + 1) * height + h) * width + w] = u - w;
              valid[index] = 1.f;
              ok = 1;
            }
          }
        }
        if (!ok) {
          flow[((bi * 2 + 0) * height + h) * width + w] = 0.f;
          flow[((bi * 2 + 1) * height + h) * width + w] = 0.f;
          valid[index] = 0.f;
        }
      }
  }
}
