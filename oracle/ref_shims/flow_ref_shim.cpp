// ORACLE build shim: compiles the reference's flow_cpu.cpp in place (path passed as FLOW_SRC) and exposes its
// flow_kernel<float> through a C ABI on raw pointers.  NOTE the reference advances Kinv / KT cumulatively inside
// the pixel loop (flow_cpu.cpp:16,22), so it is only meaningful for batch size 1.
#include FLOW_SRC

extern "C" void ref_flow_forward(const float* depth_src, const float* depth_tgt, const float* KT, const float* Kinv,
                                 float* flow, float* valid, int b, int h, int w) {
  flow_kernel<float>(b * h * w, depth_src, depth_tgt, h, w, KT, Kinv, flow, valid);
}
