// libgdrnpp_hip.so — library-level entry points (version, last error).
#include "gemm_split.hpp"
#include <cstring>
#include <atomic>
#include <mutex>

namespace gdrnpp {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
void* shim_scratch(size_t bytes) {
  static thread_local void* block = nullptr;
  static thread_local size_t cap = 0;
  static thread_local int block_dev = -1;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) { set_error("hipGetDevice: %s", hipGetErrorString(e)); return nullptr; }
  if (block && block_dev == dev && cap >= bytes) return block;
  if (block) { (void)hipFree(block); block = nullptr; cap = 0; }
  const size_t want = bytes < ((size_t)1 << 20) ? ((size_t)1 << 20) : bytes + (bytes >> 2);
  e = hipMalloc(&block, want);
  if (e != hipSuccess) { block = nullptr; set_error("hipMalloc(%zu): %s", want, hipGetErrorString(e)); return nullptr; }
  cap = want; block_dev = dev;
  return block;
}
int ensure_dynamic_lds(const void* kernel, int bytes) {
  struct Entry { const void* fn; int dev, bytes; };
  static std::mutex mu;
  static Entry table[128];
  static int n = 0;
  int dev = 0;
  GDRNPP_HIP_TRY(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  Entry* e = nullptr;
  for (int i = 0; i < n; ++i)
    if (table[i].fn == kernel && table[i].dev == dev) e = &table[i];
  if (e && e->bytes >= bytes) return 0;
  GDRNPP_HIP_TRY(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  if (e) e->bytes = bytes;
  else if (n < 128) table[n++] = Entry{kernel, dev, bytes};
  return 0;
}
// tuning switches: written by gdrnpp_set_option, read by launches on any host thread
static std::atomic<int> g_opt_glds{1}, g_opt_mi4{-1}, g_opt_pipe{3}, g_opt_pipe_conv{0}, g_opt_panel{4}, g_opt_big_tiles{256}, g_opt_dw_tile{-1}, g_opt_splitk_small{1}, g_opt_split2_wide{-1}, g_opt_mlp_fused_pipe{1}, g_opt_dw_lds_w{1}, g_opt_dw_lds_pad{0};
int option_split_gemm_glds() { return g_opt_glds; }
int option_split_gemm_pipe() { return g_opt_pipe; }
int option_split_gemm_pipe_conv() { return g_opt_pipe_conv; }
int option_split_gemm_mi4() { return g_opt_mi4; }
int option_split_gemm_panel() { return g_opt_panel; }
int option_split_gemm_big_tiles() { return g_opt_big_tiles; }
int option_dwconv_tile() { return g_opt_dw_tile; }
int option_splitk_small_tiles() { return g_opt_splitk_small; }
int option_split2_wide() { return g_opt_split2_wide; }
int option_mlp_fused_pipe() { return g_opt_mlp_fused_pipe; }
int option_dwconv_lds_w() { return g_opt_dw_lds_w; }
int option_dwconv_lds_pad() { return g_opt_dw_lds_pad; }
}  // namespace gdrnpp

namespace {
// PMC calibration streams (tools/pmc_traffic.py): read n floats with 4-byte or 16-byte lanes, fold, write one
// float per workgroup — a known byte count in the two access widths the path's kernels use.
template <int WIDTH>
__global__ void stream_read_kernel(const float* __restrict__ p, size_t n, float* __restrict__ out) {
  float acc = 0.f;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  if (WIDTH == 4) {
    const float4* q = reinterpret_cast<const float4*>(p);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n / 4; i += stride) {
      const float4 v = q[i];
      acc += (v.x + v.y) + (v.z + v.w);
    }
  } else {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) acc += p[i];
  }
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd(out + blockIdx.x, acc);
}
// stream-overlap probe: one wave spins on the constant-rate wall clock (s_memrealtime) until `ticks` have passed
__global__ void spin_kernel(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
}  // namespace

extern "C" {
int gdrnpp_debug_spin(int micros, void* stream) {
  GDRNPP_REQUIRE(micros >= 1 && micros <= 1000000, GDRNPP_EINVAL, "gdrnpp_debug_spin: micros must be in [1, 1000000]");
  int dev = 0, khz = 0;
  GDRNPP_HIP_TRY(hipGetDevice(&dev));
  GDRNPP_HIP_TRY(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev));
  if (khz <= 0) khz = 100000;      // 100 MHz: the constant-rate counter of every CDNA part
  hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long)micros * khz / 1000);
  return gdrnpp::check_launch("gdrnpp_debug_spin");
}
int gdrnpp_debug_stream_read(const float* p, size_t n, int lane_bytes, float* out_blocks, int blocks, void* stream) {
  GDRNPP_REQUIRE(p && out_blocks && n > 0 && blocks > 0 && (lane_bytes == 4 || lane_bytes == 16), GDRNPP_EINVAL,
                 "gdrnpp_debug_stream_read: bad arguments");
  if (lane_bytes == 16)
    hipLaunchKernelGGL(stream_read_kernel<4>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, n, out_blocks);
  else
    hipLaunchKernelGGL(stream_read_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, n, out_blocks);
  return gdrnpp::check_launch("gdrnpp_debug_stream_read");
}
int gdrnpp_copy_d2d(void* dst, const void* src, size_t bytes, void* stream) {
  if (bytes == 0) return 0;
  GDRNPP_REQUIRE(dst && src, GDRNPP_EINVAL, "gdrnpp_copy_d2d: null pointer");
  GDRNPP_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return 0;
}
int gdrnpp_set_option(const char* name, int value) {
  GDRNPP_REQUIRE(name, GDRNPP_EINVAL, "gdrnpp_set_option: null name");
  if (!strcmp(name, "split_gemm_glds")) { gdrnpp::g_opt_glds = value != 0; return 0; }
  if (!strcmp(name, "split_gemm_pipe_conv")) { gdrnpp::g_opt_pipe_conv = value != 0; return 0; }
  if (!strcmp(name, "split_gemm_pipe")) { gdrnpp::g_opt_pipe = value == 3 ? 3 : (value != 0 ? 2 : 0); return 0; }
  if (!strcmp(name, "split_gemm_panel")) { gdrnpp::g_opt_panel = value < 2 ? 0 : (value > 64 ? 64 : value); return 0; }
  if (!strcmp(name, "split_gemm_big_tiles")) { gdrnpp::g_opt_big_tiles = value < 1 ? 1 : value; return 0; }
  if (!strcmp(name, "splitk_small_tiles")) { gdrnpp::g_opt_splitk_small = value != 0; return 0; }
  if (!strcmp(name, "dwconv_tile")) { gdrnpp::g_opt_dw_tile = (value >= 0 && value <= 2) ? value : -1; return 0; }
  if (!strcmp(name, "split2_wide")) { gdrnpp::g_opt_split2_wide = value < 0 ? -1 : (value != 0); return 0; }
  if (!strcmp(name, "mlp_fused_pipe")) { gdrnpp::g_opt_mlp_fused_pipe = value != 0; return 0; }
  if (!strcmp(name, "dwconv_lds_w")) { gdrnpp::g_opt_dw_lds_w = value != 0; return 0; }
  if (!strcmp(name, "dwconv_lds_pad")) { gdrnpp::g_opt_dw_lds_pad = value < 0 ? 0 : (value > 112 * 1024 ? 112 * 1024 : value); return 0; }
  if (!strcmp(name, "split_gemm_mi4")) { gdrnpp::g_opt_mi4 = value < 0 ? -1 : (value != 0); return 0; }
  gdrnpp::set_error("gdrnpp_set_option: unknown option '%s'", name);
  return GDRNPP_EINVAL;
}
int gdrnpp_version(void) { return 110; /* 0.1.1 */ }
const char* gdrnpp_last_error(void) { return gdrnpp::g_err; }
}
