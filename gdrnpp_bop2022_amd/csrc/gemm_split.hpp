// Declarations shared by the split-GEMM kernels (gemm_split.hip: register-staged and LDS-DMA forms; gemm_split_pipe.hip:
// the software-pipelined LDS-DMA form).  See gemm_split.hip for the numerical scheme (exact 3-way bf16 operand split, six
// partial products on v_mfma_f32_32x32x16_bf16 with fp32 accumulation).
#pragma once
#include "common.hpp"

namespace gdrnpp {
namespace splitgemm {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using f32x4v = __attribute__((ext_vector_type(4))) float;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

constexpr int BM = 128, BN = 128, BK = 16, KB = BK / 8, PLANE = BM + 4;
constexpr int OPER_SLOTS = 3 * KB * PLANE;  // uint4 slots per operand image (register-staged kernel)
constexpr int W_TILE_SLOTS = 3 * KB * BN;   // uint4 slots of one packed 128x16 weight tile
enum { EPI_BIAS = 0, EPI_GELU = 1, EPI_SCALE_RES = 2 };

// Order of the six partial products inside one k-tile, identical in every split-GEMM kernel so that their results are
// bitwise equal: grouped by the weight split (l, m, m, h, h, h) so that a kernel needs at most two of the three weight
// fragment sets in registers at a time, small terms first inside a group.  (A split index, B split index): 0 = h, 1 = m, 2 = l.
#define GDRNPP_SPLIT_PRODUCT_ORDER        \
  constexpr int TA[6] = {0, 1, 0, 2, 1, 0}; \
  constexpr int TB[6] = {2, 1, 1, 0, 0, 0};

// CONV: A is an NHWC image [.,H,W,Cin] and the GEMM row m = output pixel, k = (tap, channel) of a KHxKW convolution with
// stride and symmetric zero padding (implicit im2col: the k-tile's 16 channels of one tap are 64 contiguous bytes per
// pixel).  Every row keeps a pointer to its anchor input pixel (oy*stride, ox*stride), which is always inside the image.
// The k-tiles of a convolution run 32-channel group outer, tap, then the group's two 16-channel chunks (one 128-byte line
// of a pixel; a single chunk per group when C/16 is odd); k-tile (group g, tap, half h) reads weight tile
// tap * (C/16) + 2g + h of the (tap, channel)-ordered packed weight.  The KH*KW taps of a group re-read the same lines
// back to back, so the tap overlap is served by L2 (tap-outer order puts a whole pass over the tile's channels, 16 MB
// per XCD at 64x64x256, between two reads of a pixel and fetched the image 12x).
// H, W, C: input image; OH, OW: output image; KW x (K / (KW*C)) taps, stride, zero padding `pad` on every side.
// nk_split > 0: split-K (linear only), blockIdx.y-th chunk of nk_split k-tiles -> partial C
struct ConvGeom { int H, W, C, OH, OW, KW, stride, pad; int nk_split; };

using gdrnpp::gelu_erf;  // common.hpp

// Software-pipelined LDS-DMA kernel (gemm_split_pipe.hip).  Handles the linear form and the 3x3/1/1 convolution with
// M*K*4 (resp. the image bytes) below 4 GiB; returns -1 when the problem is outside its domain (the caller then uses the
// kernels of gemm_split.hip), else the launch status.
int launch_split_pipe(const float* A, const uint4* Wp, const float* bias, const float* gamma, const float* resid, float* C,
                      int M, int N, int K, int epilogue, bool conv, ConvGeom cg, int a_stages, hipStream_t st, const char* what);

// The same kernel as a split-K launch (linear form): grid (256x128 tiles, K chunks of nk_split k-tiles), raw partial sums to
// partials[chunk][M][N]; -1 outside its domain.
int launch_split_pipe_splitk(const float* A, const uint4* Wp, float* partials, int M, int N, int K, int nk_split, int a_stages,
                             hipStream_t st, const char* what);

// 128-row form of the pipelined kernel (linear): fused epilogue (nk_split == 0) or split-K partials to C = partials[chunk][M][N]
int launch_split_pipe128(const float* A, const uint4* Wp, const float* bias, const float* gamma, const float* resid, float* C,
                         int M, int N, int K, int epilogue, int nk_split, hipStream_t st, const char* what);

}  // namespace splitgemm

int option_split_gemm_pipe();       // 0: off, 2 / 3 (default): pipelined kernel with that many A stages for 256-row tiles (linear form)
int option_split_gemm_panel();      // row blocks per tile panel of wide layers in the pipelined kernel (default 4; 0: row-major)
int option_split_gemm_big_tiles();  // 256x128 tiles (pipelined / LDS-DMA kernels) from this many of them on
int option_split_gemm_pipe_conv();  // 1: the 3x3/1/1 convolution uses it too (default 0: measured 1 % slower than the LDS-DMA kernel)

}  // namespace gdrnpp
