/*
 * ORACLE — test infrastructure only.
 * CPU restatement of the ROI crop of read_data_test (core/gdrn_modeling/datasets/data_loader.py:754-797):
 *   crop_resize_by_warp_affine  core/utils/data_utils.py:115-133
 *   get_affine_transform        core/utils/data_utils.py:136-184 (rot = 0, shift = 0), get_3rd_point :193-195
 *   cv2.getAffineTransform + cv2.warpAffine(flags=INTER_LINEAR | INTER_NEAREST, BORDER_CONSTANT 0)
 *
 * OpenCV is a third-party dependency that is neither in /root/reference nor installable here (version unpinned:
 * it arrives through mmcv-full, requirements/requirements.txt:22).  Its published algorithm (modules/imgproc/src/
 * imgwarp.cpp, 3.4/4.x series) is restated:
 *   getAffineTransform : 6x6 system solved by LU with partial pivoting in double (cv::solve default DECOMP_LU)
 *   warpAffine         : M inverted in double; per column adelta/bdelta = cvRound(M[0|3]*x*1024); per row
 *                        X0 = cvRound((M[1]*y+M[2])*1024) + round_delta; fixed-point source coordinate with
 *                        5 fractional bits (INTER_BITS), AB_BITS = 10
 *   remap, 8U bilinear : integer weights tab*32768 (entry 0 is {32767,0,0,1} after the table's sum fix-up), value =
 *                        (sum + 2^14) >> 15; out-of-image taps read the border value 0
 *   remap, 32F bilinear: float weights (1-fy)(1-fx)…, left-to-right float sum; nearest: plain fetch, border 0
 * PARITY UNPINNED: cv2 cannot be imported to check these statements; the tests pin only
 * self-consistency properties (identity warps, integer translations, interior bilinear against closed form).
 * cvRound = round-half-to-even (lrint).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define AB_BITS 10
#define AB_SCALE (1 << AB_BITS)
#define INTER_BITS 5
#define INTER_TAB_SIZE (1 << INTER_BITS)
#define COEF_BITS 15
#define COEF_SCALE (1 << COEF_BITS)

static int cv_round(double v) { return (int)lrint(v); }
static short sat_short(int v) { return (short)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }

/* cv::LUImpl on a 6x6 system, one right-hand side */
static int lu_solve6(double* A, double* b) {
  const int m = 6;
  for (int i = 0; i < m; i++) {
    int k = i;
    for (int j = i + 1; j < m; j++)
      if (fabs(A[j * m + i]) > fabs(A[k * m + i])) k = j;
    if (fabs(A[k * m + i]) < 2.220446049250313e-16 * 100) return 0; /* DBL_EPSILON*100 */
    if (k != i) {
      for (int j = i; j < m; j++) { double t = A[i * m + j]; A[i * m + j] = A[k * m + j]; A[k * m + j] = t; }
      double t = b[i]; b[i] = b[k]; b[k] = t;
    }
    double d = -1 / A[i * m + i];
    for (int j = i + 1; j < m; j++) {
      double alpha = A[j * m + i] * d;
      for (int kk = i + 1; kk < m; kk++) A[j * m + kk] += alpha * A[i * m + kk];
      b[j] += alpha * b[i];
    }
  }
  for (int i = m - 1; i >= 0; i--) {
    double s = b[i];
    for (int k = i + 1; k < m; k++) s -= A[i * m + k] * b[k];
    b[i] = s / A[i * m + i];
  }
  return 1;
}

/* get_affine_transform(center, scale, rot=0, output_size) -> cv2.getAffineTransform(src, dst) as double[6] */
void oracle_get_affine_transform(double cx, double cy, double scale, int out_w, int out_h, double* M) {
  float src[3][2], dst[3][2];
  double src_dir1 = 0.0 * 0.0 + (scale * -0.5) * 1.0; /* get_dir([0, src_w*-0.5], 0)[1] */
  double src_dir0 = 0.0 * 1.0 - (scale * -0.5) * 0.0;
  float dst_dir[2] = {0.f, (float)(out_w * -0.5)};
  src[0][0] = (float)(cx + scale * 0.0); /* center + scale_tmp * shift, shift = 0 */
  src[0][1] = (float)(cy + scale * 0.0);
  src[1][0] = (float)(cx + src_dir0 + scale * 0.0);
  src[1][1] = (float)(cy + src_dir1 + scale * 0.0);
  dst[0][0] = (float)(out_w * 0.5);
  dst[0][1] = (float)(out_h * 0.5);
  dst[1][0] = (float)(out_w * 0.5) + dst_dir[0];
  dst[1][1] = (float)(out_h * 0.5) + dst_dir[1];
  /* get_3rd_point(a, b) = b + (-(a-b).y, (a-b).x) in float32 */
  {
    float dx = src[0][0] - src[1][0], dy = src[0][1] - src[1][1];
    src[2][0] = src[1][0] + (-dy);
    src[2][1] = src[1][1] + dx;
    dx = dst[0][0] - dst[1][0]; dy = dst[0][1] - dst[1][1];
    dst[2][0] = dst[1][0] + (-dy);
    dst[2][1] = dst[1][1] + dx;
  }
  double a[36], b[6];
  for (int i = 0; i < 3; i++) {
    int j = i * 12, k = i * 12 + 6;
    a[j] = a[k + 3] = src[i][0];
    a[j + 1] = a[k + 4] = src[i][1];
    a[j + 2] = a[k + 5] = 1;
    a[j + 3] = a[j + 4] = a[j + 5] = 0;
    a[k] = a[k + 1] = a[k + 2] = 0;
    b[i * 2] = dst[i][0];
    b[i * 2 + 1] = dst[i][1];
  }
  if (!lu_solve6(a, b)) memset(b, 0, sizeof(double) * 6);
  memcpy(M, b, sizeof(double) * 6);
}

static void invert_affine(const double* Min, double* M) {
  memcpy(M, Min, sizeof(double) * 6);
  double D = M[0] * M[4] - M[1] * M[3];
  D = D != 0 ? 1. / D : 0;
  double A11 = M[4] * D, A22 = M[0] * D;
  M[0] = A11; M[1] *= -D;
  M[3] *= -D; M[4] = A22;
  double b1 = -M[0] * M[2] - M[1] * M[5];
  double b2 = -M[3] * M[2] - M[4] * M[5];
  M[2] = b1; M[5] = b2;
}

static void bilinear_itab(int alpha, int* w) {
  int fy = alpha >> INTER_BITS, fx = alpha & (INTER_TAB_SIZE - 1);
  if (alpha == 0) { w[0] = 32767; w[1] = 0; w[2] = 0; w[3] = 1; return; } /* saturate_cast<short>(32768) + sum fix-up */
  w[0] = (32 - fy) * (32 - fx) * 32; w[1] = (32 - fy) * fx * 32; w[2] = fy * (32 - fx) * 32; w[3] = fy * fx * 32;
}
static void bilinear_ftab(int alpha, float* w) {
  int fy = alpha >> INTER_BITS, fx = alpha & (INTER_TAB_SIZE - 1);
  float sc = 1.f / INTER_TAB_SIZE;
  float vy0 = 1.f - fy * sc, vy1 = fy * sc, vx0 = 1.f - fx * sc, vx1 = fx * sc;
  w[0] = vy0 * vx0; w[1] = vy0 * vx1; w[2] = vy1 * vx0; w[3] = vy1 * vx1;
}

/* source coordinate of destination pixel (x, y): integer part (sx, sy) and 10-bit alpha */
static void src_coord(const double* M, int x, int y, int nearest, int* sx, int* sy, int* alpha) {
  int adelta = cv_round(M[0] * x * AB_SCALE), bdelta = cv_round(M[3] * x * AB_SCALE);
  int round_delta = nearest ? AB_SCALE / 2 : AB_SCALE / INTER_TAB_SIZE / 2;
  int X0 = cv_round((M[1] * y + M[2]) * AB_SCALE) + round_delta;
  int Y0 = cv_round((M[4] * y + M[5]) * AB_SCALE) + round_delta;
  if (nearest) {
    *sx = sat_short((X0 + adelta) >> AB_BITS);
    *sy = sat_short((Y0 + bdelta) >> AB_BITS);
    *alpha = 0;
  } else {
    int X = (X0 + adelta) >> (AB_BITS - INTER_BITS), Y = (Y0 + bdelta) >> (AB_BITS - INTER_BITS);
    *sx = sat_short(X >> INTER_BITS);
    *sy = sat_short(Y >> INTER_BITS);
    *alpha = (Y & (INTER_TAB_SIZE - 1)) * INTER_TAB_SIZE + (X & (INTER_TAB_SIZE - 1));
  }
}

/* cv2.warpAffine(src u8[H,W,cn], M, (ow,oh), INTER_LINEAR) */
void oracle_warp_affine_u8(const unsigned char* src, int H, int W, int cn, const double* M0, unsigned char* dst,
                           int ow, int oh) {
  double M[6];
  invert_affine(M0, M);
  for (int y = 0; y < oh; ++y)
    for (int x = 0; x < ow; ++x) {
      int sx, sy, alpha, w[4];
      src_coord(M, x, y, 0, &sx, &sy, &alpha);
      bilinear_itab(alpha, w);
      for (int k = 0; k < cn; ++k) {
        int v[4];
        for (int t = 0; t < 4; ++t) {
          int xx = sx + (t & 1), yy = sy + (t >> 1);
          v[t] = (xx >= 0 && xx < W && yy >= 0 && yy < H) ? src[((size_t)yy * W + xx) * cn + k] : 0;
        }
        int s = (v[0] * w[0] + v[1] * w[1] + v[2] * w[2] + v[3] * w[3] + (1 << (COEF_BITS - 1))) >> COEF_BITS;
        dst[((size_t)y * ow + x) * cn + k] = (unsigned char)(s < 0 ? 0 : (s > 255 ? 255 : s));
      }
    }
}

/* cv2.warpAffine(src f32[H,W,cn], M, (ow,oh), INTER_LINEAR | INTER_NEAREST) */
void oracle_warp_affine_f32(const float* src, int H, int W, int cn, const double* M0, float* dst, int ow, int oh,
                            int nearest) {
  double M[6];
  invert_affine(M0, M);
  for (int y = 0; y < oh; ++y)
    for (int x = 0; x < ow; ++x) {
      int sx, sy, alpha;
      src_coord(M, x, y, nearest, &sx, &sy, &alpha);
      if (nearest) {
        for (int k = 0; k < cn; ++k)
          dst[((size_t)y * ow + x) * cn + k] =
              (sx >= 0 && sx < W && sy >= 0 && sy < H) ? src[((size_t)sy * W + sx) * cn + k] : 0.f;
        continue;
      }
      float w[4];
      bilinear_ftab(alpha, w);
      for (int k = 0; k < cn; ++k) {
        float v[4];
        for (int t = 0; t < 4; ++t) {
          int xx = sx + (t & 1), yy = sy + (t >> 1);
          v[t] = (xx >= 0 && xx < W && yy >= 0 && yy < H) ? src[((size_t)yy * W + xx) * cn + k] : 0.f;
        }
        dst[((size_t)y * ow + x) * cn + k] = v[0] * w[0] + v[1] * w[1] + v[2] * w[2] + v[3] * w[3];
      }
    }
}
