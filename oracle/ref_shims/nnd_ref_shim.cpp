// ORACLE build shim: compiles the reference's nnd_cpu.cpp in place (path passed as NND_SRC)
// and exposes its nnd_forward / nnd_backward through a C ABI on raw pointers.
#include NND_SRC

extern "C" {
int ref_nnd_forward(const float* xyz1, const float* xyz2, float* dist1, float* dist2, int* idx1, int* idx2, int b,
                    int n, int m) {
  at::Tensor t1{(void*)xyz1, {b, n, 3, 0}}, t2{(void*)xyz2, {b, m, 3, 0}};
  at::Tensor d1{dist1, {b, n, 0, 0}}, d2{dist2, {b, m, 0, 0}}, i1{idx1, {b, n, 0, 0}}, i2{idx2, {b, m, 0, 0}};
  return nnd_forward(t1, t2, d1, d2, i1, i2);
}
int ref_nnd_backward(const float* xyz1, const float* xyz2, float* g1, float* g2, const float* gd1, const float* gd2,
                     const int* idx1, const int* idx2, int b, int n, int m) {
  at::Tensor t1{(void*)xyz1, {b, n, 3, 0}}, t2{(void*)xyz2, {b, m, 3, 0}};
  at::Tensor a{g1, {b, n, 3, 0}}, c{g2, {b, m, 3, 0}}, d{(void*)gd1, {b, n, 0, 0}}, e{(void*)gd2, {b, m, 0, 0}};
  at::Tensor i1{(void*)idx1, {b, n, 0, 0}}, i2{(void*)idx2, {b, m, 0, 0}};
  return nnd_backward(t1, t2, a, c, d, e, i1, i2);
}
}
