/*
 * ORACLE — test infrastructure only.
 * CPU restatement of the chamfer / NN-distance forward and backward:
 *   core/csrc/torch_nndistance/src/nnd_cpu.cpp:3-25 (nnsearch),
 *   :87-132 (nnd_backward == nnd_cuda_kernel.cu:164-183 without atomics).
 * The reference accumulates `double d = x2*x2+y2*y2+z2*z2` where the right-hand
 * side is evaluated in float (nnd_cpu.cpp:15), so the comparison is an fp32
 * comparison and the FIRST minimum wins.
 * Pinned against the reference source compiled unmodified
 * (oracle/_ref/nnd_ref*.so, tests/test_nnd.py::test_oracle_vs_reference).
 */
void oracle_nnsearch(int b, int n, int m, const float* xyz1, const float* xyz2, float* dist, int* idx) {
  for (int i = 0; i < b; i++)
    for (int j = 0; j < n; j++) {
      float x1 = xyz1[(i * n + j) * 3 + 0], y1 = xyz1[(i * n + j) * 3 + 1], z1 = xyz1[(i * n + j) * 3 + 2];
      double best = 0;
      int besti = 0;
      for (int k = 0; k < m; k++) {
        float x2 = xyz2[(i * m + k) * 3 + 0] - x1;
        float y2 = xyz2[(i * m + k) * 3 + 1] - y1;
        float z2 = xyz2[(i * m + k) * 3 + 2] - z1;
        float df = x2 * x2 + y2 * y2 + z2 * z2;
        double d = df;
        if (k == 0 || d < best) { best = d; besti = k; }
      }
      dist[i * n + j] = (float)best;
      idx[i * n + j] = besti;
    }
}

void oracle_nnd_forward(int b, int n, int m, const float* xyz1, const float* xyz2, float* dist1, float* dist2,
                        int* idx1, int* idx2) {
  oracle_nnsearch(b, n, m, xyz1, xyz2, dist1, idx1);
  oracle_nnsearch(b, m, n, xyz2, xyz1, dist2, idx2);
}

/* one direction of the gradient; grads must be pre-zeroed by the caller */
static void grad_dir(int b, int n, int m, const float* xyz1, const float* xyz2, const float* gd1, const int* idx1,
                     float* g1, float* g2) {
  for (int i = 0; i < b; i++)
    for (int j = 0; j < n; j++) {
      int j2 = idx1[i * n + j];
      float g = gd1[i * n + j] * 2;
      for (int c = 0; c < 3; c++) {
        float v = g * (xyz1[(i * n + j) * 3 + c] - xyz2[(i * m + j2) * 3 + c]);
        g1[(i * n + j) * 3 + c] += v;
        g2[(i * m + j2) * 3 + c] += -v;
      }
    }
}

void oracle_nnd_backward(int b, int n, int m, const float* xyz1, const float* xyz2, float* gradxyz1, float* gradxyz2,
                         const float* graddist1, const float* graddist2, const int* idx1, const int* idx2) {
  for (int i = 0; i < b * n * 3; i++) gradxyz1[i] = 0.f;
  for (int i = 0; i < b * m * 3; i++) gradxyz2[i] = 0.f;
  grad_dir(b, n, m, xyz1, xyz2, graddist1, idx1, gradxyz1, gradxyz2);
  grad_dir(b, m, n, xyz2, xyz1, graddist2, idx2, gradxyz2, gradxyz1);
}
