// Small dense helpers of the differentiable primitive fits (src/primitive_forward.py:708-843,
// src/fitting_utils.py:420-455): batched eigen-decomposition of symmetric 3x3 matrices in
// fp64.  The fits only need the right singular vectors of tall (n x 3) matrices; those are the
// eigenvectors of the 3x3 Gram matrix, which a single thread diagonalises with cyclic Jacobi
// rotations — no host round trip, no LAPACK call per segment.
#include "common.h"

__global__ void pn_sym3_eig_kernel(const double* __restrict__ G, int M, double* __restrict__ evals,
                                   double* __restrict__ evecs) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= M) return;
  double a[3][3], v[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      a[i][j] = 0.5 * (G[(size_t)t * 9 + i * 3 + j] + G[(size_t)t * 9 + j * 3 + i]);
      v[i][j] = i == j ? 1.0 : 0.0;
    }
  for (int sweep = 0; sweep < 32; ++sweep) {
    const double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
    const double diag = fabs(a[0][0]) + fabs(a[1][1]) + fabs(a[2][2]);
    if (off <= 1e-300 || off <= 1e-18 * diag) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        if (a[p][q] == 0.0) continue;
        const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
        const double tt = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(tt * tt + 1.0), s = tt * c;
        for (int k = 0; k < 3; ++k) {  // A <- A J
          const double akp = a[k][p], akq = a[k][q];
          a[k][p] = c * akp - s * akq;
          a[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; ++k) {  // A <- J^T A
          const double apk = a[p][k], aqk = a[q][k];
          a[p][k] = c * apk - s * aqk;
          a[q][k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; ++k) {  // V <- V J
          const double vkp = v[k][p], vkq = v[k][q];
          v[k][p] = c * vkp - s * vkq;
          v[k][q] = s * vkp + c * vkq;
        }
      }
  }
  // sort descending
  int ord[3] = {0, 1, 2};
  double w[3] = {a[0][0], a[1][1], a[2][2]};
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2 - i; ++j)
      if (w[ord[j]] < w[ord[j + 1]]) {
        const int tmp = ord[j];
        ord[j] = ord[j + 1];
        ord[j + 1] = tmp;
      }
  for (int c = 0; c < 3; ++c) {
    const int o = ord[c];
    evals[(size_t)t * 3 + c] = w[o];
    // sign convention: the component of largest magnitude is positive
    int big = 0;
    for (int k = 1; k < 3; ++k)
      if (fabs(v[k][o]) > fabs(v[big][o])) big = k;
    const double sg = v[big][o] < 0 ? -1.0 : 1.0;
    for (int k = 0; k < 3; ++k) evecs[(size_t)t * 9 + k * 3 + c] = sg * v[k][o];
  }
}

extern "C" int pn_sym3_eig_f64(const double* G, int M, double* evals, double* evecs, void* stream) {
  PN_CHECK_ARG(G && evals && evecs && M > 0, "pn_sym3_eig_f64: bad arguments");
  hipLaunchKernelGGL(pn_sym3_eig_kernel, dim3(pn_cdiv(M, 64)), dim3(64), 0, (hipStream_t)stream, G,
                     M, evals, evecs);
  PN_CHECK_LAUNCH();
  return PN_OK;
}
