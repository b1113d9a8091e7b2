// Test-only host harness around parsenet_codebase_amd/csrc/fit_math.h: runs the arithmetic of the
// batched primitive fits (moment table, dual-number fit with the lane = tangent convention, the
// cone's second pass, residual distances and the adjoint) for ONE segment on the CPU, exactly
// as the gfx950 kernels of fitbatch.hip sequence it, so that tests/test_fit_math_host.py can
// hold it against the oracle without a GPU.  Not part of the product library.
#include "../../parsenet_codebase_amd/csrc/fit_math.h"

#include <string.h>

extern "C" int fbh_segment(const float* P, const float* Nrm, const float* w, int n, int type, const float* gtP,
                           int ngt, int sqrt_flag, double* params /*16*/, double* moments /*64*/,
                           float* dist_out, double* gw /*n*/, double* jac_out /*16*64 or NULL*/) {
  double M[FB_NMOM];
  for (int m = 0; m < FB_NMOM; ++m) M[m] = 0.0;
  for (int j = 0; j < n; ++j) {
    const double z[7] = {1.0, P[3 * j], P[3 * j + 1], P[3 * j + 2], Nrm[3 * j], Nrm[3 * j + 1], Nrm[3 * j + 2]};
    const double ww = (double)w[j];
    const double wp[4] = {1.0, ww, ww * ww, ww * ww * ww};
    for (int m = 0; m < M_USED; ++m) {
      const FbMono mono = fb_table[m];
      M[m] += wp[mono.e] * z[mono.i1] * z[mono.i2] * z[mono.i3];
    }
  }
  memcpy(moments, M, sizeof(M));
  static double jac[FB_NPAR][FB_NMOM];
  int status = 0;
  double lamb = 0.0;
  for (int lane = 0; lane < FB_NMOM; ++lane) {
    Dd out[FB_NPAR];
    int st;
    fit_segment(M, lane, type, n, out, &st, &lamb);
    for (int k = 0; k < FB_NPAR; ++k) {
      jac[k][lane] = lane < M_USED ? out[k].d : 0.0;
      if (lane == 0) params[k] = out[k].v;
    }
    if (lane == 0) status = st;
  }
  params[15] = lamb;
  double cone_direct = 0.0;
  if (type == FB_CONE && !(status & 2)) {
    float c[3], a[3];
    for (int i = 0; i < 3; ++i) {
      c[i] = (float)params[i];
      a[i] = (float)params[3 + i];
    }
    double tot[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = 0; j < n; ++j) cone_point(P[3 * j], P[3 * j + 1], P[3 * j + 2], w[j], c, a, tot);
    const double ws = tot[7] + FB_EPS, raw = tot[0] / ws;
    const double lo = 1e-3, hi = 3.142 / 2 - 1e-3;
    const double mask = (raw >= lo && raw <= hi) ? 1.0 : 0.0;
    for (int l = 0; l < FB_NMOM; ++l) {
      double v = 0.0;
      for (int k = 0; k < 6; ++k) v += tot[1 + k] * jac[k][l];
      v = v / ws - (l == M_S1 ? tot[0] / (ws * ws) : 0.0);
      jac[6][l] = mask * v;
    }
    params[6] = fmin(fmax(raw, lo), hi);
    cone_direct = mask / ws;
  }
  float th[FB_NT];
  for (int k = 0; k < FB_NT; ++k) th[k] = (float)params[k];
  double acc[FB_NT + 1];
  for (int k = 0; k <= FB_NT; ++k) acc[k] = 0.0;
  for (int t = 0; t < ngt; ++t) {
    const Df d = residual_point(type, gtP[3 * t], gtP[3 * t + 1], gtP[3 * t + 2], th, sqrt_flag);
    if (d.v != d.v) status |= 4;
    acc[0] += d.v;
    for (int k = 0; k < FB_NT; ++k) acc[1 + k] += d.d[k];
  }
  double dparam[FB_NT];
  *dist_out = (float)(acc[0] / (ngt > 0 ? ngt : 1));
  for (int k = 0; k < FB_NT; ++k) dparam[k] = acc[1 + k] / (ngt > 0 ? ngt : 1);
  double gM[FB_NMOM];
  for (int l = 0; l < FB_NMOM; ++l) {
    double v = 0.0;
    for (int k = 0; k < FB_NT; ++k) v += dparam[k] * jac[k][l];
    gM[l] = v;   // g_dist = 1
  }
  for (int j = 0; j < n; ++j) {
    const double z[7] = {1.0, P[3 * j], P[3 * j + 1], P[3 * j + 2], Nrm[3 * j], Nrm[3 * j + 1], Nrm[3 * j + 2]};
    double g = wmom_bwd_point(gM, (double)w[j], z);
    if (type == FB_CONE && cone_direct != 0.0) {
      float c[3], a[3], ux, uy, uz, nu, t;
      for (int i = 0; i < 3; ++i) {
        c[i] = (float)params[i];
        a[i] = (float)params[3 + i];
      }
      const float f = cone_acos_term(P[3 * j], P[3 * j + 1], P[3 * j + 2], c, a, &ux, &uy, &uz, &nu, &t);
      g += dparam[6] * cone_direct * (double)f;
    }
    gw[j] = g;
  }
  if (jac_out) memcpy(jac_out, jac, sizeof(jac));
  return status;
}
