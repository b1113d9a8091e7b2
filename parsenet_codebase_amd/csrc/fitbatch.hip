// Batched differentiable primitive fits for gfx950: every analytic segment (plane, sphere,
// cylinder, cone) of every shape of a step in FIVE launches, forward and backward, instead of
// the reference's serial Python loop of ~250 tensor operations per segment
//   src/primitive_forward.py:708-843 (Fit.fit_*_torch), :925-1047 (fit_one_shape_torch),
//   src/fitting_utils.py:32-85 (LeastSquares.lstsq, best_lambda), :385-455 (CustomSVD),
//   src/primitives.py:58-206 (ComputePrimitiveDistance), src/residual_utils.py:154-208.
//
// In the training path the points and normals carry no gradient; the only differentiable
// input of a fit is the segment's membership column w.  Every fit is a small closed-form
// function of WEIGHTED MOMENTS of the shape's sub-sampled points
//     M_m = sum_i w_i^e(m) * phi_m(p_i, n_i),      phi_m a monomial of degree <= 3,
// so the stage splits into
//   1. pn_weighted_moments_f64   M (60 moments per segment) — the only O(n) pass of the fit,
//                                linear in w, w^2, w^3: its adjoint is one more pass;
//   2. pn_primitive_fit_f64      moments -> parameters in fp64: 3x3 Jacobi eigenvectors,
//                                normal equations, rank test and ridge search on the device
//                                (no host synchronisation).  ONE WAVE PER SEGMENT: all lanes
//                                run the same arithmetic on dual numbers, lane l carrying the
//                                tangent with respect to moment l — the wave leaves the full
//                                Jacobian d(params)/d(moments) behind.  The eigenvector tangent
//                                is the transpose of the reference's CustomSVD backward
//                                (guarded 1/(s_i - s_j)), so reverse mode through the stored
//                                Jacobian reproduces the reference's gradient;
//   3. pn_cone_angle_f64         the cone's half angle needs a second pass over the points
//                                once apex and axis are known;
//   4. pn_primitive_residual_f32 mean squared distance of the ground-truth points of each
//                                segment to its primitive, with d(distance)/d(params) from
//                                fp32 dual numbers in the same pass;
//   5. pn_weighted_moments_bwd_f32  d(loss)/dw for all segments.
// MFMA is deliberately not used: these are 3x3 problems.
#include "common.h"

#include "fit_math.h"

// ---- stage 1: weighted moments -------------------------------------------------------
// grid (S, FB_CH), block 256 = 4 point groups x 64 moments.  Points of the shape are taken with
// stride `stride` (fit_one_shape_torch keeps every 2nd point, every 4th for analytic
// primitives); the weight of point j is W[shape, row, stride * j] + eps.
__global__ __launch_bounds__(256) void pn_wmom_fwd_kernel(
    const float* __restrict__ P, const float* __restrict__ Nrm, const float* __restrict__ W, int N,
    int Cp, int stride, float eps, const int* __restrict__ seg_shape, const int* __restrict__ seg_row,
    double* __restrict__ partial) {
  __shared__ double zt[FB_TILE][8];   // [1, p, n, pad]
  __shared__ double wp[FB_TILE][4];   // [1, w, w^2, w^3]
  __shared__ double red[4][FB_NMOM];
  const int s = blockIdx.x, ch = blockIdx.y;
  const int b = seg_shape[s], row = seg_row[s];
  const int n = (N + stride - 1) / stride;
  const int per = (n + FB_CH - 1) / FB_CH;
  const int j_begin = ch * per, j_end = min(n, j_begin + per);
  const float* Pb = P + (size_t)b * N * 3;
  const float* Nb = Nrm + (size_t)b * N * 3;
  const float* Wr = W + ((size_t)b * Cp + row) * N;
  const int m = threadIdx.x & 63, g = threadIdx.x >> 6;
  const FbMono mono = fb_table[m];
  double acc = 0.0;
  for (int j0 = j_begin; j0 < j_end; j0 += FB_TILE) {
    const int cnt = min(FB_TILE, j_end - j0);
    __syncthreads();
    if (threadIdx.x < cnt) {
      const int i = (j0 + threadIdx.x) * stride;
      const double w = (double)(Wr[i] + eps);   // the reference adds EPS in fp32
      zt[threadIdx.x][0] = 1.0;
      zt[threadIdx.x][1] = Pb[3 * i + 0];
      zt[threadIdx.x][2] = Pb[3 * i + 1];
      zt[threadIdx.x][3] = Pb[3 * i + 2];
      zt[threadIdx.x][4] = Nb[3 * i + 0];
      zt[threadIdx.x][5] = Nb[3 * i + 1];
      zt[threadIdx.x][6] = Nb[3 * i + 2];
      wp[threadIdx.x][0] = 1.0;
      wp[threadIdx.x][1] = w;
      wp[threadIdx.x][2] = w * w;
      wp[threadIdx.x][3] = w * w * w;
    }
    __syncthreads();
    for (int t = g; t < cnt; t += 4)
      acc += wp[t][mono.e] * zt[t][mono.i1] * zt[t][mono.i2] * zt[t][mono.i3];
  }
  red[g][m] = acc;
  __syncthreads();
  if (g == 0) {
    const double v = (red[0][m] + red[1][m]) + (red[2][m] + red[3][m]);
    partial[((size_t)s * FB_CH + ch) * FB_NMOM + m] = m < M_USED ? v : 0.0;
  }
}

// ---- stage 2: moments -> parameters + Jacobian -------------------------------------------
// One wave per segment.  params[s][0..15]; jac[s][k][l] = d params_k / d moment_l.
//   plane    : a(3), d
//   sphere   : centre(3), r
//   cylinder : axis(3), centre(3), r
//   cone     : apex(3), axis(3), [theta: stage 3]
// status bit 0: lstsq failed (non-finite / no full-rank ridge system); bit 1: null cone
// (condition number of w n above 1e5, the reference returns a zero cone without gradient).
__global__ __launch_bounds__(64) void pn_primfit_kernel(const double* __restrict__ partial,
                                                        const int* __restrict__ seg_type,
                                                        const int* __restrict__ seg_rows, int S,
                                                        double* __restrict__ params, double* __restrict__ jac,
                                                        int* __restrict__ status) {
  const int s = blockIdx.x;
  const int lane = threadIdx.x;
  __shared__ double msum[FB_NMOM];
  {
    double v = 0.0;
    for (int ch = 0; ch < FB_CH; ++ch) v += partial[((size_t)s * FB_CH + ch) * FB_NMOM + lane];
    msum[lane] = v;
  }
  __syncthreads();
  Dd out[FB_NPAR];
  int st = 0;
  double lamb = 0.0;
  fit_segment(msum, lane, seg_type[s], seg_rows[s], out, &st, &lamb);
  for (int k = 0; k < FB_NPAR; ++k) {
    jac[((size_t)s * FB_NPAR + k) * FB_NMOM + lane] = lane < M_USED ? out[k].d : 0.0;
    if (lane == 0) params[(size_t)s * FB_NPAR + k] = out[k].v;
  }
  if (lane == 0) {
    params[(size_t)s * FB_NPAR + 15] = lamb;   // diagnostic: ridge parameter used (0 = full rank)
    status[s] = st;
  }
}

// ---- stage 3: cone half angle ----------------------------------------------------------
// theta = clamp( sum_i w_i acos(min(|normalize(p_i - c) . a|, 0.999)) / (sum w + eps),
//                1e-3, 3.142/2 - 1e-3 )                           (primitive_forward.py:834-841)
// One block per segment (cones only do work).  Writes params[s][6] = theta, the Jacobian row of
// theta with respect to the moments (through apex, axis and sum w), and cone_direct[s] =
// [clamp mask] / (sum w + eps): the factor of the direct path d theta / d w_i = cone_direct * acos_i
// that the backward pass re-evaluates per point.
__global__ __launch_bounds__(256) void pn_cone_angle_kernel(
    const float* __restrict__ P, const float* __restrict__ W, int N, int Cp, int stride, float eps,
    const int* __restrict__ seg_shape, const int* __restrict__ seg_row, const int* __restrict__ seg_type,
    const int* __restrict__ status, double* __restrict__ params, double* __restrict__ jac,
    double* __restrict__ cone_direct) {
  const int s = blockIdx.x;
  if (seg_type[s] != FB_CONE) {
    if (threadIdx.x == 0) cone_direct[s] = 0.0;
    return;
  }
  if (status[s] & 2) {   // null cone: theta = 0, no gradient
    if (threadIdx.x == 0) cone_direct[s] = 0.0;
    return;
  }
  const int b = seg_shape[s], row = seg_row[s];
  const float* Pb = P + (size_t)b * N * 3;
  const float* Wr = W + ((size_t)b * Cp + row) * N;
  float c[3], a[3];
  for (int i = 0; i < 3; ++i) {
    c[i] = (float)params[(size_t)s * FB_NPAR + i];
    a[i] = (float)params[(size_t)s * FB_NPAR + 3 + i];
  }
  const int n = (N + stride - 1) / stride;
  // sums: [0] sum w acos, [1..3] d/dc, [4..6] d/da, [7] sum w
  double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int j = threadIdx.x; j < n; j += 256) {
    const int i = j * stride;
    const float w = Wr[i] + eps;
    cone_point(Pb[3 * i], Pb[3 * i + 1], Pb[3 * i + 2], w, c, a, acc);
  }
  __shared__ double red[4][8];
  __shared__ double tot[8];
  for (int k = 0; k < 8; ++k) {
    const double v = pn_wave_sum_d(acc[k]);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < 8) tot[threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) +
                                          (red[2][threadIdx.x] + red[3][threadIdx.x]);
  __syncthreads();
  const double ws = tot[7] + FB_EPS;
  const double raw = tot[0] / ws;
  const double lo = 1e-3, hi = 3.142 / 2 - 1e-3;
  const double mask = (raw >= lo && raw <= hi) ? 1.0 : 0.0;
  if (threadIdx.x < FB_NMOM) {
    const int l = threadIdx.x;
    double v = 0.0;
    for (int k = 0; k < 6; ++k) v += tot[1 + k] * jac[((size_t)s * FB_NPAR + k) * FB_NMOM + l];
    v = v / ws - (l == M_S1 ? tot[0] / (ws * ws) : 0.0);
    jac[((size_t)s * FB_NPAR + 6) * FB_NMOM + l] = mask * v;
  }
  if (threadIdx.x == 0) {
    params[(size_t)s * FB_NPAR + 6] = fmin(fmax(raw, lo), hi);
    cone_direct[s] = mask / ws;
  }
}

// ---- stage 4: residual distances (src/primitives.py:58-206) --------------------------------
// one block per segment; the ground-truth points of segment s are
// P[shape, gt_idx[gt_off[s] .. gt_off[s+1])].  dist[s] = mean distance, dparam[s][k] = d dist / d params_k.
// status bit 2: NaN distance (the reference raises in distance_from_cylinder).
__global__ __launch_bounds__(256) void pn_prim_residual_kernel(
    const float* __restrict__ P, int N, const int* __restrict__ seg_shape, const int* __restrict__ seg_type,
    const int* __restrict__ gt_off, const int* __restrict__ gt_idx, const double* __restrict__ params,
    int sqrt_flag, float* __restrict__ dist, double* __restrict__ dparam, int* __restrict__ status) {
  const int s = blockIdx.x;
  const int type = seg_type[s];
  const int b = seg_shape[s];
  const float* Pb = P + (size_t)b * N * 3;
  float th[FB_NT];
  for (int k = 0; k < FB_NT; ++k) th[k] = (float)params[(size_t)s * FB_NPAR + k];
  const int lo = gt_off[s], hi = gt_off[s + 1];
  double acc[FB_NT + 1];
  for (int k = 0; k <= FB_NT; ++k) acc[k] = 0.0;
  int nan_seen = 0;
  for (int t = lo + threadIdx.x; t < hi; t += 256) {
    const int i = gt_idx[t];
    const Df d = residual_point(type, Pb[3 * i], Pb[3 * i + 1], Pb[3 * i + 2], th, sqrt_flag);
    nan_seen |= (d.v != d.v);
    acc[0] += (double)d.v;
    for (int k = 0; k < FB_NT; ++k) acc[1 + k] += (double)d.d[k];
  }
  __shared__ double red[4][FB_NT + 1];
  __shared__ int nanflag;
  if (threadIdx.x == 0) nanflag = 0;
  __syncthreads();
  for (int k = 0; k <= FB_NT; ++k) {
    const double v = pn_wave_sum_d(acc[k]);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
  }
  if (nan_seen) atomicOr(&nanflag, 1);
  __syncthreads();
  if (threadIdx.x <= FB_NT) {
    const int k = threadIdx.x;
    const double v = ((red[0][k] + red[1][k]) + (red[2][k] + red[3][k])) / (double)max(hi - lo, 1);
    if (k == 0) dist[s] = (float)v;
    else dparam[(size_t)s * FB_NPAR + (k - 1)] = v;
  }
  if (threadIdx.x == 0 && nanflag) atomicOr(&status[s], 4);
}

// ---- stage 5: d loss / d w --------------------------------------------------------------
// gM_l = g_dist[s] * sum_k dparam[s][k] jac[s][k][l];   gw_i = sum_m gM_m e_m w_i^(e_m - 1) phi_m(z_i)
// (+ the cone's direct path).  grid (S, chunks of 256 points); gW is zero-initialised by the
// caller, entries (shape, row, stride * j) are written.
__global__ __launch_bounds__(256) void pn_wmom_bwd_kernel(
    const float* __restrict__ P, const float* __restrict__ Nrm, const float* __restrict__ W, int N, int Cp,
    int stride, float eps, const int* __restrict__ seg_shape, const int* __restrict__ seg_row,
    const int* __restrict__ seg_type, const float* __restrict__ g_dist, const double* __restrict__ dparam,
    const double* __restrict__ jac, const double* __restrict__ params, const double* __restrict__ cone_direct,
    float* __restrict__ gW) {
  __shared__ double gM[FB_NMOM];
  const int s = blockIdx.x;
  const int b = seg_shape[s], row = seg_row[s];
  const double gd = (double)g_dist[s];
  if (threadIdx.x < FB_NMOM) {
    double v = 0.0;
    for (int k = 0; k < FB_NT; ++k)
      v += dparam[(size_t)s * FB_NPAR + k] * jac[((size_t)s * FB_NPAR + k) * FB_NMOM + threadIdx.x];
    gM[threadIdx.x] = gd * v;
  }
  __syncthreads();
  const int n = (N + stride - 1) / stride;
  const int j = blockIdx.y * 256 + threadIdx.x;
  if (j >= n) return;
  const int i = j * stride;
  const float* Pb = P + (size_t)b * N * 3;
  const float* Nb = Nrm + (size_t)b * N * 3;
  const size_t wi = ((size_t)b * Cp + row) * N + i;
  const float wf = W[wi] + eps;
  const double w = (double)wf;
  const double z[7] = {1.0, Pb[3 * i], Pb[3 * i + 1], Pb[3 * i + 2], Nb[3 * i], Nb[3 * i + 1], Nb[3 * i + 2]};
  double g = wmom_bwd_point(gM, w, z);
  if (seg_type[s] == FB_CONE && cone_direct[s] != 0.0) {
    float c[3], a[3], ux, uy, uz, nu, t;
    for (int k = 0; k < 3; ++k) {
      c[k] = (float)params[(size_t)s * FB_NPAR + k];
      a[k] = (float)params[(size_t)s * FB_NPAR + 3 + k];
    }
    const float f = cone_acos_term((float)z[1], (float)z[2], (float)z[3], c, a, &ux, &uy, &uz, &nu, &t);
    g += gd * dparam[(size_t)s * FB_NPAR + 6] * cone_direct[s] * (double)f;
  }
  gW[wi] = (float)g;
}

// ---- B-spline surface evaluation (src/fitting_utils.py:609-622) ----------------------------
// out[s, u, v, :] = A_s ( sum_ij nu[u,i] nv[v,j] ctrl[s,i,j,:] ) + t_s   (affine: the
// de-standardisation of primitive_forward.py:60-72; NULL = identity); `wrap` appends the first
// u-row again (closed splines, primitive_forward.py:377-385): out has (gu + wrap) * gv points.
__global__ __launch_bounds__(256) void pn_bspline_eval_kernel(const float* __restrict__ nu,
                                                              const float* __restrict__ nv,
                                                              const float* __restrict__ ctrl,
                                                              const float* __restrict__ affine, int gu, int gv,
                                                              int cu, int cv, int wrap, float* __restrict__ out) {
  extern __shared__ float sm[];
  float* sc = sm;                   // ctrl of this item: cu*cv*3
  float* tmp = sm + cu * cv * 3;    // nu @ ctrl: gu*cv*3
  const int s = blockIdx.x;
  const float* cs = ctrl + (size_t)s * cu * cv * 3;
  for (int t = threadIdx.x; t < cu * cv * 3; t += 256) sc[t] = cs[t];
  __syncthreads();
  for (int t = threadIdx.x; t < gu * cv * 3; t += 256) {
    const int u = t / (cv * 3), r = t - u * cv * 3;   // r = j*3 + c
    float acc = 0.f;
    for (int i = 0; i < cu; ++i) acc = fmaf(nu[u * cu + i], sc[i * cv * 3 + r], acc);
    tmp[t] = acc;
  }
  __syncthreads();
  float A[12];
  if (affine)
    for (int k = 0; k < 12; ++k) A[k] = affine[(size_t)s * 12 + k];
  const int rows = gu + wrap;
  for (int t = threadIdx.x; t < rows * gv; t += 256) {
    const int ur = t / gv, v = t - ur * gv;
    const int u = ur < gu ? ur : 0;
    float x = 0.f, y = 0.f, z = 0.f;
    for (int j = 0; j < cv; ++j) {
      const float b = nv[v * cv + j];
      x = fmaf(b, tmp[(u * cv + j) * 3 + 0], x);
      y = fmaf(b, tmp[(u * cv + j) * 3 + 1], y);
      z = fmaf(b, tmp[(u * cv + j) * 3 + 2], z);
    }
    float* o = out + ((size_t)s * rows * gv + t) * 3;
    if (affine) {
      o[0] = A[0] * x + A[1] * y + A[2] * z + A[3];
      o[1] = A[4] * x + A[5] * y + A[6] * z + A[7];
      o[2] = A[8] * x + A[9] * y + A[10] * z + A[11];
    } else {
      o[0] = x; o[1] = y; o[2] = z;
    }
  }
}

// adjoint: gctrl[s,i,j,:] = sum_uv nu[u,i] nv[v,j] A_s^T gout[s,u,v,:]  (wrapped row added to row 0)
__global__ __launch_bounds__(256) void pn_bspline_eval_bwd_kernel(const float* __restrict__ nu,
                                                                  const float* __restrict__ nv,
                                                                  const float* __restrict__ gout,
                                                                  const float* __restrict__ affine, int gu,
                                                                  int gv, int cu, int cv, int wrap,
                                                                  float* __restrict__ gctrl) {
  extern __shared__ float sm[];
  float* g = sm;                    // A^T gout: gu*gv*3
  float* tmp = sm + gu * gv * 3;    // sum_v nv[v,j] g[u,v,:]: gu*cv*3
  const int s = blockIdx.x;
  const int rows = gu + wrap;
  const float* gs = gout + (size_t)s * rows * gv * 3;
  float A[12];
  if (affine)
    for (int k = 0; k < 12; ++k) A[k] = affine[(size_t)s * 12 + k];
  for (int t = threadIdx.x; t < gu * gv; t += 256) {
    float x = gs[3 * t], y = gs[3 * t + 1], z = gs[3 * t + 2];
    if (wrap && t < gv) {   // row gu repeats row 0
      x += gs[3 * (gu * gv + t)];
      y += gs[3 * (gu * gv + t) + 1];
      z += gs[3 * (gu * gv + t) + 2];
    }
    if (affine) {
      g[3 * t + 0] = A[0] * x + A[4] * y + A[8] * z;
      g[3 * t + 1] = A[1] * x + A[5] * y + A[9] * z;
      g[3 * t + 2] = A[2] * x + A[6] * y + A[10] * z;
    } else {
      g[3 * t] = x; g[3 * t + 1] = y; g[3 * t + 2] = z;
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < gu * cv * 3; t += 256) {
    const int u = t / (cv * 3), r = t - u * cv * 3, j = r / 3, c = r - 3 * j;
    float acc = 0.f;
    for (int v = 0; v < gv; ++v) acc = fmaf(nv[v * cv + j], g[(u * gv + v) * 3 + c], acc);
    tmp[t] = acc;
  }
  __syncthreads();
  for (int t = threadIdx.x; t < cu * cv * 3; t += 256) {
    const int i = t / (cv * 3), r = t - i * cv * 3;
    float acc = 0.f;
    for (int u = 0; u < gu; ++u) acc = fmaf(nu[u * cu + i], tmp[u * cv * 3 + r], acc);
    gctrl[(size_t)s * cu * cv * 3 + t] = acc;
  }
}

// ---- C ABI -----------------------------------------------------------------------------
extern "C" int pn_weighted_moments_chunks(void) { return FB_CH; }
extern "C" int pn_weighted_moments_count(void) { return FB_NMOM; }

extern "C" int pn_weighted_moments_f64(const float* P, const float* Nrm, const float* W, int B, int N, int Cp,
                                       int stride, float eps, const int* seg_shape, const int* seg_row, int S,
                                       double* partial, void* stream) {
  PN_CHECK_ARG(P && Nrm && W && seg_shape && seg_row && partial, "pn_weighted_moments_f64: null argument");
  PN_CHECK_ARG(B > 0 && N > 0 && Cp > 0 && stride > 0 && S > 0, "pn_weighted_moments_f64: bad sizes");
  PN_PROF("fit_moments", (hipStream_t)stream);
  hipLaunchKernelGGL(pn_wmom_fwd_kernel, dim3(S, FB_CH), dim3(256), 0, (hipStream_t)stream, P, Nrm, W, N, Cp,
                     stride, eps, seg_shape, seg_row, partial);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

extern "C" int pn_primitive_fit_f64(const double* partial, const int* seg_type, const int* seg_rows, int S,
                                    double* params, double* jac, int* status, void* stream) {
  PN_CHECK_ARG(partial && seg_type && seg_rows && params && jac && status && S > 0,
               "pn_primitive_fit_f64: bad arguments");
  PN_PROF("fit_solve", (hipStream_t)stream);
  hipLaunchKernelGGL(pn_primfit_kernel, dim3(S), dim3(64), 0, (hipStream_t)stream, partial, seg_type, seg_rows,
                     S, params, jac, status);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

extern "C" int pn_cone_angle_f64(const float* P, const float* W, int B, int N, int Cp, int stride, float eps,
                                 const int* seg_shape, const int* seg_row, const int* seg_type,
                                 const int* status, int S, double* params, double* jac, double* cone_direct,
                                 void* stream) {
  PN_CHECK_ARG(P && W && seg_shape && seg_row && seg_type && status && params && jac && cone_direct && S > 0,
               "pn_cone_angle_f64: bad arguments");
  PN_PROF("fit_cone_angle", (hipStream_t)stream);
  hipLaunchKernelGGL(pn_cone_angle_kernel, dim3(S), dim3(256), 0, (hipStream_t)stream, P, W, N, Cp, stride, eps,
                     seg_shape, seg_row, seg_type, status, params, jac, cone_direct);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

extern "C" int pn_primitive_residual_f32(const float* P, int B, int N, const int* seg_shape, const int* seg_type,
                                         const int* gt_off, const int* gt_idx, int S, const double* params,
                                         int sqrt_flag, float* dist, double* dparam, int* status, void* stream) {
  PN_CHECK_ARG(P && seg_shape && seg_type && gt_off && gt_idx && params && dist && dparam && status && S > 0,
               "pn_primitive_residual_f32: bad arguments");
  PN_PROF("fit_residual", (hipStream_t)stream);
  hipLaunchKernelGGL(pn_prim_residual_kernel, dim3(S), dim3(256), 0, (hipStream_t)stream, P, N, seg_shape,
                     seg_type, gt_off, gt_idx, params, sqrt_flag, dist, dparam, status);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

extern "C" int pn_weighted_moments_bwd_f32(const float* P, const float* Nrm, const float* W, int B, int N, int Cp,
                                           int stride, float eps, const int* seg_shape, const int* seg_row,
                                           const int* seg_type, int S, const float* g_dist, const double* dparam,
                                           const double* jac, const double* params, const double* cone_direct,
                                           float* gW, void* stream) {
  PN_CHECK_ARG(P && Nrm && W && seg_shape && seg_row && seg_type && g_dist && dparam && jac && params &&
                   cone_direct && gW && S > 0,
               "pn_weighted_moments_bwd_f32: bad arguments");
  const int n = (N + stride - 1) / stride;
  PN_PROF("fit_moments_bwd", (hipStream_t)stream);
  hipLaunchKernelGGL(pn_wmom_bwd_kernel, dim3(S, pn_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, P, Nrm, W,
                     N, Cp, stride, eps, seg_shape, seg_row, seg_type, g_dist, dparam, jac, params, cone_direct,
                     gW);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

extern "C" int pn_bspline_eval_f32(const float* nu, const float* nv, const float* ctrl, const float* affine, int S,
                                   int gu, int gv, int cu, int cv, int wrap, float* out, void* stream) {
  PN_CHECK_ARG(nu && nv && ctrl && out && S > 0 && gu > 0 && gv > 0 && cu > 0 && cv > 0 && (wrap == 0 || wrap == 1),
               "pn_bspline_eval_f32: bad arguments");
  const size_t lds = (size_t)(cu * cv * 3 + gu * cv * 3) * sizeof(float);
  PN_CHECK_ARG(lds <= 64 * 1024, "pn_bspline_eval_f32: grid too large for the LDS tile (%zu bytes)", lds);
  PN_PROF("bspline_eval", (hipStream_t)stream);
  hipLaunchKernelGGL(pn_bspline_eval_kernel, dim3(S), dim3(256), lds, (hipStream_t)stream, nu, nv, ctrl, affine,
                     gu, gv, cu, cv, wrap, out);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

extern "C" int pn_bspline_eval_bwd_f32(const float* nu, const float* nv, const float* gout, const float* affine,
                                       int S, int gu, int gv, int cu, int cv, int wrap, float* gctrl,
                                       void* stream) {
  PN_CHECK_ARG(nu && nv && gout && gctrl && S > 0 && gu > 0 && gv > 0 && cu > 0 && cv > 0 &&
                   (wrap == 0 || wrap == 1),
               "pn_bspline_eval_bwd_f32: bad arguments");
  const size_t lds = (size_t)(gu * gv * 3 + gu * cv * 3) * sizeof(float);
  PN_CHECK_ARG(lds <= 64 * 1024, "pn_bspline_eval_bwd_f32: grid too large for the LDS tile (%zu bytes)", lds);
  PN_PROF("bspline_eval_bwd", (hipStream_t)stream);
  hipLaunchKernelGGL(pn_bspline_eval_bwd_kernel, dim3(S), dim3(256), lds, (hipStream_t)stream, nu, nv, gout,
                     affine, gu, gv, cu, cv, wrap, gctrl);
  PN_CHECK_LAUNCH();
  return PN_OK;
}
