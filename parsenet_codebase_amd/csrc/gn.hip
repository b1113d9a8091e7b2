// GroupNorm (+ReLU) (+max over points) for the per-point heads, channel-first (B,C,N) fp32.
//
// Replaces torch's GroupNorm -> ReLU (-> max over N) chains of
//   src/PointNet.py:216-218 (bnmlp1 + relu + max), :274-283 (bn1, bn2, bn_seg_prob1,
//   bn_prim_prob1 + relu)
// whose statistics kernel launches one block per (item, group) — 32 blocks on a 256-CU chip for
// a 164 MB tensor.  Here one 256-thread block owns one (item, channel) ROW of N contiguous
// floats (>= 1024 blocks), reads it with 16-byte lanes, and everything per-group is derived
// from the per-row partial sums:
//   fwd   rows -> (sum, sum of squares[, max, argmax, min, argmin])      pn_gn_rows_fwd
//         group mean / rstd in fp64 from the row sums                     pn_gn_group_moments
//         out = relu(gamma * (y - mean) * rstd + beta)                    pn_gn_apply_fwd
//   bwd   rows -> (sum gz, sum gz*yhat), gz = gout * [z > 0]              pn_gn_rows_bwd
//         c1, c2 (group means of gamma*gz and gamma*gz*yhat), dgamma, dbeta   pn_gn_group_bwd
//         dy = rstd * (gamma*gz - c1 - yhat*c2)                           pn_gn_apply_bwd
// The max variant (bnmlp1 -> relu -> max over N) uses the monotonicity of norm + ReLU per
// channel: only the row maximum (minimum where gamma < 0) is normalised; the backward places
// the upstream gradient at the arg position and adds the dense mean/variance terms.
// Round 6: every row kernel takes an optional ROW BIAS rb[b * bs + c] (bs = C: one value per (item, channel); bs = 0:
// one per channel) that is added to y at load, e = fl(y + rb): the convolution's bias (src/PointNet.py:196, 268-284:
// Conv1d(..., bias=True) followed by GroupNorm) and the per-item global term of conv1 are no longer written out by
// a separate pass over the (B,C,N) tensor — the same fp32 addition, so the results are bit-identical.
#include "common.h"

#define GN_T 256
#define GN_RB(ROW, C_) (rowbias ? rowbias[(size_t)((ROW) / (C_)) * rb_bstride + ((ROW) % (C_))] : 0.f)

__device__ static inline float gn_block_sum(float v, float* sh) {
  v = pn_wave_sum(v);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) sh[w] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// rows: one block per (b,c).  want_ext: also the row max/min and their positions.
__global__ __launch_bounds__(GN_T) void pn_gn_rows_fwd_kernel(const float* __restrict__ y, int N,
                                                              float* __restrict__ rsum,
                                                              float* __restrict__ rsq, int want_ext,
                                                              float* __restrict__ rmax,
                                                              int* __restrict__ amax,
                                                              float* __restrict__ rmin,
                                                              int* __restrict__ amin,
                                                              const float* __restrict__ rowbias, int rb_bstride,
                                                              int C) {
  __shared__ float sh[4];
  __shared__ unsigned long long shk[2][4];
  const size_t row = blockIdx.x;
  const float rbv = GN_RB(row, C);
  const float* __restrict__ p = y + row * N;
  float s = 0.f, q = 0.f;
  float mx = -__builtin_inff(), mn = __builtin_inff();
  int ax = 0, an = 0;
  const int n4 = ((reinterpret_cast<uintptr_t>(p) & 15) == 0) ? (N >> 2) : 0;
  for (int i = threadIdx.x; i < n4; i += GN_T) {
    const float4 v = reinterpret_cast<const float4*>(p)[i];
    const float e[4] = {v.x + rbv, v.y + rbv, v.z + rbv, v.w + rbv};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      s += e[u];
      q = __builtin_fmaf(e[u], e[u], q);
      if (want_ext) {
        if (e[u] > mx) { mx = e[u]; ax = 4 * i + u; }
        if (e[u] < mn) { mn = e[u]; an = 4 * i + u; }
      }
    }
  }
  for (int i = 4 * n4 + threadIdx.x; i < N; i += GN_T) {
    const float e = p[i] + rbv;
    s += e;
    q = __builtin_fmaf(e, e, q);
    if (want_ext) {
      if (e > mx) { mx = e; ax = i; }
      if (e < mn) { mn = e; an = i; }
    }
  }
  const float ts = gn_block_sum(s, sh);
  const float tq = gn_block_sum(q, sh);
  if (threadIdx.x == 0) {
    rsum[row] = ts;
    rsq[row] = tq;
  }
  if (want_ext) {
    // (value, position) keys: max -> larger value, then smaller position; min via negation
    unsigned long long kx = ((unsigned long long)pn_f2ord(mx) << 32) | (0xffffffffu - (uint32_t)ax);
    unsigned long long kn = ((unsigned long long)pn_f2ord(-mn) << 32) | (0xffffffffu - (uint32_t)an);
    kx = pn_wave_max_u64(kx);
    kn = pn_wave_max_u64(kn);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (l == 0) {
      shk[0][w] = kx;
      shk[1][w] = kn;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int i = 1; i < 4; ++i) {
        kx = shk[0][i] > kx ? shk[0][i] : kx;
        kn = shk[1][i] > kn ? shk[1][i] : kn;
      }
      rmax[row] = pn_ord2f((uint32_t)(kx >> 32));
      amax[row] = (int)(0xffffffffu - (uint32_t)kx);
      rmin[row] = -pn_ord2f((uint32_t)(kn >> 32));
      amin[row] = (int)(0xffffffffu - (uint32_t)kn);
    }
  }
}

// one thread per (b,g): fp64 moments of the group from its rows
__global__ void pn_gn_group_moments_kernel(const float* __restrict__ rsum,
                                           const float* __restrict__ rsq, int BG, int Cg, int N,
                                           float eps, float* __restrict__ mean,
                                           float* __restrict__ rstd) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= BG) return;
  double s = 0.0, q = 0.0;
  for (int c = 0; c < Cg; ++c) {
    s += (double)rsum[(size_t)t * Cg + c];
    q += (double)rsq[(size_t)t * Cg + c];
  }
  const double cnt = (double)Cg * (double)N;
  const double m = s / cnt;
  double var = q / cnt - m * m;
  if (var < 0.0) var = 0.0;
  mean[t] = (float)m;
  rstd[t] = (float)(1.0 / sqrt(var + (double)eps));
}

__global__ __launch_bounds__(GN_T) void pn_gn_apply_fwd_kernel(const float* __restrict__ y,
                                                               const float* __restrict__ mean,
                                                               const float* __restrict__ rstd,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, int C,
                                                               int Cg, int N, int relu,
                                                               float* __restrict__ out,
                                                               const float* __restrict__ rowbias, int rb_bstride) {
  const size_t row = blockIdx.x;
  const float rbv = GN_RB(row, C);
  const int c = (int)(row % C);
  const int grp = (int)(row / Cg);  // (b*C + c) / Cg = b*G + c/Cg
  const float a = gamma[c] * rstd[grp];
  const float sft = __builtin_fmaf(-mean[grp], a, beta[c]);
  const float* __restrict__ p = y + row * N;
  float* __restrict__ o = out + row * N;
  const bool al = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(o)) & 15) == 0;
  const int n4 = al ? (N >> 2) : 0;
  for (int i = threadIdx.x; i < n4; i += GN_T) {
    float4 v = reinterpret_cast<const float4*>(p)[i];
    v.x = __builtin_fmaf(v.x + rbv, a, sft);
    v.y = __builtin_fmaf(v.y + rbv, a, sft);
    v.z = __builtin_fmaf(v.z + rbv, a, sft);
    v.w = __builtin_fmaf(v.w + rbv, a, sft);
    if (relu) {
      v.x = fmaxf(v.x, 0.f);
      v.y = fmaxf(v.y, 0.f);
      v.z = fmaxf(v.z, 0.f);
      v.w = fmaxf(v.w, 0.f);
    }
    reinterpret_cast<float4*>(o)[i] = v;
  }
  for (int i = 4 * n4 + threadIdx.x; i < N; i += GN_T) {
    float v = __builtin_fmaf(p[i] + rbv, a, sft);
    o[i] = relu ? fmaxf(v, 0.f) : v;
  }
}

// backward rows: ra = sum gz, rb = sum gz * yhat with gz = gout * [relu ? z > 0 : 1]
__global__ __launch_bounds__(GN_T) void pn_gn_rows_bwd_kernel(
    const float* __restrict__ gout, const float* __restrict__ y, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
    int C, int Cg, int N, int relu, float* __restrict__ ra, float* __restrict__ rb,
    const float* __restrict__ rowbias, int rb_bstride) {
  __shared__ float sh[4];
  const size_t row = blockIdx.x;
  const float rbv = GN_RB(row, C);
  const int c = (int)(row % C);
  const int grp = (int)(row / Cg);
  const float mu = mean[grp], r = rstd[grp], g = gamma[c], bt = beta[c];
  const float* __restrict__ py = y + row * N;
  const float* __restrict__ pg = gout + row * N;
  float a = 0.f, b = 0.f;
  for (int i = threadIdx.x; i < N; i += GN_T) {
    const float yh = ((py[i] + rbv) - mu) * r;
    const float z = __builtin_fmaf(g, yh, bt);
    const float gz = (!relu || z > 0.f) ? pg[i] : 0.f;
    a += gz;
    b = __builtin_fmaf(gz, yh, b);
  }
  const float ta = gn_block_sum(a, sh);
  const float tb = gn_block_sum(b, sh);
  if (threadIdx.x == 0) {
    ra[row] = ta;
    rb[row] = tb;
  }
}

// one thread per (b,g): c1 = mean_group(gamma*gz), c2 = mean_group(gamma*gz*yhat)
__global__ void pn_gn_group_bwd_kernel(const float* __restrict__ ra, const float* __restrict__ rb,
                                       const float* __restrict__ gamma, int BG, int C, int Cg,
                                       double count, float* __restrict__ c1c2) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= BG) return;
  const int G = C / Cg, g = t % G;
  double s1 = 0.0, s2 = 0.0;
  for (int c = 0; c < Cg; ++c) {
    const double gm = (double)gamma[g * Cg + c];
    s1 += gm * (double)ra[(size_t)t * Cg + c];
    s2 += gm * (double)rb[(size_t)t * Cg + c];
  }
  c1c2[2 * t] = (float)(s1 / count);
  c1c2[2 * t + 1] = (float)(s2 / count);
}

// dy = rstd * (gamma*gz - c1 - yhat*c2).  sparse: gz is non-zero only at position arg[row]
// with value gsp[row] (the max variant); otherwise gz = gout * relu mask.
__global__ __launch_bounds__(GN_T) void pn_gn_apply_bwd_kernel(
    const float* __restrict__ gout, const float* __restrict__ y, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ c1c2, int C, int Cg, int N, int relu, const float* __restrict__ gsp,
    const int* __restrict__ arg, float* __restrict__ dy, const float* __restrict__ rowbias, int rb_bstride) {
  const size_t row = blockIdx.x;
  const float rbv = GN_RB(row, C);
  const int c = (int)(row % C);
  const int grp = (int)(row / Cg);
  const float mu = mean[grp], r = rstd[grp], g = gamma[c], bt = beta[c];
  const float c1 = c1c2[2 * grp], c2 = c1c2[2 * grp + 1];
  const float* __restrict__ py = y + row * N;
  float* __restrict__ pd = dy + row * N;
  if (gsp) {
    const int at = arg[row];
    const float gv = gsp[row];
    for (int i = threadIdx.x; i < N; i += GN_T) {
      const float yh = ((py[i] + rbv) - mu) * r;
      float v = -c1 - yh * c2;
      if (i == at) v += g * gv;
      pd[i] = r * v;
    }
  } else {
    const float* __restrict__ pg = gout + row * N;
    for (int i = threadIdx.x; i < N; i += GN_T) {
      const float yh = ((py[i] + rbv) - mu) * r;
      const float z = __builtin_fmaf(g, yh, bt);
      const float gz = (!relu || z > 0.f) ? pg[i] : 0.f;
      pd[i] = r * (g * gz - c1 - yh * c2);
    }
  }
}

extern "C" int pn_gn_rows_fwd_f32(const float* y, int B, int C, int N, float* rsum, float* rsq,
                                  float* rmax, int* amax, float* rmin, int* amin, const float* rowbias,
                                  int rb_bstride, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PN_CHECK_ARG(y && rsum && rsq && B > 0 && C > 0 && N > 0, "pn_gn_rows_fwd_f32: bad arguments");
  const int want = rmax != nullptr;
  PN_CHECK_ARG(!want || (amax && rmin && amin), "pn_gn_rows_fwd_f32: incomplete extreme outputs");
  PN_PROF("gn_rows_fwd", stream);
  hipLaunchKernelGGL(pn_gn_rows_fwd_kernel, dim3(B * C), dim3(GN_T), 0, stream, y, N, rsum, rsq,
                     want, rmax, amax, rmin, amin, rowbias, rb_bstride, C);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

extern "C" int pn_gn_group_moments_f32(const float* rsum, const float* rsq, int B, int C,
                                       int groups, int N, float eps, float* mean, float* rstd,
                                       void* stream) {
  PN_CHECK_ARG(rsum && rsq && mean && rstd && groups > 0 && C % groups == 0,
               "pn_gn_group_moments_f32: bad arguments");
  const int BG = B * groups;
  hipLaunchKernelGGL(pn_gn_group_moments_kernel, dim3(pn_cdiv(BG, 64)), dim3(64), 0,
                     (hipStream_t)stream, rsum, rsq, BG, C / groups, N, eps, mean, rstd);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

extern "C" int pn_gn_apply_fwd_f32(const float* y, const float* mean, const float* rstd,
                                   const float* gamma, const float* beta, int B, int C, int groups,
                                   int N, int relu, float* out, const float* rowbias, int rb_bstride,
                                   void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PN_CHECK_ARG(y && mean && rstd && gamma && beta && out, "pn_gn_apply_fwd_f32: null pointer");
  PN_PROF("gn_apply_fwd", stream);
  hipLaunchKernelGGL(pn_gn_apply_fwd_kernel, dim3(B * C), dim3(GN_T), 0, stream, y, mean, rstd,
                     gamma, beta, C, C / groups, N, relu, out, rowbias, rb_bstride);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

extern "C" int pn_gn_rows_bwd_f32(const float* gout, const float* y, const float* mean,
                                  const float* rstd, const float* gamma, const float* beta, int B,
                                  int C, int groups, int N, int relu, float* ra, float* rb,
                                  const float* rowbias, int rb_bstride, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PN_CHECK_ARG(gout && y && mean && rstd && gamma && beta && ra && rb, "pn_gn_rows_bwd_f32: null");
  PN_PROF("gn_rows_bwd", stream);
  hipLaunchKernelGGL(pn_gn_rows_bwd_kernel, dim3(B * C), dim3(GN_T), 0, stream, gout, y, mean, rstd,
                     gamma, beta, C, C / groups, N, relu, ra, rb, rowbias, rb_bstride);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

extern "C" int pn_gn_group_bwd_f32(const float* ra, const float* rb, const float* gamma, int B,
                                   int C, int groups, int N, float* c1c2, void* stream) {
  PN_CHECK_ARG(ra && rb && gamma && c1c2, "pn_gn_group_bwd_f32: null pointer");
  const int BG = B * groups;
  const double count = (double)(C / groups) * (double)N;
  hipLaunchKernelGGL(pn_gn_group_bwd_kernel, dim3(pn_cdiv(BG, 64)), dim3(64), 0, (hipStream_t)stream,
                     ra, rb, gamma, BG, C, C / groups, count, c1c2);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

extern "C" int pn_gn_apply_bwd_f32(const float* gout, const float* y, const float* mean,
                                   const float* rstd, const float* gamma, const float* beta,
                                   const float* c1c2, int B, int C, int groups, int N, int relu,
                                   const float* gsp, const int* arg, float* dy, const float* rowbias,
                                   int rb_bstride, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PN_CHECK_ARG(y && mean && rstd && gamma && beta && c1c2 && dy && (gout || (gsp && arg)),
               "pn_gn_apply_bwd_f32: null pointer");
  PN_PROF("gn_apply_bwd", stream);
  hipLaunchKernelGGL(pn_gn_apply_bwd_kernel, dim3(B * C), dim3(GN_T), 0, stream, gout, y, mean,
                     rstd, gamma, beta, c1c2, C, C / groups, N, relu, gsp, arg, dy, rowbias, rb_bstride);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

// ---- the (B,C) tail of max_n relu(GroupNorm(y)) ------------------------------------------------------------------
// norms._GroupNormReLUMax picked the row extremum by the sign of gamma, normalised it, applied the affine map and
// the ReLU as ten tensor-library launches on B x C values (and four more in the backward pass); one launch each
// here, the same fp32 operations in the same order (no contraction: separately rounded multiply and add).
__global__ __launch_bounds__(256) void pn_gn_max_finish_kernel(const float* __restrict__ rmax, const int* __restrict__ amax,
                                                               const float* __restrict__ rmin, const int* __restrict__ amin,
                                                               const float* __restrict__ mean, const float* __restrict__ rstd,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               int BC, int C, int groups, float* __restrict__ yhat,
                                                               float* __restrict__ z, int* __restrict__ arg,
                                                               float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= BC) return;
  const int c = i % C, b = i / C, g = b * groups + c / (C / groups);
  const float ga = gamma[c];
  const bool pos = ga >= 0.f;
  const float ext = pos ? rmax[i] : rmin[i];
  arg[i] = pos ? amax[i] : amin[i];
  const float yh = __fmul_rn(__fsub_rn(ext, mean[g]), rstd[g]);
  const float zz = __fadd_rn(__fmul_rn(ga, yh), beta[c]);
  yhat[i] = yh;
  z[i] = zz;
  out[i] = zz < 0.f ? 0.f : zz;          // relu as the tensor library's clamp: a NaN stays a NaN, -0 stays -0
}

__global__ __launch_bounds__(256) void pn_gn_max_bwd_prep_kernel(const float* __restrict__ g, const float* __restrict__ z,
                                                                 const float* __restrict__ yhat, int BC,
                                                                 float* __restrict__ gz, float* __restrict__ rb) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= BC) return;
  const float v = __fmul_rn(g[i], z[i] > 0.f ? 1.f : 0.f);
  gz[i] = v;
  rb[i] = __fmul_rn(v, yhat[i]);
}

extern "C" int pn_gn_max_finish_f32(const float* rmax, const int* amax, const float* rmin, const int* amin,
                                    const float* mean, const float* rstd, const float* gamma, const float* beta,
                                    int B, int C, int groups, float* yhat, float* z, int* arg, float* out,
                                    void* stream) {
  PN_CHECK_ARG(rmax && amax && rmin && amin && mean && rstd && gamma && beta && yhat && z && arg && out && B > 0 &&
               C > 0 && groups > 0 && C % groups == 0, "pn_gn_max_finish_f32: bad arguments");
  hipLaunchKernelGGL(pn_gn_max_finish_kernel, dim3(pn_cdiv(B * C, 256)), dim3(256), 0, (hipStream_t)stream, rmax, amax,
                     rmin, amin, mean, rstd, gamma, beta, B * C, C, groups, yhat, z, arg, out);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

extern "C" int pn_gn_max_bwd_prep_f32(const float* g, const float* z, const float* yhat, int B, int C, float* gz,
                                      float* rb, void* stream) {
  PN_CHECK_ARG(g && z && yhat && gz && rb && B > 0 && C > 0, "pn_gn_max_bwd_prep_f32: bad arguments");
  hipLaunchKernelGGL(pn_gn_max_bwd_prep_kernel, dim3(pn_cdiv(B * C, 256)), dim3(256), 0, (hipStream_t)stream, g, z, yhat,
                     B * C, gz, rb);
  PN_CHECK_LAUNCH();
  return PN_OK;
}
