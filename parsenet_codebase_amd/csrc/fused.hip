// Round-3 fusions of what used to be strings of small tensor-library launches on the timed
// path (profiles/r02_cfg5_torch_sites.txt: 905 launches, 6.2 ms per cfg5 step):
//   * edge-conv backward, step A with all its reductions (d gamma, d beta, the group means c1 / c2
//     of the normalisation gradient) — graph.py used ~20 launches per layer for them;
//   * the triplet embedding loss of src/segment_loss.py:85-123, forward and backward, one
//     workgroup per (shape, segment pair);
//   * memberships: centres . embedding^T, weights_normalize (src/fitting_utils.py:306-325) and the
//     nearest-centre labels of the non-maximum suppression (src/mean_shift.py:176-178) in one pass
//     over the points, with the exact backward;
//   * per-channel affine map + activation (evaluation-mode BatchNorm1d + LeakyReLU / ReLU of the
//     frozen SplineNets' heads, src/model.py:160-176).
// All reductions have a fixed order (partial sums per workgroup, combined in index order, fp64
// where a cancellation could matter): results are reproducible run to run.
#include "common.h"

// =============================================================================================
// edge-conv backward, step A + reductions
// =============================================================================================
// t[b,n,c] = gamma[c] * gout[b,c,n] * (z > 0 ? 1 : slope),  z = gamma*yhat + beta,
// yhat = (yext - mean) * rstd; partial[b][nblk][c] = (sum gz, sum gz*yhat) over the block's 32 points.
__global__ __launch_bounds__(256) void pn_ecb_prep_kernel(
    const float* __restrict__ gout, const float* __restrict__ yext, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
    int N, int Cout, int Cg, int per_sample, float slope, float* __restrict__ t,
    float2* __restrict__ partial) {
  __shared__ float tile[32][33];
  __shared__ float2 red[8][32];
  const int b = blockIdx.z;
  const int c0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int G = Cout / Cg;
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, n = n0 + tx;
    if (n < N && c < Cout) tile[i][tx] = gout[((size_t)b * Cout + c) * N + n];
  }
  __syncthreads();
  float a = 0.f, bm = 0.f;
  const int c = c0 + tx;
  if (c < Cout) {
    const int sidx = (per_sample ? b : 0) * G + c / Cg;
    const float mu = mean[sidx], r = rstd[sidx], ga = gamma[c], be = beta[c];
    for (int i = ty; i < 32; i += 8) {
      const int n = n0 + i;
      if (n < N) {
        const size_t o = ((size_t)b * N + n) * Cout + c;
        const float yh = (yext[o] - mu) * r;
        const float z = __builtin_fmaf(ga, yh, be);
        const float gz = tile[tx][i] * (z > 0.f ? 1.f : slope);
        t[o] = gz * ga;
        a += gz;
        bm += gz * yh;
      }
    }
  }
  red[ty][tx] = make_float2(a, bm);
  __syncthreads();
  if (ty == 0 && c < Cout) {
    float2 s = red[0][tx];
#pragma unroll
    for (int j = 1; j < 8; ++j) {
      s.x += red[j][tx].x;
      s.y += red[j][tx].y;
    }
    partial[((size_t)b * gridDim.y + blockIdx.y) * Cout + c] = s;
  }
}

// AB[b][c] = (sum_n gz, sum_n gz*yhat) in fp64.  Block = 16 channels x 16 strided sub-sums over the
// point blocks, combined in sub-sum order (fixed order; a single thread walking all ~300 partials of
// a channel made this a latency-bound 80 us launch).
__global__ __launch_bounds__(256) void pn_ecb_reduce_kernel(const float2* __restrict__ partial, int nblk,
                                                            int Cout, double2* __restrict__ AB) {
  __shared__ double2 red[16][16];
  const int b = blockIdx.y;
  const int cl = threadIdx.x & 15, jg = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  double a = 0.0, bm = 0.0;
  if (c < Cout)
    for (int j = jg; j < nblk; j += 16) {
      const float2 p = partial[((size_t)b * nblk + j) * Cout + c];
      a += (double)p.x;
      bm += (double)p.y;
    }
  red[jg][cl] = make_double2(a, bm);
  __syncthreads();
  if (jg == 0 && c < Cout) {
    double2 s = red[0][cl];
#pragma unroll
    for (int j = 1; j < 16; ++j) {
      s.x += red[j][cl].x;
      s.y += red[j][cl].y;
    }
    AB[(size_t)b * Cout + c] = s;
  }
}

// dbeta, dgamma over the batch and the group means c1c2[s][g] = (sum gamma_c A, sum gamma_c B) / M
// over the channels of group g (and over the batch for batch statistics).
__global__ __launch_bounds__(256) void pn_ecb_finish_kernel(const double2* __restrict__ AB, int B, int Cout, int Cg,
                                                            int per_sample, int dense, double M,
                                                            const float* __restrict__ gamma,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            float* __restrict__ c1c2) {
  const int G = Cout / Cg;
  for (int c = threadIdx.x; c < Cout; c += blockDim.x) {
    double a = 0.0, bm = 0.0;
    for (int b = 0; b < B; ++b) {
      const double2 v = AB[(size_t)b * Cout + c];
      a += v.x;
      bm += v.y;
    }
    dbeta[c] = (float)a;
    dgamma[c] = (float)bm;
  }
  const int S = per_sample ? B : 1;
  for (int e = threadIdx.x; e < S * G; e += blockDim.x) {
    const int sidx = e / G, g = e - sidx * G;
    double s1 = 0.0, s2 = 0.0;
    if (dense)
      for (int c = g * Cg; c < (g + 1) * Cg; ++c)
        for (int b = per_sample ? sidx : 0; b < (per_sample ? sidx + 1 : B); ++b) {
          const double2 v = AB[(size_t)b * Cout + c];
          s1 += (double)gamma[c] * v.x;
          s2 += (double)gamma[c] * v.y;
        }
    c1c2[(size_t)e * 2] = (float)(s1 / M);
    c1c2[(size_t)e * 2 + 1] = (float)(s2 / M);
  }
}

extern "C" size_t pn_edgeconv_bwd_stats_workspace(int B, int N, int Cout) {
  return pn_align_up((size_t)B * pn_cdiv(N, 32) * Cout * sizeof(float2), 256) +
         pn_align_up((size_t)B * Cout * sizeof(double2), 256);
}

extern "C" int pn_edgeconv_bwd_stats_f32(const float* gout, const float* yext, const float* mean, const float* rstd,
                                         const float* gamma, const float* beta, int B, int N, int k, int Cout,
                                         int groups, int per_sample, int dense, float slope, float* t,
                                         float* dgamma, float* dbeta, float* c1c2, void* workspace,
                                         size_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PN_CHECK_ARG(gout && yext && mean && rstd && gamma && beta && t && dgamma && dbeta && c1c2 && workspace,
               "pn_edgeconv_bwd_stats_f32: null pointer");
  PN_CHECK_ARG(B > 0 && N > 0 && k > 0 && groups > 0 && Cout % groups == 0,
               "pn_edgeconv_bwd_stats_f32: B=%d N=%d k=%d Cout=%d groups=%d", B, N, k, Cout, groups);
  PN_CHECK_ARG(workspace_bytes >= pn_edgeconv_bwd_stats_workspace(B, N, Cout),
               "pn_edgeconv_bwd_stats_f32: workspace too small");
  const int nblk = pn_cdiv(N, 32), Cg = Cout / groups;
  float2* partial = (float2*)workspace;
  double2* AB = (double2*)((char*)workspace + pn_align_up((size_t)B * nblk * Cout * sizeof(float2), 256));
  const double M = (double)Cg * N * k * (per_sample ? 1 : B);
  PN_PROF("edgeconv_bwd_stats", stream);
  hipLaunchKernelGGL(pn_ecb_prep_kernel, dim3(pn_cdiv(Cout, 32), nblk, B), dim3(256), 0, stream, gout, yext, mean,
                     rstd, gamma, beta, N, Cout, Cg, per_sample, slope, t, partial);
  hipLaunchKernelGGL(pn_ecb_reduce_kernel, dim3(pn_cdiv(Cout, 16), B), dim3(256), 0, stream, (const float2*)partial,
                     nblk, Cout, AB);
  hipLaunchKernelGGL(pn_ecb_finish_kernel, dim3(1), dim3(256), 0, stream, (const double2*)AB, B, Cout, Cg, per_sample,
                     dense, M, gamma, dgamma, dbeta, c1c2);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

// =============================================================================================
// triplet embedding loss (src/segment_loss.py:85-123)
// =============================================================================================
// Item p: rows ia[p][0..num) (anchor / positive segment) and ib[p][0..num) (negative segment) of
// the unit-row embedding E (rows of D floats).  c[i][j] = relu(|a_i - p_j|^2 - |a_i - n_j|^2 + margin);
// loss_p = (sum_ij c - sum_i c_ii) / (#(c > 0) + 1) * w[p].  One workgroup per item; rows staged in
// LDS.  out[p] = loss_p (summed on the host side in index order by a tiny reduction below).
#define TRI_MAXNUM 32
template <int D>
__global__ __launch_bounds__(256) void pn_triplet_fwd_kernel(const float* __restrict__ E, const int64_t* __restrict__ ia,
                                                             const int64_t* __restrict__ ib, const float* __restrict__ w,
                                                             int num, float margin, float* __restrict__ item_loss,
                                                             float* __restrict__ item_scale) {
  __shared__ float P1[TRI_MAXNUM][D + 1];
  __shared__ float P2[TRI_MAXNUM][D + 1];
  __shared__ float rs[4];
  __shared__ int rc[4];
  const int p = blockIdx.x;
  for (int e = threadIdx.x; e < num * D; e += 256) {
    const int r = e / D, d = e - r * D;
    P1[r][d] = E[(size_t)ia[(size_t)p * num + r] * D + d];
    P2[r][d] = E[(size_t)ib[(size_t)p * num + r] * D + d];
  }
  __syncthreads();
  float s = 0.f;
  int cnt = 0;
  for (int e = threadIdx.x; e < num * num; e += 256) {
    const int i = e / num, j = e - i * num;
    float dp = 0.f, dn = 0.f;
    for (int d = 0; d < D; ++d) {
      const float a = P1[i][d];
      const float x = a - P1[j][d], y = a - P2[j][d];
      dp = __builtin_fmaf(x, x, dp);
      dn = __builtin_fmaf(y, y, dn);
    }
    const float c = fmaxf(dp - dn + margin, 0.f);
    cnt += c > 0.f;
    if (i != j) s += c;
  }
  s = pn_wave_sum(s);
  cnt = pn_wave_sum_i(cnt);
  if ((threadIdx.x & 63) == 0) {
    rs[threadIdx.x >> 6] = s;
    rc[threadIdx.x >> 6] = cnt;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float tot = (rs[0] + rs[1]) + (rs[2] + rs[3]);
    const float sat = (float)(rc[0] + rc[1] + rc[2] + rc[3]) + 1.f;
    const float sc = w[p] / sat;
    item_loss[p] = tot * sc;
    item_scale[p] = sc;
  }
}

// sum of the item losses in index order (one thread: P <= a few hundred)
__global__ void pn_triplet_sum_kernel(const float* __restrict__ item_loss, int P, float* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float s = 0.f;
    for (int p = 0; p < P; ++p) s += item_loss[p];
    out[0] = s;
  }
}

// Backward: gE[row] = sum over the items of g * scale_p * d c_ij / d row for every active (c > 0,
// i != j) entry:  d/d a_i = 2 (n_j - p_j),  d/d p_j = -2 (a_i - p_j),  d/d n_j = 2 (a_i - n_j)
// (a_i = P1[i], p_j = P1[j]).  A point can be sampled by several items (and by both sides of
// one): kernel 1 writes the gradient rows of every item to slots[(p * 2 + side) * num + r][D],
// kernel 2 adds the slots that name the same embedding row IN SLOT ORDER — one workgroup per
// slot, the first slot of a row does the work — and stores the row.  No atomics: bit-reproducible
// run to run (round 4).  Rows no item names keep what gE held on entry (the caller zeroes it).
template <int D>
__global__ __launch_bounds__(256) void pn_triplet_bwd_kernel(const float* __restrict__ E, const int64_t* __restrict__ ia,
                                                             const int64_t* __restrict__ ib,
                                                             const float* __restrict__ item_scale,
                                                             const float* __restrict__ gout, int num, float margin,
                                                             float* __restrict__ slots) {
  __shared__ float P1[TRI_MAXNUM][D + 1];
  __shared__ float P2[TRI_MAXNUM][D + 1];
  __shared__ unsigned int act[TRI_MAXNUM];      // act[i] bit j: c_ij > 0 and i != j
  const int p = blockIdx.x;
  for (int e = threadIdx.x; e < num * D; e += 256) {
    const int r = e / D, d = e - r * D;
    P1[r][d] = E[(size_t)ia[(size_t)p * num + r] * D + d];
    P2[r][d] = E[(size_t)ib[(size_t)p * num + r] * D + d];
  }
  if (threadIdx.x < TRI_MAXNUM) act[threadIdx.x] = 0u;
  __syncthreads();
  for (int e = threadIdx.x; e < num * num; e += 256) {
    const int i = e / num, j = e - i * num;
    float dp = 0.f, dn = 0.f;
    for (int d = 0; d < D; ++d) {
      const float a = P1[i][d];
      const float x = a - P1[j][d], y = a - P2[j][d];
      dp = __builtin_fmaf(x, x, dp);
      dn = __builtin_fmaf(y, y, dn);
    }
    if (dp - dn + margin > 0.f && i != j) atomicOr(&act[i], 1u << j);
  }
  __syncthreads();
  const float g2 = 2.f * gout[0] * item_scale[p];
  // thread -> (row r, channel d): three gradient rows per r
  for (int e = threadIdx.x; e < num * D; e += 256) {
    const int r = e / D, d = e - r * D;
    // as anchor i = r: sum_j act[r][j] * (n_j - p_j)
    float ga = 0.f;
    const unsigned int mr = act[r];
    for (int j = 0; j < num; ++j)
      if ((mr >> j) & 1u) ga += P2[j][d] - P1[j][d];
    // as positive j = r: - sum_i act[i][r] * (a_i - p_r);  as negative j = r: + sum_i act[i][r] * (a_i - n_r)
    float gp = 0.f, gn = 0.f;
    const float pr = P1[r][d], nr = P2[r][d];
    for (int i = 0; i < num; ++i)
      if ((act[i] >> r) & 1u) {
        gp -= P1[i][d] - pr;
        gn += P1[i][d] - nr;
      }
    slots[((size_t)(p * 2 + 0) * num + r) * D + d] = g2 * (ga + gp);
    slots[((size_t)(p * 2 + 1) * num + r) * D + d] = g2 * gn;
  }
}

// slot s = (p * 2 + side) * num + r names row (side ? ib : ia)[p * num + r]
__device__ static inline int64_t pn_tri_slot_row(const int64_t* __restrict__ ia, const int64_t* __restrict__ ib,
                                                 int num, int s) {
  const int r = s % num, ps = s / num;
  return (ps & 1) ? ib[(size_t)(ps >> 1) * num + r] : ia[(size_t)(ps >> 1) * num + r];
}

#define TRI_SCAN 4096     // slot ids staged per pass (16 KiB of LDS)
template <int D>
__global__ __launch_bounds__(D) void pn_triplet_combine_kernel(const float* __restrict__ slots,
                                                                const int64_t* __restrict__ ia,
                                                                const int64_t* __restrict__ ib, int num, int S,
                                                                float* __restrict__ gE) {
  __shared__ unsigned int hit[TRI_SCAN / 32];
  const int s = blockIdx.x, d = threadIdx.x;
  const int64_t row = pn_tri_slot_row(ia, ib, num, s);
  // an earlier slot with the same row owns it
  int earlier = 0;
  for (int u = d; u < s; u += D) earlier |= (pn_tri_slot_row(ia, ib, num, u) == row);
  if (__syncthreads_or(earlier)) return;
  float acc = slots[(size_t)s * D + d];
  for (int base = s + 1; base < S; base += TRI_SCAN) {
    const int cnt = S - base < TRI_SCAN ? S - base : TRI_SCAN;
    for (int u = d; u < TRI_SCAN / 32; u += D) hit[u] = 0u;
    __syncthreads();
    for (int u = d; u < cnt; u += D)
      if (pn_tri_slot_row(ia, ib, num, base + u) == row) atomicOr(&hit[u >> 5], 1u << (u & 31));
    __syncthreads();
    for (int w = 0; w < (cnt + 31) / 32; ++w) {
      unsigned int m = hit[w];                      // workgroup-uniform
      while (m) {
        const int bit = __builtin_ctz(m);
        m &= m - 1;
        acc += slots[(size_t)(base + w * 32 + bit) * D + d];
      }
    }
    __syncthreads();
  }
  gE[(size_t)row * D + d] = acc;
}

extern "C" int pn_triplet_fwd_f32(const float* E, int rows, int D, const int64_t* ia, const int64_t* ib,
                                  const float* w, int P, int num, float margin, float* item_loss,
                                  float* item_scale, float* loss, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PN_CHECK_ARG(E && ia && ib && w && item_loss && item_scale && loss, "pn_triplet_fwd_f32: null pointer");
  PN_CHECK_ARG(P > 0 && rows > 0 && num >= 1 && num <= TRI_MAXNUM, "pn_triplet_fwd_f32: P=%d num=%d (max %d)", P,
               num, TRI_MAXNUM);
  if (D != 128) {
    pn_set_error("pn_triplet_fwd_f32: embedding size %d (128 supported)", D);
    return PN_ERR_UNSUPPORTED;
  }
  PN_PROF("triplet_fwd", stream);
  hipLaunchKernelGGL(pn_triplet_fwd_kernel<128>, dim3(P), dim3(256), 0, stream, E, ia, ib, w, num, margin, item_loss,
                     item_scale);
  hipLaunchKernelGGL(pn_triplet_sum_kernel, dim3(1), dim3(64), 0, stream, (const float*)item_loss, P, loss);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

extern "C" size_t pn_triplet_bwd_workspace(int P, int num, int D) {
  return pn_align_up((size_t)P * 2 * num * D * sizeof(float), 256);
}

extern "C" int pn_triplet_bwd_f32(const float* E, int rows, int D, const int64_t* ia, const int64_t* ib,
                                  const float* item_scale, const float* gout, int P, int num, float margin,
                                  float* gE, void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PN_CHECK_ARG(E && ia && ib && item_scale && gout && gE && workspace, "pn_triplet_bwd_f32: null pointer");
  PN_CHECK_ARG(P > 0 && rows > 0 && num >= 1 && num <= TRI_MAXNUM, "pn_triplet_bwd_f32: P=%d num=%d", P, num);
  if (D != 128) {
    pn_set_error("pn_triplet_bwd_f32: embedding size %d (128 supported)", D);
    return PN_ERR_UNSUPPORTED;
  }
  PN_CHECK_ARG(workspace_bytes >= pn_triplet_bwd_workspace(P, num, D), "pn_triplet_bwd_f32: workspace too small");
  float* slots = (float*)workspace;
  PN_PROF("triplet_bwd", stream);
  hipLaunchKernelGGL(pn_triplet_bwd_kernel<128>, dim3(P), dim3(256), 0, stream, E, ia, ib, item_scale, gout, num,
                     margin, slots);
  hipLaunchKernelGGL(pn_triplet_combine_kernel<128>, dim3(P * 2 * num), dim3(128), 0, stream, (const float*)slots, ia,
                     ib, num, P * 2 * num, gE);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

// =============================================================================================
// memberships: centres . embedding^T -> weights_normalize (+ nearest-centre labels)
// =============================================================================================
// cen (B,CP,D) padded centre rows (rows >= ncl[b] ignored), emb (B,N,D), bw (B).  Per point n:
//   Wraw[c] = cen_c . emb_n                    (fp32 fma chain over the channels in order)
//   p[c]    = exp(clamp(Wraw[c] / b^2 / 2, -75, 75))   for c < ncl, 0 for padding rows
//   prob[c] = p[c] / sum_c p[c]                (src/fitting_utils.py:314-317)
//   label   = first arg-max over c < ncl of Wraw[c]      (src/mean_shift.py:176-178)
// then per row c (second kernel): m = min_n prob, s = max_n (prob - m) + eps and
//   Wn[c][n] = ncl > 1 ? (prob - m) / s : prob           (:319-324)
// Block = 64 points x 4 waves; wave w owns centre rows [w * CP/4, (w+1) * CP/4).
#define MB_D 128
template <int CP>
__global__ __launch_bounds__(256) void pn_member_fwd_kernel(const float* __restrict__ cen, const float* __restrict__ emb,
                                                            const float* __restrict__ bw, const int64_t* __restrict__ ncl,
                                                            int N, float* __restrict__ Wraw, float* __restrict__ prob,
                                                            int64_t* __restrict__ labels) {
  constexpr int RW = CP / 4;                         // rows per wave
  __shared__ float sc[CP][MB_D];                     // centre rows (wave-uniform reads)
  __shared__ float psum[4][64];
  __shared__ float pbest[4][64];
  __shared__ int pbidx[4][64];
  const int b = blockIdx.y, n0 = blockIdx.x * 64;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int nc = (int)ncl[b];
  for (int e = threadIdx.x; e < CP * MB_D; e += 256) sc[e / MB_D][e % MB_D] = cen[(size_t)b * CP * MB_D + e];
  __syncthreads();
  // lane = point: its embedding row streams through registers 4 channels at a time (a 128-byte
  // line serves 8 consecutive loads of the lane; the 4 waves of the block share the tile in L1)
  const int nl = n0 + lane < N ? n0 + lane : N - 1;
  const float4* __restrict__ er = (const float4*)(emb + ((size_t)b * N + nl) * MB_D);
  float acc[RW];
#pragma unroll
  for (int i = 0; i < RW; ++i) acc[i] = 0.f;
  for (int d4 = 0; d4 < MB_D / 4; ++d4) {
    const float4 e = er[d4];
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      const float* __restrict__ cr = &sc[wave * RW + i][4 * d4];
      float a = acc[i];
      a = __builtin_fmaf(cr[0], e.x, a);
      a = __builtin_fmaf(cr[1], e.y, a);
      a = __builtin_fmaf(cr[2], e.z, a);
      a = __builtin_fmaf(cr[3], e.w, a);
      acc[i] = a;
    }
  }
  const float bb = bw[b];
  const float b2 = bb * bb;
  float p[RW];
  float s = 0.f, best = -INFINITY;
  int bidx = 0x7fffffff;
#pragma unroll
  for (int i = 0; i < RW; ++i) {
    const int c = wave * RW + i;
    const float x = acc[i] / b2 / 2.f;
    p[i] = c < nc ? expf(fminf(fmaxf(x, -75.f), 75.f)) : 0.f;
    s += p[i];
    if (c < nc && acc[i] > best) {
      best = acc[i];
      bidx = c;
    }
  }
  psum[wave][lane] = s;
  pbest[wave][lane] = best;
  pbidx[wave][lane] = bidx;
  __syncthreads();
  const float tot = ((psum[0][lane] + psum[1][lane]) + psum[2][lane]) + psum[3][lane];
  const int n = n0 + lane;
  if (n < N) {
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      const size_t o = ((size_t)b * CP + wave * RW + i) * N + n;
      Wraw[o] = acc[i];
      prob[o] = p[i] / tot;
    }
    if (wave == 0 && labels) {
      float bv = pbest[0][lane];
      int bi = pbidx[0][lane];
#pragma unroll
      for (int w = 1; w < 4; ++w)
        if (pbest[w][lane] > bv) {       // strictly greater: ties stay with the smaller row index
          bv = pbest[w][lane];
          bi = pbidx[w][lane];
        }
      labels[(size_t)b * N + n] = bi == 0x7fffffff ? 0 : bi;
    }
  }
}

// one workgroup per (b, c) row: min / max of prob with first-index ties, then Wn.
// rowstat[b][c] = (m, s, argmin, argmax) as 4 floats (indices bit-cast).
__global__ __launch_bounds__(256) void pn_member_rows_kernel(const float* __restrict__ prob,
                                                             const int64_t* __restrict__ ncl, int CP, int N, float eps,
                                                             float* __restrict__ Wn, float4* __restrict__ rowstat) {
  __shared__ float smin[4], smax[4];
  __shared__ int simin[4], simax[4];
  const int row = blockIdx.x, b = row / CP, c = row % CP;
  const int nc = (int)ncl[b];
  const float* __restrict__ pr = prob + (size_t)row * N;
  float* __restrict__ wr = Wn + (size_t)row * N;
  if (c >= nc) {
    for (int n = threadIdx.x; n < N; n += 256) wr[n] = 0.f;
    if (threadIdx.x == 0) rowstat[row] = make_float4(0.f, 1.f, __int_as_float(0), __int_as_float(0));
    return;
  }
  float mn = INFINITY, mx = -INFINITY;
  int imn = 0x7fffffff, imx = 0x7fffffff;
  for (int n = threadIdx.x; n < N; n += 256) {
    const float v = pr[n];
    if (v < mn) { mn = v; imn = n; }
    if (v > mx) { mx = v; imx = n; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float omn = __shfl_xor(mn, o, 64), omx = __shfl_xor(mx, o, 64);
    const int oimn = __shfl_xor(imn, o, 64), oimx = __shfl_xor(imx, o, 64);
    if (omn < mn || (omn == mn && oimn < imn)) { mn = omn; imn = oimn; }
    if (omx > mx || (omx == mx && oimx < imx)) { mx = omx; imx = oimx; }
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    smin[wave] = mn; simin[wave] = imn; smax[wave] = mx; simax[wave] = imx;
  }
  __syncthreads();
  mn = smin[0]; imn = simin[0]; mx = smax[0]; imx = simax[0];
#pragma unroll
  for (int w = 1; w < 4; ++w) {
    if (smin[w] < mn || (smin[w] == mn && simin[w] < imn)) { mn = smin[w]; imn = simin[w]; }
    if (smax[w] > mx || (smax[w] == mx && simax[w] < imx)) { mx = smax[w]; imx = simax[w]; }
  }
  const float s = (mx - mn) + eps;
  if (threadIdx.x == 0) rowstat[row] = make_float4(mn, s, __int_as_float(imn), __int_as_float(imx));
  if (nc > 1) {
    for (int n = threadIdx.x; n < N; n += 256) wr[n] = (pr[n] - mn) / s;
  } else {
    for (int n = threadIdx.x; n < N; n += 256) wr[n] = pr[n];
  }
}

// Backward, rows: R0 = sum_n g, R1 = sum_n g * (prob - m)  ->  rowgrad[b][c] = (g_den, g_m)
//   g_den = -R1 / s^2 (goes to the arg-max column),  g_m = -(R0 / s + g_den) (to the arg-min column)
__global__ __launch_bounds__(256) void pn_member_bwd_rows_kernel(const float* __restrict__ gWn,
                                                                 const float* __restrict__ prob,
                                                                 const float4* __restrict__ rowstat,
                                                                 const int64_t* __restrict__ ncl, int CP, int N,
                                                                 float2* __restrict__ rowgrad) {
  __shared__ float r0s[4], r1s[4];
  const int row = blockIdx.x, b = row / CP, c = row % CP;
  const int nc = (int)ncl[b];
  if (c >= nc || nc <= 1) {
    if (threadIdx.x == 0) rowgrad[row] = make_float2(0.f, 0.f);
    return;
  }
  const float4 st = rowstat[row];
  const float* __restrict__ g = gWn + (size_t)row * N;
  const float* __restrict__ pr = prob + (size_t)row * N;
  float r0 = 0.f, r1 = 0.f;
  for (int n = threadIdx.x; n < N; n += 256) {
    const float gv = g[n];
    r0 += gv;
    r1 = __builtin_fmaf(gv, pr[n] - st.x, r1);
  }
  r0 = pn_wave_sum(r0);
  r1 = pn_wave_sum(r1);
  if ((threadIdx.x & 63) == 0) {
    r0s[threadIdx.x >> 6] = r0;
    r1s[threadIdx.x >> 6] = r1;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float R0 = (r0s[0] + r0s[1]) + (r0s[2] + r0s[3]), R1 = (r1s[0] + r1s[1]) + (r1s[2] + r1s[3]);
    const float gden = -R1 / (st.y * st.y);
    rowgrad[row] = make_float2(gden, -(R0 / st.y + gden));
  }
}

// Backward, points: thread per point, loop over the centre rows.
template <int CP>
__global__ __launch_bounds__(256) void pn_member_bwd_points_kernel(
    const float* __restrict__ gWn, const float* __restrict__ Wraw, const float* __restrict__ prob,
    const float4* __restrict__ rowstat, const float2* __restrict__ rowgrad, const float* __restrict__ bw,
    const int64_t* __restrict__ ncl, int N, float* __restrict__ gWraw) {
  __shared__ float4 sst[CP];
  __shared__ float2 sgr[CP];
  const int b = blockIdx.y, n = blockIdx.x * 256 + threadIdx.x;
  const int nc = (int)ncl[b];
  if (threadIdx.x < CP) {
    sst[threadIdx.x] = rowstat[(size_t)b * CP + threadIdx.x];
    sgr[threadIdx.x] = rowgrad[(size_t)b * CP + threadIdx.x];
  }
  __syncthreads();
  if (n >= N) return;
  const float bb = bw[b];
  const float b2 = bb * bb;
  float gp[CP];
  float dot = 0.f, tot = 0.f;
#pragma unroll
  for (int c = 0; c < CP; ++c) {
    gp[c] = 0.f;
    if (c < nc) {
      const size_t o = ((size_t)b * CP + c) * N + n;
      float g = gWn[o];
      if (nc > 1) {
        g = g / sst[c].y;
        if (n == __float_as_int(sst[c].w)) g += sgr[c].x;
        if (n == __float_as_int(sst[c].z)) g += sgr[c].y;
      }
      gp[c] = g;
      dot = __builtin_fmaf(g, prob[o], dot);
      const float x = Wraw[o] / b2 / 2.f;
      tot += expf(fminf(fmaxf(x, -75.f), 75.f));
    }
  }
#pragma unroll
  for (int c = 0; c < CP; ++c) {
    const size_t o = ((size_t)b * CP + c) * N + n;
    float out = 0.f;
    if (c < nc) {
      const float x = Wraw[o] / b2 / 2.f;
      const bool inside = x >= -75.f && x <= 75.f;          // torch.clamp passes the gradient inclusively
      // d prob / d p: (g_c - sum g prob) / tot;  d p / d x = p (inside the clamp);  p / tot = prob
      out = inside ? (gp[c] - dot) * prob[o] / b2 / 2.f : 0.f;
    }
    gWraw[o] = out;
  }
  (void)tot;
}

extern "C" int pn_membership_fwd_f32(const float* cen, const float* emb, const float* bw, const int64_t* ncl, int B,
                                     int CP, int N, int D, float eps, float* Wraw, float* prob, float* Wn,
                                     float* rowstat, int64_t* labels, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PN_CHECK_ARG(cen && emb && bw && ncl && Wraw && prob && Wn && rowstat, "pn_membership_fwd_f32: null pointer");
  PN_CHECK_ARG(B > 0 && N > 0, "pn_membership_fwd_f32: B=%d N=%d", B, N);
  if (D != MB_D || (CP != 16 && CP != 32 && CP != 64)) {
    pn_set_error("pn_membership_fwd_f32: D=%d CP=%d (D = 128 and CP in {16, 32, 64} supported)", D, CP);
    return PN_ERR_UNSUPPORTED;
  }
  PN_PROF("membership_fwd", stream);
  dim3 grid(pn_cdiv(N, 64), B);
#define MB_F(C)                                                                                              \
  hipLaunchKernelGGL(pn_member_fwd_kernel<C>, grid, dim3(256), 0, stream, cen, emb, bw, ncl, N, Wraw, prob, \
                     labels)
  if (CP == 16) MB_F(16);
  else if (CP == 32) MB_F(32);
  else MB_F(64);
#undef MB_F
  hipLaunchKernelGGL(pn_member_rows_kernel, dim3(B * CP), dim3(256), 0, stream, (const float*)prob, ncl, CP, N, eps,
                     Wn, (float4*)rowstat);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

extern "C" int pn_membership_bwd_f32(const float* gWn, const float* Wraw, const float* prob, const float* rowstat,
                                     const float* bw, const int64_t* ncl, int B, int CP, int N, float* rowgrad,
                                     float* gWraw, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PN_CHECK_ARG(gWn && Wraw && prob && rowstat && bw && ncl && rowgrad && gWraw, "pn_membership_bwd_f32: null pointer");
  if (CP != 16 && CP != 32 && CP != 64) {
    pn_set_error("pn_membership_bwd_f32: CP=%d (16, 32, 64 supported)", CP);
    return PN_ERR_UNSUPPORTED;
  }
  PN_PROF("membership_bwd", stream);
  hipLaunchKernelGGL(pn_member_bwd_rows_kernel, dim3(B * CP), dim3(256), 0, stream, gWn, prob, (const float4*)rowstat,
                     ncl, CP, N, (float2*)rowgrad);
  dim3 grid(pn_cdiv(N, 256), B);
#define MB_B(C)                                                                                          \
  hipLaunchKernelGGL(pn_member_bwd_points_kernel<C>, grid, dim3(256), 0, stream, gWn, Wraw, prob,        \
                     (const float4*)rowstat, (const float2*)rowgrad, bw, ncl, N, gWraw)
  if (CP == 16) MB_B(16);
  else if (CP == 32) MB_B(32);
  else MB_B(64);
#undef MB_B
  PN_CHECK_LAUNCH();
  return PN_OK;
}

// =============================================================================================
// per-channel affine map + activation on (B,C,N): y = act(x * scale[c] + shift[c])
// =============================================================================================
// act: 0 none, 1 ReLU, 2 LeakyReLU(slope).  Backward: gx = gy * scale[c] * act'(y).
__global__ void pn_affine_act_fwd_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                         const float* __restrict__ shift, long long total, int C, int N, int act,
                                         float slope, float* __restrict__ y) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int c = (int)((e / N) % C);
  float v = x[e] * scale[c] + shift[c];
  if (act == 1) v = fmaxf(v, 0.f);
  else if (act == 2) v = v > 0.f ? v : v * slope;
  y[e] = v;
}

__global__ void pn_affine_act_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ y,
                                         const float* __restrict__ scale, long long total, int C, int N, int act,
                                         float slope, float* __restrict__ gx) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int c = (int)((e / N) % C);
  float g = gy[e] * scale[c];
  if (act == 1) g = y[e] > 0.f ? g : 0.f;
  else if (act == 2) g = y[e] > 0.f ? g : g * slope;
  gx[e] = g;
}

extern "C" int pn_affine_act_fwd_f32(const float* x, const float* scale, const float* shift, int B, int C, int N,
                                     int act, float slope, float* y, void* stream) {
  PN_CHECK_ARG(x && scale && shift && y && B > 0 && C > 0 && N > 0 && act >= 0 && act <= 2,
               "pn_affine_act_fwd_f32: bad arguments");
  const long long total = (long long)B * C * N;
  hipLaunchKernelGGL(pn_affine_act_fwd_kernel, dim3(pn_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, scale,
                     shift, total, C, N, act, slope, y);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

extern "C" int pn_affine_act_bwd_f32(const float* gy, const float* y, const float* scale, int B, int C, int N, int act,
                                     float slope, float* gx, void* stream) {
  PN_CHECK_ARG(gy && y && scale && gx && B > 0 && C > 0 && N > 0 && act >= 0 && act <= 2,
               "pn_affine_act_bwd_f32: bad arguments");
  const long long total = (long long)B * C * N;
  hipLaunchKernelGGL(pn_affine_act_bwd_kernel, dim3(pn_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, gy, y,
                     scale, total, C, N, act, slope, gx);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

// =============================================================================================
// Standardisation of the spline segments (src/fitting_utils.py:512-553, standardize_point_torch) for S segments: the
// two parts that are NOT arithmetic on the points —
//   select: the confident points (w > 0.8; fewer than 400 of them: the kf largest memberships instead — round 5 ran a
//           topk over half of every row for this, 0.084 ms, plus a scatter and a where), as a byte mask;
//   scale : extent | max - min | of the weighted rotated SELECTED points per axis and the division by it.
// Weighted mean, centring, covariance and the rotation stay the tensor expressions (and rocBLAS products) of round 5
// ON PURPOSE: LAPACK's geev, which the reference uses for the minor axis, returns an eigenvector whose SIGN flips
// with the last bits of the covariance on the evidence boxes (tools/probes/std_sign_probe.py, profiles/r06_std_sign_probe.txt:
// the matrix a fully fused kernel produced — fp64 sums, 3-7 ulp away in the off-diagonals — turns the fixture's frame by
// 180 degrees), so the covariance keeps the bits it had.
// One 256-thread workgroup per segment; max / min / select are exact operations: results are bit-identical to the
// tensor-library form.
// =============================================================================================
#define STD_T 256

// block-wide sum of small non-negative counts: wave sums by shuffle, four partials through LDS (two barriers)
__device__ static inline int std_block_count(int v, int* shi) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) shi[threadIdx.x >> 6] = v;
  __syncthreads();
  return shi[0] + shi[1] + shi[2] + shi[3];
}

// sel (S,n) bytes.  kf: size of the fallback selection (n / 4 or n / 2).  Ties at the kf-th largest membership go
// to the smaller index (torch.topk leaves them unspecified).  The row's order-preserving integer images live in
// registers (STD_E per thread, n <= 256 STD_E; longer rows re-read them from memory): a round of the bisection is
// STD_E compares and one block count.
#define STD_E 40
__global__ __launch_bounds__(STD_T) void pn_std_select_kernel(const float* __restrict__ w, int n, int kf,
                                                              unsigned char* __restrict__ sel) {
  __shared__ int shi[STD_T];
  const int s = blockIdx.x, t = threadIdx.x;
  const float* __restrict__ ws = w + (size_t)s * n;
  unsigned char* __restrict__ ss = sel + (size_t)s * n;
  const bool in_regs = n <= STD_T * STD_E;
  uint32_t v[STD_E];
  int c = 0;
#pragma unroll
  for (int e = 0; e < STD_E; ++e) {
    const int i = e * STD_T + t;
    v[e] = 0u;                                   // (below every real image: never counted)
    if (in_regs && i < n) {
      const float x = ws[i];
      v[e] = pn_f2ord(x);
      c += x > 0.8f ? 1 : 0;
    }
  }
  if (!in_regs)
    for (int i = t; i < n; i += STD_T) c += ws[i] > 0.8f ? 1 : 0;
  const int cnt = std_block_count(c, shi);
  if (cnt >= 400) {
    for (int i = t; i < n; i += STD_T) ss[i] = ws[i] > 0.8f ? 1 : 0;
    return;
  }
  // the kf-th largest membership: largest T with #(ord(w) >= T) >= kf, bit by bit
  uint32_t T = 0;
  for (int bit = 31; bit >= 0; --bit) {
    const uint32_t cand = T | (1u << bit);
    int cc = 0;
    if (in_regs) {
#pragma unroll
      for (int e = 0; e < STD_E; ++e) cc += v[e] >= cand ? 1 : 0;
    } else {
      for (int i = t; i < n; i += STD_T) cc += pn_f2ord(ws[i]) >= cand ? 1 : 0;
    }
    if (std_block_count(cc, shi) >= kf) T = cand;
  }
  int ab = 0;
  for (int i = t; i < n; i += STD_T) ab += pn_f2ord(ws[i]) > T ? 1 : 0;
  int need_eq = kf - std_block_count(ab, shi);       // ties at T: the first need_eq in index order
  for (int base = 0; base < n; base += STD_T) {
    const int i = base + t;
    const uint32_t o = i < n ? pn_f2ord(ws[i]) : 0u;
    const int tie = (i < n && o == T) ? 1 : 0;
    // inclusive rank of this tie among the ties of the chunk (index order = thread order)
    __syncthreads();
    shi[t] = tie;
    __syncthreads();
    for (int d = 1; d < STD_T; d <<= 1) {
      const int u = t >= d ? shi[t - d] : 0;
      __syncthreads();
      shi[t] += u;
      __syncthreads();
    }
    const int incl = shi[t], total = shi[STD_T - 1];
    if (i < n) ss[i] = (o > T || (tie && incl - 1 < need_eq)) ? 1 : 0;
    need_eq -= total < need_eq ? total : need_eq;
  }
}

// Pr (S,n,3) rotated centred points -> std (S,3) = | max - min | over the selected points of Pr * w per axis,
// pts (S,n,3) = Pr / (std + eps)
__global__ __launch_bounds__(STD_T) void pn_std_scale_kernel(const float* __restrict__ Pr, const float* __restrict__ w,
                                                             const unsigned char* __restrict__ sel, int n, float eps,
                                                             float* __restrict__ pts, float* __restrict__ stdv) {
  __shared__ float shm[6][STD_T];
  const int s = blockIdx.x, t = threadIdx.x;
  const float* __restrict__ ws = w + (size_t)s * n;
  const float* __restrict__ Ps = Pr + (size_t)s * n * 3;
  const unsigned char* __restrict__ ss = sel + (size_t)s * n;
  float* __restrict__ os = pts + (size_t)s * n * 3;
  float hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
  float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
  for (int i = t; i < n; i += STD_T) {
    if (ss[i]) {
      const float wi = ws[i];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float v = Ps[3 * i + a] * wi;
        hi[a] = fmaxf(hi[a], v);
        lo[a] = fminf(lo[a], v);
      }
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    shm[a][t] = hi[a];
    shm[3 + a][t] = lo[a];
  }
  __syncthreads();
  for (int o = STD_T / 2; o > 0; o >>= 1) {
    if (t < o) {
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        shm[a][t] = fmaxf(shm[a][t], shm[a][t + o]);
        shm[3 + a][t] = fminf(shm[3 + a][t], shm[3 + a][t + o]);
      }
    }
    __syncthreads();
  }
  float sd[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) sd[a] = fabsf(shm[a][0] - shm[3 + a][0]);
  if (t < 3) stdv[3 * s + t] = sd[t];
  for (int i = t; i < n; i += STD_T) {
#pragma unroll
    for (int a = 0; a < 3; ++a) os[3 * i + a] = Ps[3 * i + a] / (sd[a] + eps);
  }
}

extern "C" int pn_standardize_select_f32(const float* w, int S, int n, int kf, unsigned char* sel, void* stream) {
  PN_CHECK_ARG(w && sel && S > 0 && n > 0 && kf >= 1 && kf <= n, "pn_standardize_select_f32: bad arguments (S=%d n=%d kf=%d)",
               S, n, kf);
  PN_PROF("standardize", (hipStream_t)stream);
  hipLaunchKernelGGL(pn_std_select_kernel, dim3(S), dim3(STD_T), 0, (hipStream_t)stream, w, n, kf, sel);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

extern "C" int pn_standardize_scale_f32(const float* Pr, const float* w, const unsigned char* sel, int S, int n, float eps,
                                        float* pts, float* stdv, void* stream) {
  PN_CHECK_ARG(Pr && w && sel && pts && stdv && S > 0 && n > 0, "pn_standardize_scale_f32: bad arguments");
  PN_PROF("standardize", (hipStream_t)stream);
  hipLaunchKernelGGL(pn_std_scale_kernel, dim3(S), dim3(STD_T), 0, (hipStream_t)stream, Pr, w, sel, n, eps, pts, stdv);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

// =============================================================================================
// Adam on ONE flat parameter buffer (train_parsenet.py:96, train_parsenet_e2e.py:88, train_open_splines.py:81:
// optim.Adam(model.parameters(), lr) with torch's defaults).  torch's rule, operation by operation:
//   m <- m + (g - m) (1 - beta1);  v <- beta2 v + (1 - beta2) g g
//   p <- p - (lr / (1 - beta1^t)) m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
// All parameters of the model, their gradients (dp.FlatGradBucket) and both moments are contiguous: one launch,
// 16-byte lanes, no per-tensor grouping on the host.
// =============================================================================================
__global__ __launch_bounds__(256) void pn_adam_flat_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                           float* __restrict__ m, float* __restrict__ v, long long n,
                                                           float one_minus_b1, float b2, float one_minus_b2,
                                                           float step_size, float bc2_sqrt, float eps) {
  const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= n) return;
  if (i + 4 <= n) {
    float4 P = *reinterpret_cast<float4*>(p + i), M = *reinterpret_cast<float4*>(m + i);
    float4 V = *reinterpret_cast<float4*>(v + i);
    const float4 G = *reinterpret_cast<const float4*>(g + i);
#define PN_ADAM1(PP, GG, MM, VV)                          \
  {                                                       \
    MM = MM + (GG - MM) * one_minus_b1;                   \
    VV = b2 * VV + one_minus_b2 * GG * GG;                \
    PP = PP - step_size * (MM / (sqrtf(VV) / bc2_sqrt + eps)); \
  }
    PN_ADAM1(P.x, G.x, M.x, V.x);
    PN_ADAM1(P.y, G.y, M.y, V.y);
    PN_ADAM1(P.z, G.z, M.z, V.z);
    PN_ADAM1(P.w, G.w, M.w, V.w);
    *reinterpret_cast<float4*>(p + i) = P;
    *reinterpret_cast<float4*>(m + i) = M;
    *reinterpret_cast<float4*>(v + i) = V;
  } else {
    for (long long j = i; j < n; ++j) {
      float P = p[j], M = m[j], V = v[j];
      const float G = g[j];
      PN_ADAM1(P, G, M, V);
      p[j] = P, m[j] = M, v[j] = V;
    }
  }
#undef PN_ADAM1
}

extern "C" int pn_adam_flat_f32(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1,
                                float beta2, float eps, int step, void* stream) {
  PN_CHECK_ARG(p && g && m && v && n > 0 && step >= 1, "pn_adam_flat_f32: bad arguments");
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(pn_adam_flat_kernel, dim3((unsigned)pn_cdiv(pn_cdiv(n, 4), 256)), dim3(256), 0, (hipStream_t)stream, p,
                     g, m, v, n, 1.0f - beta1, beta2, 1.0f - beta2, (float)((double)lr / bc1), (float)sqrt(bc2), eps);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

// =============================================================================================
// The gradients autograd hands over (one tensor per parameter) into the flat bucket of dp.FlatGradBucket: ONE launch
// for up to 64 tensors — the pointers travel in the kernel arguments — instead of one device-to-device copy per
// parameter (torch._foreach_copy_ issues 46 of them for the segmentation network: 134 copy launches per cfg5 step
// in profiles/r05_cfg5_profile_only_kernel_stats.csv).
// =============================================================================================
struct PnGatherTable {
  const float* src[64];
  long long off[64];
  long long n[64];
};

__global__ __launch_bounds__(256) void pn_gather_flat_kernel(PnGatherTable T, float* __restrict__ flat) {
  const int e = blockIdx.y;
  const float* __restrict__ s = T.src[e];
  float* __restrict__ d = flat + T.off[e];
  const long long n = T.n[e];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    d[i] = s[i];
}

// srcs / offs / ns: HOST arrays of ``count`` entries: tensor e = ns[e] floats at srcs[e] -> flat + offs[e]
extern "C" int pn_gather_flat_f32(const float* const* srcs, const long long* offs, const long long* ns, int count,
                                  float* flat, void* stream) {
  PN_CHECK_ARG(srcs && offs && ns && flat && count >= 0, "pn_gather_flat_f32: bad arguments");
  for (int e0 = 0; e0 < count; e0 += 64) {
    PnGatherTable T;
    const int m = count - e0 < 64 ? count - e0 : 64;
    long long nmax = 0;
    for (int e = 0; e < m; ++e) {
      PN_CHECK_ARG(srcs[e0 + e] && ns[e0 + e] >= 0 && offs[e0 + e] >= 0, "pn_gather_flat_f32: bad entry %d", e0 + e);
      T.src[e] = srcs[e0 + e];
      T.off[e] = offs[e0 + e];
      T.n[e] = ns[e0 + e];
      if (T.n[e] > nmax) nmax = T.n[e];
    }
    int bx = pn_cdiv(nmax, 256 * 16);
    bx = bx < 1 ? 1 : (bx > 64 ? 64 : bx);
    hipLaunchKernelGGL(pn_gather_flat_kernel, dim3(bx, m), dim3(256), 0, (hipStream_t)stream, T, flat);
  }
  PN_CHECK_LAUNCH();
  return PN_OK;
}

// =============================================================================================
// non-maximum suppression of the shifted points (src/mean_shift.py:139-179), device side
// =============================================================================================
// counts[b][membership[b][n]] += 1 (integer atomics: order-independent)
__global__ void pn_nms_count_kernel(const int64_t* __restrict__ membership, int N, long long total,
                                    int* __restrict__ counts) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int b = (int)(e / N);
  atomicAdd(&counts[(size_t)b * N + membership[e]], 1);
}

// ascending indices of the entries with flag[b][n] > 0, one workgroup of 1024 threads per item:
// out[b][0..cnt) = indices (at most cap are written), out[b][cnt..cap) = 0, count[b] = cnt (uncapped).
__global__ __launch_bounds__(1024) void pn_nms_compact_kernel(const int* __restrict__ flag, int N, int cap,
                                                              int64_t* __restrict__ out, int64_t* __restrict__ count) {
  __shared__ int wsum[16];
  __shared__ int carry;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int n0 = 0; n0 < N; n0 += 1024) {
    const int n = n0 + tid;
    const bool on = n < N && flag[(size_t)b * N + n] > 0;
    const unsigned long long m = __ballot(on);
    const int before = pn_mbcnt(m);
    if (lane == 0) wsum[wave] = __popcll(m);
    __syncthreads();
    int base = carry;
    for (int w = 0; w < wave; ++w) base += wsum[w];
    const int pos = base + before;
    if (on && pos < cap) out[(size_t)b * cap + pos] = n;
    __syncthreads();
    if (tid == 0) {
      int t = 0;
      for (int w = 0; w < 16; ++w) t += wsum[w];
      carry += t;
    }
    __syncthreads();
  }
  const int cnt = carry;
  for (int p = cnt + tid; p < cap; p += 1024) out[(size_t)b * cap + p] = 0;
  if (tid == 0) count[b] = cnt;
}

// vote of every occupied centre u (row of G = Cu . Cu^T, U x U): the first v maximising
// [2 - 2 G[u][v] < bw] * cnt[uq[v]]  (distance < b, not b^2, like the reference) -> hits[uq[v]] = 1.
// One wave per row; rows >= nocc[b] and columns >= nocc[b] are padding.
__global__ __launch_bounds__(256) void pn_nms_vote_kernel(const float* __restrict__ G, const int64_t* __restrict__ uq,
                                                          const int64_t* __restrict__ nocc,
                                                          const int* __restrict__ counts, const float* __restrict__ bw,
                                                          int N, int U, int* __restrict__ hits) {
  const int b = blockIdx.y, lane = threadIdx.x & 63;
  const int u = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int no = (int)min((long long)nocc[b], (long long)U);
  if (u >= no) return;
  const float bb = bw[b];
  const float* __restrict__ g = G + ((size_t)b * U + u) * U;
  const int64_t* __restrict__ uqb = uq + (size_t)b * U;
  const int* __restrict__ cb = counts + (size_t)b * N;
  float best = -1.f;
  int bi = 0x7fffffff;
  for (int v = lane; v < no; v += 64) {
    const float d = 2.0f - 2.0f * g[v];
    const float sc = d < bb ? (float)cb[uqb[v]] : 0.f;
    if (sc > best) {       // ascending v per lane: the first maximum of the lane is kept
      best = sc;
      bi = v;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ob > best || (ob == best && oi < bi)) {
      best = ob;
      bi = oi;
    }
  }
  if (lane == 0) hits[(size_t)b * N + uqb[bi]] = 1;
}

extern "C" int pn_nms_occupied_f32(const int64_t* membership, int B, int N, int U, int* counts, int64_t* uq,
                                   int64_t* nocc, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PN_CHECK_ARG(membership && counts && uq && nocc && B > 0 && N > 0 && U > 0, "pn_nms_occupied_f32: bad arguments");
  PN_CHECK_HIP(hipMemsetAsync(counts, 0, (size_t)B * N * sizeof(int), stream));
  PN_PROF("nms_occupied", stream);
  const long long total = (long long)B * N;
  hipLaunchKernelGGL(pn_nms_count_kernel, dim3(pn_cdiv(total, 256)), dim3(256), 0, stream, membership, N, total, counts);
  hipLaunchKernelGGL(pn_nms_compact_kernel, dim3(B), dim3(1024), 0, stream, (const int*)counts, N, U, uq, nocc);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

extern "C" int pn_nms_vote_f32(const float* G, const int64_t* uq, const int64_t* nocc, const int* counts,
                               const float* bw, int B, int N, int U, int cmax, int* hits, int64_t* cid,
                               int64_t* ncl, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PN_CHECK_ARG(G && uq && nocc && counts && bw && hits && cid && ncl && B > 0 && N > 0 && U > 0 && cmax > 0,
               "pn_nms_vote_f32: bad arguments");
  PN_CHECK_HIP(hipMemsetAsync(hits, 0, (size_t)B * N * sizeof(int), stream));
  PN_PROF("nms_vote", stream);
  hipLaunchKernelGGL(pn_nms_vote_kernel, dim3(pn_cdiv(U, 4), B), dim3(256), 0, stream, G, uq, nocc, counts, bw, N, U,
                     hits);
  hipLaunchKernelGGL(pn_nms_compact_kernel, dim3(B), dim3(1024), 0, stream, (const int*)hits, N, cmax, cid, ncl);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

// =============================================================================================
// SplineNet head: max over the points of  act(x * scale[c] + shift[c]) * w[s][n]
// =============================================================================================
// src/model.py:160-170 under eval(): conv5 -> bn5 -> LeakyReLU, "x *= weights", adaptive max pool —
// for a FROZEN network (no gradient to x): out[s][c] = max_n, idx[s][c] = its first arg-max,
// val[s][c] = the activation there (the factor of w in the product).  One workgroup per (s, c) row.
__global__ __launch_bounds__(256) void pn_wmax_fwd_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const float* __restrict__ w,
                                                          int C, int N, int act, float slope, float* __restrict__ out,
                                                          int* __restrict__ idx, float* __restrict__ val) {
  __shared__ float sm[4], sv[4];
  __shared__ int si[4];
  const int row = blockIdx.x, s = row / C, c = row - s * C;
  const float sc = scale[c], sh = shift[c];
  const float* __restrict__ xr = x + (size_t)row * N;
  const float* __restrict__ wr = w + (size_t)s * N;
  float best = -INFINITY, bv = 0.f;
  int bi = 0x7fffffff;
  for (int n = threadIdx.x; n < N; n += 256) {
    float v = xr[n] * sc + sh;
    if (act == 1) v = fmaxf(v, 0.f);
    else if (act == 2) v = v > 0.f ? v : v * slope;
    const float p = v * wr[n];
    if (p > best) {
      best = p;
      bi = n;
      bv = v;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64), ov = __shfl_xor(bv, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ob > best || (ob == best && oi < bi)) {
      best = ob;
      bi = oi;
      bv = ov;
    }
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    sm[wave] = best;
    si[wave] = bi;
    sv[wave] = bv;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < 4; ++k)
      if (sm[k] > best || (sm[k] == best && si[k] < bi)) {
        best = sm[k];
        bi = si[k];
        bv = sv[k];
      }
    out[row] = best;
    idx[row] = bi;
    val[row] = bv;
  }
}

// gw[s][n] = sum over the channels whose arg-max is n of g[s][c] * val[s][c], channels in order.
// One wave per segment walks the channels 64 at a time.  Every lane ORs its bit into a 64-bit mask of its
// point (LDS); the lowest lane of a mask adds the terms of the mask's lanes in lane = channel order to the
// point's accumulator — the groups of a chunk side by side, each in the serial order.  (Round 3 resolved the
// groups one after the other through ballots: up to 64 rounds per chunk, 115 us for 1 024 channels.)
__global__ __launch_bounds__(64) void pn_wmax_bwd_kernel(const float* __restrict__ g, const int* __restrict__ idx,
                                                         const float* __restrict__ val, int C, int N,
                                                         float* __restrict__ gw) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long wm_mask[];   // [N] masks, [N] sums, [64] terms
  float* acc = reinterpret_cast<float*>(wm_mask + N);
  float* term = acc + N;
  const int s = blockIdx.x, lane = threadIdx.x;
  for (int n = lane; n < N; n += 64) {
    acc[n] = 0.f;
    wm_mask[n] = 0ull;
  }
  int i_n = -1;
  float v_n = 0.f;
  if (lane < C) {
    i_n = idx[(size_t)s * C + lane];
    v_n = g[(size_t)s * C + lane] * val[(size_t)s * C + lane];
  }
  __syncthreads();
  for (int base = 0; base < C; base += 64) {
    int i = i_n;
    const float v = v_n;
    if (i < 0 || i >= N) i = -1;
    const int cn = base + 64 + lane;          // the next chunk's loads under this chunk's work
    i_n = -1;
    v_n = 0.f;
    if (cn < C) {
      i_n = idx[(size_t)s * C + cn];
      v_n = g[(size_t)s * C + cn] * val[(size_t)s * C + cn];
    }
    term[lane] = v;
    if (i >= 0) atomicOr(&wm_mask[i], 1ull << lane);
    __syncthreads();
    unsigned long long m = i >= 0 ? wm_mask[i] : 0ull;
    const bool first = i >= 0 && (m & ((1ull << lane) - 1ull)) == 0ull;
    __syncthreads();                          // every lane has its mask: the first lanes may clear them
    if (first) {
      wm_mask[i] = 0ull;
      float sum = acc[i];
      while (m) {                             // ascending lanes = ascending channels: the serial order
        sum += term[__ffsll((long long)m) - 1];
        m &= m - 1ull;
      }
      acc[i] = sum;
    }
    __syncthreads();
  }
  for (int n = lane; n < N; n += 64) gw[(size_t)s * N + n] = acc[n];
}

extern "C" int pn_weighted_max_fwd_f32(const float* x, const float* scale, const float* shift, const float* w, int S,
                                       int C, int N, int act, float slope, float* out, int* idx, float* val,
                                       void* stream) {
  PN_CHECK_ARG(x && scale && shift && w && out && idx && val && S > 0 && C > 0 && N > 0 && act >= 0 && act <= 2,
               "pn_weighted_max_fwd_f32: bad arguments");
  PN_PROF("weighted_max_fwd", (hipStream_t)stream);
  hipLaunchKernelGGL(pn_wmax_fwd_kernel, dim3(S * C), dim3(256), 0, (hipStream_t)stream, x, scale, shift, w, C, N, act,
                     slope, out, idx, val);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

extern "C" int pn_weighted_max_bwd_f32(const float* g, const int* idx, const float* val, int S, int C, int N,
                                       float* gw, void* stream) {
  PN_CHECK_ARG(g && idx && val && gw && S > 0 && C > 0 && N > 0, "pn_weighted_max_bwd_f32: bad arguments");
  const size_t smem = pn_align_up((size_t)N * 12 + 64 * sizeof(float), 16);
  if (smem > 160 * 1024 - 256) {
    pn_set_error("pn_weighted_max_bwd_f32: N=%d exceeds the LDS accumulator (13 600 points)", N);
    return PN_ERR_UNSUPPORTED;
  }
  static unsigned attr_devs = 0;
  if (pn_first_on_device(&attr_devs)) {
    PN_CHECK_HIP(hipFuncSetAttribute((const void*)pn_wmax_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     160 * 1024 - 256));
  }
  PN_PROF("weighted_max_bwd", (hipStream_t)stream);
  hipLaunchKernelGGL(pn_wmax_bwd_kernel, dim3(S), dim3(64), smem, (hipStream_t)stream, g, idx, val, C, N, gw);
  PN_CHECK_LAUNCH();
  return PN_OK;
}
