// Edge features and fused edge convolution for gfx950.
//
// (1) pn_edge_feature_*: the API form of get_graph_feature
//       src/model.py:25-53, src/PointNet.py:72-103, :106-140
//     feat[b,n,kk,:] = cat(x[:,idx[b,n,kk]] - x[:,n], x[:,n])  laid out (B,N,k,2C) — the
//     memory the reference's permuted (B,2C,N,k) view aliases.  Pure HBM streaming: rows of
//     the point-major copy of x are read with 16-byte lanes, 512-byte rows are written.
//
// (2) pn_edgeconv_*: what the encoders actually run,
//       conv1x1(no bias) -> Group/BatchNorm -> LeakyReLU -> max over k
//       src/PointNet.py:157-165,180-191,203-214; src/model.py:75-86,146-159
//     restructured for the hardware instead of materialising the (B,2C,N,k) tensor:
//       W [xj - xi ; xi] = Wa xj + (Wb - Wa) xi = P[j] + Q[i]
//     so the 1x1 convolution collapses to ONE dense (B*N, C) x (C, 2*Cout) GEMM on points
//     (80x fewer FLOPs than on edges), and the edge stage is a gather-reduce over rows of P:
//     per (point, channel) the extreme of y = P[j]+Q[i] over the k neighbours (max if the
//     norm scale gamma >= 0, min otherwise: norm and LeakyReLU are monotone per channel),
//     its arg, the sum over k (needed by the backward), and the group statistics
//     sum(y), sum(y^2) accumulated in fp64.  The backward applies the full Group/BatchNorm
//     gradient analytically, edge by edge (see pn_edgeconv_bwd_f32).
#include "common.h"

// ------------------------------------------------------------------------------------
// transpose (B,R,C) -> (B,C,R)
// ------------------------------------------------------------------------------------
__global__ void pn_transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int R,
                                    int C) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const float* ib = in + (size_t)b * R * C;
  float* ob = out + (size_t)b * R * C;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    if (r < R && c < C) tile[i][tx] = ib[(size_t)r * C + c];
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;
    if (r < R && c < C) ob[(size_t)c * R + r] = tile[tx][i];
  }
}

extern "C" int pn_transpose_f32(const float* in, float* out, int B, int R, int C, void* stream) {
  PN_CHECK_ARG(in && out && B > 0 && R > 0 && C > 0, "pn_transpose_f32: bad arguments");
  dim3 grid(pn_cdiv(C, 32), pn_cdiv(R, 32), B);
  hipLaunchKernelGGL(pn_transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, out, R, C);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

// ------------------------------------------------------------------------------------
// (1) edge features, API form
// ------------------------------------------------------------------------------------
// vector path: C % 4 == 0.  LPR = C/4 lanes per row; a wave covers 64/LPR edges per step.
__global__ __launch_bounds__(256) void pn_edge_feature_vec_kernel(
    const float* __restrict__ xt, const int64_t* __restrict__ idx, int N, int k, int C,
    float* __restrict__ feat) {
  const int b = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + wave;
  if (n >= N) return;
  const int C4 = C >> 2;
  const float4* __restrict__ xb = reinterpret_cast<const float4*>(xt + (size_t)b * N * C);
  const int64_t* __restrict__ ib = idx + ((size_t)b * N + n) * k;
  float4* __restrict__ fo = reinterpret_cast<float4*>(feat + ((size_t)b * N + n) * k * 2 * C);
  if (C4 <= 64) {
    const int epw = 64 / C4;            // edges per wave step
    const int e = lane / C4, c4 = lane - e * C4;
    if (e >= epw) return;
    const float4 ctr = xb[(size_t)n * C4 + c4];
    if (k <= 128 && epw * C4 == 64) {
      // the whole neighbour list in registers (one coalesced load: lane l holds entries l, l + 64);
      // the row gathers take their index from a lane shuffle and two of them are in flight per
      // trip — the chain "load index -> load row -> store" per step was what the wave waited for
      const int jlo = lane < k ? (int)ib[lane] : n, jhi = lane + 64 < k ? (int)ib[lane + 64] : n;
      for (int kk0 = 0; kk0 < k; kk0 += 2 * epw) {
        const int ka = kk0 + e, kb = ka + epw;
        const int sa = ka < 64 ? __shfl(jlo, ka & 63, 64) : __shfl(jhi, (ka - 64) & 63, 64);
        const int sb = kb < 64 ? __shfl(jlo, kb & 63, 64) : __shfl(jhi, (kb - 64) & 63, 64);
        const bool oa = ka < k, ob = kb < k;
        const float4 na = xb[(size_t)(oa ? sa : n) * C4 + c4];
        const float4 nbv = xb[(size_t)(ob ? sb : n) * C4 + c4];
        if (oa) {
          fo[(size_t)ka * 2 * C4 + c4] = make_float4(na.x - ctr.x, na.y - ctr.y, na.z - ctr.z, na.w - ctr.w);
          fo[(size_t)ka * 2 * C4 + C4 + c4] = ctr;
        }
        if (ob) {
          fo[(size_t)kb * 2 * C4 + c4] = make_float4(nbv.x - ctr.x, nbv.y - ctr.y, nbv.z - ctr.z, nbv.w - ctr.w);
          fo[(size_t)kb * 2 * C4 + C4 + c4] = ctr;
        }
      }
      return;
    }
    for (int kk = e; kk < k; kk += epw) {
      const int j = (int)ib[kk];
      const float4 nb = xb[(size_t)j * C4 + c4];
      float4 d;
      d.x = nb.x - ctr.x;
      d.y = nb.y - ctr.y;
      d.z = nb.z - ctr.z;
      d.w = nb.w - ctr.w;
      fo[(size_t)kk * 2 * C4 + c4] = d;
      fo[(size_t)kk * 2 * C4 + C4 + c4] = ctr;
    }
  } else {
    for (int kk = 0; kk < k; ++kk) {
      const int j = (int)ib[kk];
      for (int c4 = lane; c4 < C4; c4 += 64) {
        const float4 ctr = xb[(size_t)n * C4 + c4];
        const float4 nb = xb[(size_t)j * C4 + c4];
        float4 d;
        d.x = nb.x - ctr.x;
        d.y = nb.y - ctr.y;
        d.z = nb.z - ctr.z;
        d.w = nb.w - ctr.w;
        fo[(size_t)kk * 2 * C4 + c4] = d;
        fo[(size_t)kk * 2 * C4 + C4 + c4] = ctr;
      }
    }
  }
}

// scalar path for the narrow first layer (C = 3 or 6): one thread per output element
__global__ void pn_edge_feature_scalar_kernel(const float* __restrict__ xt,
                                              const int64_t* __restrict__ idx, int N, int k, int C,
                                              long long total, float* __restrict__ feat) {
  const long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= total) return;
  const int c2 = (int)(o % (2 * C));
  const long long e = o / (2 * C);          // edge = (b*N + n)*k + kk
  const long long pn = e / k;               // b*N + n
  const long long b = pn / N;
  const float ctr = xt[pn * C + (c2 < C ? c2 : c2 - C)];
  if (c2 < C) {
    const int j = (int)idx[e];
    feat[o] = xt[(b * N + j) * C + c2] - ctr;
  } else {
    feat[o] = ctr;
  }
}

extern "C" int pn_edge_feature_fwd_f32(const float* xt, const int64_t* idx, int B, int N, int k,
                                       int C, float* feat, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PN_CHECK_ARG(xt && idx && feat, "pn_edge_feature_fwd_f32: null pointer");
  PN_CHECK_ARG(B > 0 && N > 0 && k > 0 && C > 0, "pn_edge_feature_fwd_f32: empty input");
  PN_PROF("edge_feature_fwd", stream);
  if ((C & 3) == 0) {
    dim3 grid(pn_cdiv(N, 4), B);
    hipLaunchKernelGGL(pn_edge_feature_vec_kernel, grid, dim3(256), 0, stream, xt, idx, N, k, C,
                       feat);
  } else {
    const long long total = (long long)B * N * k * 2 * C;
    hipLaunchKernelGGL(pn_edge_feature_scalar_kernel, dim3(pn_cdiv(total, 256)), dim3(256), 0,
                       stream, xt, idx, N, k, C, total, feat);
  }
  PN_CHECK_LAUNCH();
  return PN_OK;
}

// (the backward of the API form follows the reverse-graph builder further down: it gathers)

// ------------------------------------------------------------------------------------
// (2) fused edge convolution: gather-reduce over rows of P
// ------------------------------------------------------------------------------------
#define EC_PPW_MAX 16  // points per wave (chosen per launch: ec_points_per_wave)
#define EC_PPW_MIN 2
#define EC_WAVES 4
#define EC_INFL 2      // row gathers in flight per lane (4: measured slower, 0.107 against 0.094 ms at cfg4 — occupancy)

struct float4x {
  float v[4];
};
__device__ static inline float4x ld4(const float* p) {
  const float4 t = *reinterpret_cast<const float4*>(p);
  return {{t.x, t.y, t.z, t.w}};
}
#ifdef EC_NT_GATHER
// (experiment, -DEC_NT_GATHER: row gathers that do not allocate in the vector L1 — measured 0.127 against 0.096 ms
// per launch at B = 4, N = 10 000, k = 80, Cout = 64: neighbouring points share neighbours, the L1 does serve rows)
__device__ static inline float4x ld4g(const float* p) {
  typedef float ec_f4 __attribute__((ext_vector_type(4)));
  const ec_f4 t = __builtin_nontemporal_load(reinterpret_cast<const ec_f4*>(p));
  return {{t.x, t.y, t.z, t.w}};
}
#else
#define ld4g ld4
#endif
__device__ static inline void st4(float* p, const float4x& a) {
  *reinterpret_cast<float4*>(p) = make_float4(a.v[0], a.v[1], a.v[2], a.v[3]);
}

// PQ (B,N,2*COUT): [P | Q] per point.  LPR = COUT/4 lanes cover one row; RPI rows per wave step.
// Group statistics: Cg channels per group (group g = c / Cg).
// part: double [B][gridDim.x][COUT/Cg][2], one partial (sum y, sum y^2) per workgroup and group,
// combined in index order by pn_stats_reduce_kernel (no atomics: the moments, and with them every
// activation of the layer, are bit-reproducible run to run).
// Workgroups are dealt to the 8 XCDs round-robin by their linear id, and every XCD has its own 4 MiB L2.
// The gather kernels therefore take their (item, block) from a VIRTUAL id: XCD x works through the x-th
// contiguous eighth of the item-major block sequence, so that the rows it gathers belong to one item (or two
// at a boundary) — 2.6 MB of P rows at N = 10 000, Cout = 64 — instead of to all items of the batch at once.
__device__ static inline int pn_xcd_virtual(int w, int total) {
  const int per = (total + 7) >> 3;
  return (w & 7) * per + (w >> 3);
}
static inline int pn_xcd_grid(int total) { return 8 * ((total + 7) >> 3); }

template <int COUT, typename IT>
__global__ __launch_bounds__(256) void pn_edgeconv_reduce_kernel(
    const float* __restrict__ PQ, const IT* __restrict__ idx, const float* __restrict__ gamma,
    int N, int k, int Cg, int per_sample, int nblk, int B, int ppw, float* __restrict__ yext,
    uint8_t* __restrict__ argk, float* __restrict__ s1out, double* __restrict__ part) {
  constexpr int NCH = COUT <= 256 ? 1 : COUT / 256;  // float4 chunks per lane
  constexpr int LPR = COUT <= 256 ? COUT / 4 : 64;   // lanes per row
  constexpr int RPI = 64 / LPR;                      // rows per wave step
  __shared__ double s_part[EC_WAVES][2][COUT];
  const int vid = pn_xcd_virtual(blockIdx.x, nblk * B);
  if (vid >= nblk * B) return;
  const int b = vid / nblk, blk = vid - b * nblk;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int rg = lane / LPR, cl = lane - rg * LPR;
  const float* __restrict__ PQb = PQ + (size_t)b * N * 2 * COUT;
  float sgn[NCH][4];
#pragma unroll
  for (int h = 0; h < NCH; ++h)
#pragma unroll
    for (int u = 0; u < 4; ++u) sgn[h][u] = gamma[(cl + h * 64) * 4 + u] >= 0.f ? 1.f : -1.f;
  double d1[NCH][4], d2[NCH][4];
#pragma unroll
  for (int h = 0; h < NCH; ++h)
#pragma unroll
    for (int u = 0; u < 4; ++u) d1[h][u] = d2[h][u] = 0.0;

  const int p0 = (blk * EC_WAVES + wave) * ppw;
  // The neighbour list of a point lives in registers (lane l holds entries l and l + 64; k <= 128)
  // and is fetched ONE POINT AHEAD: the row gathers then depend on a lane shuffle instead of on a
  // load of the index that has just been issued (the profile showed 80 % of the wave cycles waiting
  // on that chain of two dependent loads per step), and several gathers can be in flight.
  const bool regs = k <= 128;
  int nlo = 0, nhi = 0;
  if (regs && p0 < N) {
    const IT* __restrict__ ib0 = idx + ((size_t)b * N + p0) * k;
    nlo = lane < k ? (int)ib0[lane] : 0;
    nhi = lane + 64 < k ? (int)ib0[lane + 64] : 0;
  }
  for (int pi = 0; pi < ppw; ++pi) {
    const int i = p0 + pi;
    if (i >= N) break;  // wave-uniform
    const IT* __restrict__ ib = idx + ((size_t)b * N + i) * k;
    const int jlo = nlo, jhi = nhi;
    if (regs && pi + 1 < ppw && i + 1 < N) {
      const IT* __restrict__ ibn = ib + k;
      nlo = lane < k ? (int)ibn[lane] : 0;
      nhi = lane + 64 < k ? (int)ibn[lane + 64] : 0;
    }
    float4x q[NCH], best[NCH], s1[NCH], s2[NCH];
    int arg[NCH][4];
#pragma unroll
    for (int h = 0; h < NCH; ++h) {
      q[h] = ld4(PQb + (size_t)i * 2 * COUT + COUT + (cl + h * 64) * 4);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        best[h].v[u] = -__builtin_inff();
        s1[h].v[u] = 0.f;
        s2[h].v[u] = 0.f;
        arg[h][u] = 0;
      }
    }
    // EC_INFL steps of the neighbour loop per trip, all row gathers issued before any is used (the kernel waits
    // for gathers, not for issue slots: round 3 went from one to two in flight; four measured slower, see EC_INFL)
    for (int kk0 = 0; kk0 < k; kk0 += EC_INFL * RPI) {
      int kkv[EC_INFL];
      bool onv[EC_INFL];
      float4x vv[EC_INFL][NCH];
#pragma unroll
      for (int f = 0; f < EC_INFL; ++f) {
        kkv[f] = kk0 + rg + f * RPI;
        // (the shuffles are executed by every lane: the source lane must be active)
        const int sj = kkv[f] < 64 ? __shfl(jlo, kkv[f] & 63, 64) : __shfl(jhi, (kkv[f] - 64) & 63, 64);
        onv[f] = kkv[f] < k;
        const int jf = onv[f] ? (regs ? sj : (int)ib[kkv[f]]) : i;      // inactive: a row that exists
#pragma unroll
        for (int h = 0; h < NCH; ++h) vv[f][h] = ld4g(PQb + (size_t)jf * 2 * COUT + (cl + h * 64) * 4);
      }
#pragma unroll
      for (int f = 0; f < EC_INFL; ++f) {
        if (onv[f]) {
#pragma unroll
          for (int h = 0; h < NCH; ++h) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const float y = vv[f][h].v[u] + q[h].v[u];
              const float ys = y * sgn[h][u];
              if (ys > best[h].v[u]) {
                best[h].v[u] = ys;
                arg[h][u] = kkv[f];
              }
              s1[h].v[u] += y;
              s2[h].v[u] = __builtin_fmaf(y, y, s2[h].v[u]);
            }
          }
        }
      }
    }
    // combine the RPI row groups (lanes cl, cl+LPR, ...); ties -> smaller kk
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1) {
#pragma unroll
      for (int h = 0; h < NCH; ++h)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float ob = __shfl_xor(best[h].v[u], o, 64);
          const int oa = __shfl_xor(arg[h][u], o, 64);
          if (ob > best[h].v[u] || (ob == best[h].v[u] && oa < arg[h][u])) {
            best[h].v[u] = ob;
            arg[h][u] = oa;
          }
          s1[h].v[u] += __shfl_xor(s1[h].v[u], o, 64);
          s2[h].v[u] += __shfl_xor(s2[h].v[u], o, 64);
        }
    }
    if (rg == 0) {
#pragma unroll
      for (int h = 0; h < NCH; ++h) {
        const size_t o = ((size_t)b * N + i) * COUT + (cl + h * 64) * 4;
        float4x e;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          e.v[u] = best[h].v[u] * sgn[h][u];
          d1[h][u] += (double)s1[h].v[u];
          d2[h][u] += (double)s2[h].v[u];
        }
        st4(yext + o, e);
        st4(s1out + o, s1[h]);
        *reinterpret_cast<uchar4*>(argk + o) =
            make_uchar4((uint8_t)arg[h][0], (uint8_t)arg[h][1], (uint8_t)arg[h][2], (uint8_t)arg[h][3]);
      }
    }
  }
  // block reduction of the statistics in a fixed order, one partial per (group, moment)
  if (rg == 0) {
#pragma unroll
    for (int h = 0; h < NCH; ++h)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        s_part[wave][0][(cl + h * 64) * 4 + u] = d1[h][u];
        s_part[wave][1][(cl + h * 64) * 4 + u] = d2[h][u];
      }
  }
  __syncthreads();
  const int G = COUT / Cg;
  for (int t = threadIdx.x; t < 2 * G; t += blockDim.x) {
    const int g = t >> 1, which = t & 1;
    double acc = 0.0;
    for (int w = 0; w < EC_WAVES; ++w)
      for (int c = g * Cg; c < (g + 1) * Cg; ++c) acc += s_part[w][which][c];
    part[(((size_t)b * nblk + blk) * G + g) * 2 + which] = acc;
  }
}

// generic (any Cout): EC_GEN_PTS points per workgroup, one thread per (point, channel) pair at a
// time; used for unusual widths only.  The per-pair sums go through LDS so that the group partial
// of the workgroup is formed in a fixed order.
#define EC_GEN_PTS 16
template <typename IT>
__global__ __launch_bounds__(256) void pn_edgeconv_reduce_generic_kernel(
    const float* __restrict__ PQ, const IT* __restrict__ idx, const float* __restrict__ gamma,
    int N, int k, int Cout, int Cg, float* __restrict__ yext, uint8_t* __restrict__ argk,
    float* __restrict__ s1out, double* __restrict__ part) {
  extern __shared__ float ec_gen_sh[];  // [2][EC_GEN_PTS * Cout]
  const int b = blockIdx.y;
  const int p0 = blockIdx.x * EC_GEN_PTS;
  const float* PQb = PQ + (size_t)b * N * 2 * Cout;
  float* sh1 = ec_gen_sh;
  float* sh2 = ec_gen_sh + EC_GEN_PTS * Cout;
  for (int pc = threadIdx.x; pc < EC_GEN_PTS * Cout; pc += 256) {
    const int pl = pc / Cout, c = pc - pl * Cout, i = p0 + pl;
    float s1 = 0.f, s2 = 0.f;
    if (i < N) {
      const IT* ib = idx + ((size_t)b * N + i) * k;
      const float q = PQb[(size_t)i * 2 * Cout + Cout + c];
      const float sg = gamma[c] >= 0.f ? 1.f : -1.f;
      float best = -__builtin_inff();
      int arg = 0;
      for (int kk = 0; kk < k; ++kk) {
        const float y = PQb[(size_t)(int)ib[kk] * 2 * Cout + c] + q;
        const float ys = y * sg;
        if (ys > best) {
          best = ys;
          arg = kk;
        }
        s1 += y;
        s2 = __builtin_fmaf(y, y, s2);
      }
      const size_t o = ((size_t)b * N + i) * Cout + c;
      yext[o] = best * sg;
      s1out[o] = s1;
      argk[o] = (uint8_t)arg;
    }
    sh1[pc] = s1;
    sh2[pc] = s2;
  }
  __syncthreads();
  const int G = Cout / Cg;
  for (int t = threadIdx.x; t < 2 * G; t += 256) {
    const int g = t >> 1, which = t & 1;
    const float* sh = which ? sh2 : sh1;
    double acc = 0.0;
    for (int pl = 0; pl < EC_GEN_PTS; ++pl)
      for (int c = g * Cg; c < (g + 1) * Cg; ++c) acc += (double)sh[pl * Cout + c];
    part[(((size_t)b * gridDim.x + blockIdx.x) * G + g) * 2 + which] = acc;
  }
}

// stats[s][g][which] = sum over the workgroup partials of item s (per_sample) or of all items, in
// index order: lane l takes partials l, l + 64, ..., then a fixed xor tree.  One wave per result.
__global__ __launch_bounds__(64) void pn_stats_reduce_kernel(const double* __restrict__ part, int B, int nblk,
                                                              int G, int per_sample, double* __restrict__ stats) {
  const int out = blockIdx.x;                 // (s * G + g) * 2 + which
  const int which = out & 1, g = (out >> 1) % G, s = (out >> 1) / G;
  const int b0 = per_sample ? s : 0, nb = per_sample ? 1 : B;
  double acc = 0.0;
  // (item bb, block blk) = entry b0 * nblk + e of the (items x blocks) sequence: one stride for all entries;
  // eight loads in flight, added in the same order as one at a time
  const double* __restrict__ pp = part + ((size_t)b0 * nblk * G + g) * 2 + which;
  const size_t stride = (size_t)G * 2;
  const int total = nb * nblk;
  int e = threadIdx.x;
  for (; e + 7 * 64 < total; e += 8 * 64) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = pp[(size_t)(e + 64 * u) * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  for (; e < total; e += 64) acc += pp[(size_t)e * stride];
  acc = pn_wave_sum_d(acc);
  if (threadIdx.x == 0) stats[out] = acc;
}

extern "C" size_t pn_edgeconv_reduce_workspace(int B, int N, int Cout, int groups) {
  (void)Cout;
  return pn_align_up((size_t)B * pn_cdiv(N, EC_WAVES * EC_PPW_MIN) * groups * 2 * sizeof(double), 256);
}

// Points per wave of the gather-reduce: the kernel is bound by the latency of its row gathers (two in flight per
// wave), so the chip wants all the waves it can hold — eight per SIMD — before a wave gets a second point;
// 16 points per wave (the round-3 value) left 2.5 waves per SIMD at B = 4, N = 10 000.
static int ec_points_per_wave(int B, int N) {
  if (const char* e = getenv("PN_EC_PPW")) {       // developer override (a power of two in [EC_PPW_MIN, EC_PPW_MAX])
    const int v = atoi(e);
    if (v >= EC_PPW_MIN && v <= EC_PPW_MAX && (v & (v - 1)) == 0) return v;
  }
  int ppw = (int)(((long long)B * N) / 8192);
  int p = EC_PPW_MIN;
  while (p * 2 <= ppw && p < EC_PPW_MAX) p *= 2;
  return p;
}

template <typename IT>
static int edgeconv_reduce_fwd(const float* PQ, const IT* idx, const float* gamma, int B, int N, int k, int Cout,
                               int groups, int per_sample, float* yext, uint8_t* argk, float* s1, double* stats,
                               void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PN_CHECK_ARG(PQ && idx && gamma && yext && argk && s1 && stats && workspace,
               "pn_edgeconv_reduce_fwd: null pointer");
  PN_CHECK_ARG(B > 0 && N > 0 && k > 0 && k <= 255 && Cout > 0,
               "pn_edgeconv_reduce_fwd: bad sizes (B=%d N=%d k=%d Cout=%d)", B, N, k, Cout);
  PN_CHECK_ARG(groups > 0 && Cout % groups == 0, "pn_edgeconv_reduce_fwd: groups=%d", groups);
  PN_CHECK_ARG(workspace_bytes >= pn_edgeconv_reduce_workspace(B, N, Cout, groups),
               "pn_edgeconv_reduce_fwd: workspace too small");
  const int Cg = Cout / groups;
  double* part = (double*)workspace;
  const int ppw = ec_points_per_wave(B, N);
  int nblk = pn_cdiv(N, EC_WAVES * ppw);
  PN_PROF("edgeconv_reduce_fwd", stream);
  dim3 grid(pn_xcd_grid(nblk * B));
#define EC_GO(CO)                                                                                       \
  hipLaunchKernelGGL((pn_edgeconv_reduce_kernel<CO, IT>), grid, dim3(256), 0, stream, PQ, idx, gamma, N, k, Cg, \
                     per_sample, nblk, B, ppw, yext, argk, s1, part)
  if (Cout == 64)
    EC_GO(64);
  else if (Cout == 128)
    EC_GO(128);
  else if (Cout == 256)
    EC_GO(256);
  else if (Cout == 512)
    EC_GO(512);
  else {
    const size_t lds = (size_t)2 * EC_GEN_PTS * Cout * sizeof(float);
    if (lds > 160 * 1024) {
      pn_set_error("pn_edgeconv_reduce_fwd: Cout=%d is neither 64/128/256/512 nor <= 1280", Cout);
      return PN_ERR_UNSUPPORTED;
    }
    dim3 g2(pn_cdiv(N, EC_GEN_PTS), B);
    nblk = (int)g2.x;
    if (lds > 64 * 1024)
      PN_CHECK_HIP(hipFuncSetAttribute((const void*)pn_edgeconv_reduce_generic_kernel<IT>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(pn_edgeconv_reduce_generic_kernel<IT>, g2, dim3(256), lds, stream, PQ, idx, gamma,
                       N, k, Cout, Cg, yext, argk, s1, part);
  }
#undef EC_GO
  hipLaunchKernelGGL(pn_stats_reduce_kernel, dim3((per_sample ? B : 1) * groups * 2), dim3(64), 0, stream,
                     (const double*)part, B, nblk, groups, per_sample, stats);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

extern "C" int pn_edgeconv_reduce_fwd_f32(const float* PQ, const int64_t* idx, const float* gamma,
                                          int B, int N, int k, int Cout, int groups,
                                          int per_sample, float* yext, uint8_t* argk, float* s1,
                                          double* stats, void* workspace, size_t workspace_bytes,
                                          void* stream_) {
  return edgeconv_reduce_fwd<int64_t>(PQ, idx, gamma, B, N, k, Cout, groups, per_sample, yext, argk, s1, stats,
                                      workspace, workspace_bytes, stream_);
}
// the same on the library's int32 graph (pn_knn_graph_i32)
extern "C" int pn_edgeconv_reduce_fwd_i32(const float* PQ, const int32_t* idx, const float* gamma,
                                          int B, int N, int k, int Cout, int groups,
                                          int per_sample, float* yext, uint8_t* argk, float* s1,
                                          double* stats, void* workspace, size_t workspace_bytes,
                                          void* stream_) {
  return edgeconv_reduce_fwd<int32_t>(PQ, idx, gamma, B, N, k, Cout, groups, per_sample, yext, argk, s1, stats,
                                      workspace, workspace_bytes, stream_);
}

// mean / rstd of every group from the fp64 moments (count = elements per group)
__global__ void pn_moments_kernel(const double* __restrict__ stats, int n, double count, float eps,
                                  float* __restrict__ mean, float* __restrict__ rstd) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const double m = stats[2 * t] / count;
  double var = stats[2 * t + 1] / count - m * m;
  if (var < 0.0) var = 0.0;
  mean[t] = (float)m;
  rstd[t] = (float)(1.0 / sqrt(var + (double)eps));
}

extern "C" int pn_moments_f32(const double* stats, int n, double count, float eps, float* mean,
                              float* rstd, void* stream) {
  PN_CHECK_ARG(stats && mean && rstd && n > 0 && count > 0, "pn_moments_f32: bad arguments");
  hipLaunchKernelGGL(pn_moments_kernel, dim3(pn_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     stats, n, count, eps, mean, rstd);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

// out[b,c,n] = lrelu(gamma[c] * (yext[b,n,c] - mean) * rstd + beta[c]); mean/rstd indexed by
// (per_sample ? b : 0, c / Cg).  Point-major in, channel-first out (LDS transpose).
__global__ void pn_edgeconv_finalize_kernel(const float* __restrict__ yext,
                                            const float* __restrict__ mean,
                                            const float* __restrict__ rstd,
                                            const float* __restrict__ gamma,
                                            const float* __restrict__ beta, int N, int Cout, int Cg,
                                            int per_sample, float slope, float* __restrict__ out) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int c0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int G = Cout / Cg;
  for (int i = ty; i < 32; i += 8) {
    const int n = n0 + i, c = c0 + tx;
    if (n < N && c < Cout) {
      const int sidx = (per_sample ? b : 0) * G + c / Cg;
      const float yh = (yext[((size_t)b * N + n) * Cout + c] - mean[sidx]) * rstd[sidx];
      const float z = __builtin_fmaf(gamma[c], yh, beta[c]);
      tile[i][tx] = z > 0.f ? z : z * slope;
    }
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, n = n0 + tx;
    if (n < N && c < Cout) out[((size_t)b * Cout + c) * N + n] = tile[tx][i];
  }
}

extern "C" int pn_edgeconv_finalize_fwd_f32(const float* yext, const float* mean,
                                            const float* rstd, const float* gamma,
                                            const float* beta, int B, int N, int Cout, int groups,
                                            int per_sample, float slope, float* out, void* stream) {
  PN_CHECK_ARG(yext && mean && rstd && gamma && beta && out, "pn_edgeconv_finalize_fwd_f32: null");
  PN_CHECK_ARG(groups > 0 && Cout % groups == 0, "pn_edgeconv_finalize_fwd_f32: groups=%d", groups);
  dim3 grid(pn_cdiv(Cout, 32), pn_cdiv(N, 32), B);
  hipLaunchKernelGGL(pn_edgeconv_finalize_kernel, grid, dim3(256), 0, (hipStream_t)stream, yext,
                     mean, rstd, gamma, beta, N, Cout, Cout / groups, per_sample, slope, out);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

// Backward, step A (elementwise, with the transpose back to point-major):
//   yhat = (yext - mean) * rstd ; z = gamma*yhat + beta ; gz = gout * (z > 0 ? 1 : slope)
// writes gz and yhat as (B,N,Cout).
__global__ void pn_edgeconv_bwd_prep_kernel(const float* __restrict__ gout,
                                            const float* __restrict__ yext,
                                            const float* __restrict__ mean,
                                            const float* __restrict__ rstd,
                                            const float* __restrict__ gamma,
                                            const float* __restrict__ beta, int N, int Cout, int Cg,
                                            int per_sample, float slope, float* __restrict__ gz,
                                            float* __restrict__ yhat) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int c0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int G = Cout / Cg;
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, n = n0 + tx;
    if (n < N && c < Cout) tile[i][tx] = gout[((size_t)b * Cout + c) * N + n];
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int n = n0 + i, c = c0 + tx;
    if (n < N && c < Cout) {
      const int sidx = (per_sample ? b : 0) * G + c / Cg;
      const size_t o = ((size_t)b * N + n) * Cout + c;
      const float yh = (yext[o] - mean[sidx]) * rstd[sidx];
      const float z = __builtin_fmaf(gamma[c], yh, beta[c]);
      gz[o] = tile[tx][i] * (z > 0.f ? 1.f : slope);
      yhat[o] = yh;
    }
  }
}

extern "C" int pn_edgeconv_bwd_prep_f32(const float* gout, const float* yext, const float* mean,
                                        const float* rstd, const float* gamma, const float* beta,
                                        int B, int N, int Cout, int groups, int per_sample,
                                        float slope, float* gz, float* yhat, void* stream) {
  PN_CHECK_ARG(gout && yext && mean && rstd && gamma && beta && gz && yhat,
               "pn_edgeconv_bwd_prep_f32: null pointer");
  dim3 grid(pn_cdiv(Cout, 32), pn_cdiv(N, 32), B);
  hipLaunchKernelGGL(pn_edgeconv_bwd_prep_kernel, grid, dim3(256), 0, (hipStream_t)stream, gout,
                     yext, mean, rstd, gamma, beta, N, Cout, Cout / groups, per_sample, slope, gz,
                     yhat);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

// Backward, step B: the exact normalisation gradient on every edge.
// With t = gamma*gz (per point/channel), c1 = mean_group(t), c2 = mean_group(t*yhat) over the
// M = Cg*N*k (x B for batch statistics) edge activations of the group:
//   dy_e = rstd * ( [e is the extreme edge] * t  -  c1  -  c2 * yhat_e ),  yhat_e = (P[j]+Q[i]-mean)*rstd
//   dQ[i] = sum_kk dy  = rstd * ( t - k*c1 - c2 * rstd * (s1 - k*mean) )
//   dP[j] = sum over the edges (i -> j) that END in j of dy_e
// The second sum runs over the TRANSPOSED kNN graph: it is built once per call as a CSR whose
// lists are sorted by (source point, neighbour slot), and every dP row is then formed by ONE wave
// in list order — no floating-point atomics anywhere, the gradient is bit-reproducible run to
// run (round 4; before, the extreme edges were scattered with fp32 atomics and the order inside
// a list depended on the wave schedule).  c1c2: float [(per_sample?B:1)][G][2].

// ---- reverse kNN graph (CSR by target point) -----------------------------------------------
// Counting sort of the B*N*k edges by target with workgroup-private LDS histograms: the edges
// of one item are dealt to G workgroups in contiguous chunks (chunk g = edges [g*chunk, ...), i.e.
// ascending source points); each counts its share (ds_add), writes the histogram out, a prefix
// over the G partial histograms of every target and a scan over the targets give per-workgroup
// start positions, and the fill pass replays the same edges against LDS cursors (ds_add_rtn).
// No global atomics.  An entry is (source << 8) | slot.  Inside the bucket of one (workgroup,
// target) pair the order is the LDS unit's; pn_rev_sort_kernel then sorts every list (the buckets
// of a list are already in ascending order of their sources, so a long list is sorted bucket by
// bucket).  Targets beyond REV_LDS_MAXN are handled in windows of that many counters.
#define REV_G_MIN 64
#define REV_LDS_MAXN 16384   // 64 KiB of LDS counters
#define REV_SORT_CAP 1024    // entries a wave sorts at a time (4 KiB of LDS per wave)

static inline int pn_rev_groups(int N) {
  // a bucket holds at most one entry per source point of its chunk: N / G + 2 <= REV_SORT_CAP
  const int g = pn_cdiv(N, REV_SORT_CAP - 2);
  return g > REV_G_MIN ? g : REV_G_MIN;
}

template <typename IT>
__global__ __launch_bounds__(256) void pn_rev_count_lds_kernel(const IT* __restrict__ idx, int N, int k,
                                                               int G, int* __restrict__ part) {
  extern __shared__ int rev_hist[];
  const int b = blockIdx.y, g = blockIdx.x;
  const int w0 = blockIdx.z * REV_LDS_MAXN;
  const int wn = N - w0 < REV_LDS_MAXN ? N - w0 : REV_LDS_MAXN;
  for (int j = threadIdx.x; j < wn; j += 256) rev_hist[j] = 0;
  __syncthreads();
  const long long per = (long long)N * k;
  const long long chunk = (per + G - 1) / G;
  const long long e0 = g * chunk, e1 = e0 + chunk < per ? e0 + chunk : per;
  const IT* __restrict__ ib = idx + (size_t)b * per;
  for (long long e = e0 + threadIdx.x; e < e1; e += 256) {
    const int j = (int)ib[e] - w0;
    if ((unsigned)j < (unsigned)wn) atomicAdd(&rev_hist[j], 1);
  }
  __syncthreads();
  int* __restrict__ pb = part + ((size_t)b * G + g) * N + w0;
  for (int j = threadIdx.x; j < wn; j += 256) pb[j] = rev_hist[j];
}

// per target j: part[g][j] -> start of workgroup g inside the list of j (exclusive prefix over g,
// in place), deg[j] = length of the list (scanned over j by pn_rev_scan_kernel)
__global__ __launch_bounds__(256) void pn_rev_binprefix_kernel(int* __restrict__ part, int N, int G,
                                                               int* __restrict__ deg) {
  const int b = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
  if (j >= N) return;
  int* __restrict__ pb = part + (size_t)b * G * N + j;
  int run = 0;
#pragma unroll 8
  for (int g = 0; g < G; ++g) {
    const int d = pb[(size_t)g * N];
    pb[(size_t)g * N] = run;
    run += d;
  }
  deg[(size_t)b * N + j] = run;
}

// one block per item: exclusive scan of deg -> off (N+1 entries)
__global__ __launch_bounds__(1024) void pn_rev_scan_kernel(const int* __restrict__ deg, int N,
                                                           int* __restrict__ off) {
  __shared__ int part[1024];
  const int b = blockIdx.x, t = threadIdx.x;
  const int per = (N + 1023) / 1024;
  const int lo = min(N, t * per), hi = min(N, lo + per);
  const int* d = deg + (size_t)b * N;
  int s = 0;
  for (int i = lo; i < hi; ++i) s += d[i];
  part[t] = s;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const int v = t >= o ? part[t - o] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int run = part[t] - s;  // exclusive prefix of this thread's chunk
  int* ob = off + (size_t)b * (N + 1);
  for (int i = lo; i < hi; ++i) {
    ob[i] = run;
    run += d[i];
  }
  if (t == 1023) ob[N] = part[1023];
}

template <typename IT>
__global__ __launch_bounds__(256) void pn_rev_fill_lds_kernel(const IT* __restrict__ idx, int N, int k,
                                                              int G, const int* __restrict__ part,
                                                              const int* __restrict__ off,
                                                              uint32_t* __restrict__ rev) {
  extern __shared__ int rev_hist[];
  const int b = blockIdx.y, g = blockIdx.x;
  const int w0 = blockIdx.z * REV_LDS_MAXN;
  const int wn = N - w0 < REV_LDS_MAXN ? N - w0 : REV_LDS_MAXN;
  const int* __restrict__ pb = part + ((size_t)b * G + g) * N + w0;
  const int* __restrict__ ob = off + (size_t)b * (N + 1) + w0;
  for (int j = threadIdx.x; j < wn; j += 256) rev_hist[j] = ob[j] + pb[j];
  __syncthreads();
  const long long per = (long long)N * k;
  const long long chunk = (per + G - 1) / G;
  const long long e0 = g * chunk, e1 = e0 + chunk < per ? e0 + chunk : per;
  const IT* __restrict__ ib = idx + (size_t)b * per;
  uint32_t* __restrict__ rb = rev + (size_t)b * per;
  for (long long e = e0 + threadIdx.x; e < e1; e += 256) {
    const int j = (int)ib[e] - w0;
    if ((unsigned)j < (unsigned)wn) {
      const int pos = atomicAdd(&rev_hist[j], 1);
      const int i = (int)(e / k);
      rb[pos] = ((uint32_t)i << 8) | (uint32_t)(e - (long long)i * k);  // (source point, slot)
    }
  }
}

// Rank sort of n <= REV_SORT_CAP distinct entries at p (global), staged in the wave's LDS region.
// Executed by one whole wave; LDS traffic of a wave is ordered, the fences keep the compiler from
// moving the reads above the writes.
__device__ static inline void pn_rev_sort_segment(uint32_t* __restrict__ p, int n, uint32_t* lst, int lane) {
  if (n > REV_SORT_CAP) return;   // only a graph with repeated neighbours in a row gets here: left as filled
  const int n4 = (n + 3) & ~3;
  for (int t = lane; t < n4; t += 64) lst[t] = t < n ? p[t] : 0xffffffffu;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  for (int t = lane; t < n; t += 64) {
    const uint32_t v = lst[t];
    int rank = 0;
    for (int u = 0; u < n4; u += 4) {
      const uint4 w = *reinterpret_cast<const uint4*>(lst + u);
      rank += (int)(w.x < v) + (int)(w.y < v) + (int)(w.z < v) + (int)(w.w < v);
    }
    p[rank] = v;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// one wave per target point: its list in ascending (source, slot) order
__global__ __launch_bounds__(256) void pn_rev_sort_kernel(const int* __restrict__ off, const int* __restrict__ part,
                                                          int N, int k, int G, uint32_t* __restrict__ rev) {
  __shared__ __attribute__((aligned(16))) uint32_t s_lst[4][REV_SORT_CAP];
  const int b = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j = blockIdx.x * 4 + wave;
  if (j >= N) return;
  const int* __restrict__ ob = off + (size_t)b * (N + 1);
  const int e0 = ob[j], L = ob[j + 1] - e0;
  if (L <= 1) return;
  uint32_t* __restrict__ rb = rev + (size_t)b * N * k + e0;
  if (L <= REV_SORT_CAP) {
    pn_rev_sort_segment(rb, L, s_lst[wave], lane);
    return;
  }
  const int* __restrict__ pb = part + (size_t)b * G * N + j;
  int s = 0;  // = pb[0]
  for (int g = 0; g < G; ++g) {
    const int nxt = g + 1 < G ? pb[(size_t)(g + 1) * N] : L;
    if (nxt - s > 1) pn_rev_sort_segment(rb + s, nxt - s, s_lst[wave], lane);
    s = nxt;
  }
}

extern "C" size_t pn_edgeconv_bwd_workspace(int B, int N, int k) {
  return pn_align_up((size_t)B * N * 4, 256) + pn_align_up((size_t)B * (N + 1) * 4, 256) +
         pn_align_up((size_t)B * N * k * 4, 256) + pn_align_up((size_t)B * pn_rev_groups(N) * N * 4, 256);
}

// builds the sorted CSR of the transposed graph in ``workspace``; off (B,N+1), rev (B,N*k)
template <typename IT>
static int pn_build_rev_csr(const IT* idx, int B, int N, int k, void* workspace, size_t workspace_bytes,
                            hipStream_t stream, const int** off_out, const uint32_t** rev_out) {
  PN_CHECK_ARG(N < (1 << 24) && k <= 255, "reverse graph: N=%d (max 2^24 - 1), k=%d (max 255)", N, k);
  PN_CHECK_ARG(workspace && workspace_bytes >= pn_edgeconv_bwd_workspace(B, N, k),
               "reverse graph: workspace too small");
  const int G = pn_rev_groups(N);
  char* w = (char*)workspace;
  int* deg = (int*)w;
  w += pn_align_up((size_t)B * N * 4, 256);
  int* off = (int*)w;
  w += pn_align_up((size_t)B * (N + 1) * 4, 256);
  uint32_t* rev = (uint32_t*)w;
  w += pn_align_up((size_t)B * N * k * 4, 256);
  int* part = (int*)w;
  const int nwin = pn_cdiv(N, REV_LDS_MAXN);
  const size_t lds = (size_t)(N < REV_LDS_MAXN ? N : REV_LDS_MAXN) * sizeof(int);
  hipLaunchKernelGGL(pn_rev_count_lds_kernel<IT>, dim3(G, B, nwin), dim3(256), lds, stream, idx, N, k, G, part);
  hipLaunchKernelGGL(pn_rev_binprefix_kernel, dim3(pn_cdiv(N, 256), B), dim3(256), 0, stream, part, N, G, deg);
  hipLaunchKernelGGL(pn_rev_scan_kernel, dim3(B), dim3(1024), 0, stream, (const int*)deg, N, off);
  hipLaunchKernelGGL(pn_rev_fill_lds_kernel<IT>, dim3(G, B, nwin), dim3(256), lds, stream, idx, N, k, G,
                     (const int*)part, (const int*)off, rev);
  hipLaunchKernelGGL(pn_rev_sort_kernel, dim3(pn_cdiv(N, 4), B), dim3(256), 0, stream, (const int*)off,
                     (const int*)part, N, k, G, rev);
  PN_CHECK_LAUNCH();
  *off_out = off;
  *rev_out = rev;
  return PN_OK;
}

// dP row of one target point j, one wave per point, in list order:
//   dP[j,c] = [dense] ( -r*deg_j*(c1 + c2*r*(P[j,c]-mu)) - r^2*c2*sum_{(i->j)} Q[i,c] )
//             + r * sum_{(i,slot)->j, argk[i,c] == slot} t[i,c]
// (the second sum: the edges that are the extreme of their source point in channel c).
template <int COUT>
__global__ __launch_bounds__(256) void pn_edgeconv_bwd_gather_kernel(
    const float* __restrict__ PQ, const int* __restrict__ off, const uint32_t* __restrict__ rev,
    const float* __restrict__ t, const uint8_t* __restrict__ argk, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ c1c2, int N, int k, int Cg, int per_sample,
    int dense, int nblk, int B, float* __restrict__ dPQ) {
  constexpr int NCH = COUT <= 256 ? 1 : COUT / 256;
  constexpr int LPR = COUT <= 256 ? COUT / 4 : 64;
  constexpr int RPI = 64 / LPR;
  const int vid = pn_xcd_virtual(blockIdx.x, nblk * B);     // (one item per XCD at a time, see above)
  if (vid >= nblk * B) return;
  const int b = vid / nblk, blk = vid - b * nblk;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j = blk * 4 + wave;
  if (j >= N) return;
  const int rg = lane / LPR, cl = lane - rg * LPR;
  const int G = COUT / Cg;
  const float* __restrict__ PQb = PQ + (size_t)b * N * 2 * COUT;
  const float* __restrict__ tb = t + (size_t)b * N * COUT;
  const uint8_t* __restrict__ ab = argk + (size_t)b * N * COUT;
  const int* __restrict__ ob = off + (size_t)b * (N + 1);
  const uint32_t* __restrict__ rb = rev + (size_t)b * N * k;
  const int e0 = ob[j], e1 = ob[j + 1];
  float4x acc[NCH], ext[NCH];
#pragma unroll
  for (int h = 0; h < NCH; ++h)
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[h].v[u] = ext[h].v[u] = 0.f;
  // the source list of the point 64 entries at a time in registers (one coalesced load), two row
  // gathers in flight per trip: a fixed summation order per lane (entries rg, rg + RPI, ... ascending)
  for (int base = e0; base < e1; base += 64) {
    const uint32_t mine = base + lane < e1 ? rb[base + lane] : ((uint32_t)j << 8) | 0xffu;
    const int cnt = e1 - base < 64 ? e1 - base : 64;
    for (int t0 = 0; t0 < cnt; t0 += 2 * RPI) {
      const int ta = t0 + rg, tb_ = ta + RPI;
      const uint32_t sa = (uint32_t)__shfl((int)mine, ta & 63, 64), sb = (uint32_t)__shfl((int)mine, tb_ & 63, 64);
      const bool oa = ta < cnt, ob_ = tb_ < cnt;
      const int ia = oa ? (int)(sa >> 8) : j, ib_ = ob_ ? (int)(sb >> 8) : j;
      const uint32_t ka = oa ? (sa & 255u) : 0x100u, kb = ob_ ? (sb & 255u) : 0x100u;   // 0x100: matches no slot
      float4x qa[NCH], qb[NCH], va[NCH], vb[NCH];
      uint32_t aa[NCH], abv[NCH];
#pragma unroll
      for (int h = 0; h < NCH; ++h) {
        const int c0 = (cl + h * 64) * 4;
        if (dense) {
          qa[h] = ld4(PQb + (size_t)ia * 2 * COUT + COUT + c0);
          qb[h] = ld4(PQb + (size_t)ib_ * 2 * COUT + COUT + c0);
        }
        va[h] = ld4(tb + (size_t)ia * COUT + c0);
        vb[h] = ld4(tb + (size_t)ib_ * COUT + c0);
        aa[h] = *reinterpret_cast<const uint32_t*>(ab + (size_t)ia * COUT + c0);
        abv[h] = *reinterpret_cast<const uint32_t*>(ab + (size_t)ib_ * COUT + c0);
      }
#pragma unroll
      for (int h = 0; h < NCH; ++h)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (dense) {
            if (oa) acc[h].v[u] += qa[h].v[u];
            if (ob_) acc[h].v[u] += qb[h].v[u];
          }
          if (((aa[h] >> (8 * u)) & 255u) == ka) ext[h].v[u] += va[h].v[u];
          if (((abv[h] >> (8 * u)) & 255u) == kb) ext[h].v[u] += vb[h].v[u];
        }
    }
  }
#pragma unroll
  for (int o = LPR; o < 64; o <<= 1)
#pragma unroll
    for (int h = 0; h < NCH; ++h)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc[h].v[u] += __shfl_xor(acc[h].v[u], o, 64);
        ext[h].v[u] += __shfl_xor(ext[h].v[u], o, 64);
      }
  if (rg == 0) {
    const float deg = (float)(e1 - e0);
#pragma unroll
    for (int h = 0; h < NCH; ++h) {
      const int c0 = (cl + h * 64) * 4;
      const float4x p = ld4(PQb + (size_t)j * 2 * COUT + c0);
      float4x o;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int sidx = (per_sample ? b : 0) * G + (c0 + u) / Cg;
        const float mu = mean[sidx], r = rstd[sidx];
        const float c1 = c1c2[2 * sidx], c2 = c1c2[2 * sidx + 1];
        const float dn = dense ? -r * deg * (c1 + c2 * r * (p.v[u] - mu)) - r * r * c2 * acc[h].v[u] : 0.f;
        o.v[u] = dn + r * ext[h].v[u];
      }
      st4(dPQ + ((size_t)b * N + j) * 2 * COUT + c0, o);
    }
  }
}

// any width: one thread per (target point, channel), the same sums in the same order
__global__ __launch_bounds__(256) void pn_edgeconv_bwd_gather_generic_kernel(
    const float* __restrict__ PQ, const int* __restrict__ off, const uint32_t* __restrict__ rev,
    const float* __restrict__ t, const uint8_t* __restrict__ argk, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ c1c2, int N, int k, int Cout, int Cg,
    int per_sample, int dense, float* __restrict__ dPQ) {
  const int b = blockIdx.y;
  const long long tix = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (tix >= (long long)N * Cout) return;
  const int j = (int)(tix / Cout), c = (int)(tix - (long long)j * Cout);
  const int G = Cout / Cg;
  const float* __restrict__ PQb = PQ + (size_t)b * N * 2 * Cout;
  const float* __restrict__ tb = t + (size_t)b * N * Cout;
  const uint8_t* __restrict__ ab = argk + (size_t)b * N * Cout;
  const int* __restrict__ ob = off + (size_t)b * (N + 1);
  const uint32_t* __restrict__ rb = rev + (size_t)b * N * k;
  const int e0 = ob[j], e1 = ob[j + 1];
  float acc = 0.f, ext = 0.f;
  for (int e = e0; e < e1; ++e) {
    const uint32_t s = rb[e];
    const int i = (int)(s >> 8);
    if (dense) acc += PQb[(size_t)i * 2 * Cout + Cout + c];
    if ((uint32_t)ab[(size_t)i * Cout + c] == (s & 255u)) ext += tb[(size_t)i * Cout + c];
  }
  const int sidx = (per_sample ? b : 0) * G + c / Cg;
  const float mu = mean[sidx], r = rstd[sidx];
  const float c1 = c1c2[2 * sidx], c2 = c1c2[2 * sidx + 1];
  const float p = PQb[(size_t)j * 2 * Cout + c];
  const float dn = dense ? -r * (float)(e1 - e0) * (c1 + c2 * r * (p - mu)) - r * r * c2 * acc : 0.f;
  dPQ[((size_t)b * N + j) * 2 * Cout + c] = dn + r * ext;
}

// per source point: dQ[i] (closed form)
__global__ __launch_bounds__(256) void pn_edgeconv_bwd_point_kernel(
    const float* __restrict__ t, const float* __restrict__ s1, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ c1c2, int N, int k, int Cout, int Cg,
    int per_sample, float* __restrict__ dPQ) {
  const int b = blockIdx.y;
  const long long tix = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (tix >= (long long)N * Cout) return;
  const int i = (int)(tix / Cout), c = (int)(tix - (long long)i * Cout);
  const int G = Cout / Cg;
  const int sidx = (per_sample ? b : 0) * G + c / Cg;
  const float mu = mean[sidx], r = rstd[sidx];
  const float c1 = c1c2[2 * sidx], c2 = c1c2[2 * sidx + 1];
  const size_t o = ((size_t)b * N + i) * Cout + c;
  const float fk = (float)k;
  dPQ[((size_t)b * N + i) * 2 * Cout + Cout + c] = r * (t[o] - fk * c1 - c2 * r * (s1[o] - fk * mu));
}

// where pn_build_rev_csr leaves the transposed graph inside a workspace of pn_edgeconv_bwd_workspace(B, N, k) bytes
static inline void pn_rev_csr_in_workspace(void* workspace, int B, int N, const int** off, const uint32_t** rev) {
  char* w = (char*)workspace + pn_align_up((size_t)B * N * 4, 256);
  *off = (const int*)w;
  *rev = (const uint32_t*)(w + pn_align_up((size_t)B * (N + 1) * 4, 256));
}

// ``prebuilt``: the workspace already holds the transposed graph of idx (pn_edgeconv_csr_build_*): the graph depends
// on idx alone, so a caller can build it during the FORWARD pass on a side stream, off the backward's critical path
template <typename IT>
static int edgeconv_bwd(const float* PQ, const IT* idx, const float* t, const float* s1, const uint8_t* argk,
                        const float* mean, const float* rstd, const float* c1c2, int B, int N, int k, int Cout,
                        int groups, int per_sample, int dense, float* dPQ, void* workspace, size_t workspace_bytes,
                        void* stream_, int prebuilt = 0) {
  hipStream_t stream = (hipStream_t)stream_;
  PN_CHECK_ARG(PQ && idx && t && s1 && argk && mean && rstd && c1c2 && dPQ,
               "pn_edgeconv_bwd: null pointer");
  PN_CHECK_ARG(groups > 0 && Cout % groups == 0, "pn_edgeconv_bwd: groups=%d", groups);
  const int Cg = Cout / groups;
  const int* off = nullptr;
  const uint32_t* rev = nullptr;
  if (prebuilt) {
    PN_CHECK_ARG(workspace && workspace_bytes >= pn_edgeconv_bwd_workspace(B, N, k),
                 "pn_edgeconv_bwd: workspace too small for a prebuilt graph");
    pn_rev_csr_in_workspace(workspace, B, N, &off, &rev);
  } else {
    PN_PROF("edgeconv_bwd_csr", stream);
    const int rc = pn_build_rev_csr<IT>(idx, B, N, k, workspace, workspace_bytes, stream, &off, &rev);
    if (rc != PN_OK) return rc;
  }
  PN_PROF("edgeconv_bwd", stream);
  const int nblk = pn_cdiv(N, 4);
  dim3 grid(pn_xcd_grid(nblk * B));
#define EC_BG(CO)                                                                               \
  hipLaunchKernelGGL(pn_edgeconv_bwd_gather_kernel<CO>, grid, dim3(256), 0, stream, PQ, off, rev, \
                     t, argk, mean, rstd, c1c2, N, k, Cg, per_sample, dense, nblk, B, dPQ)
  dim3 g2(pn_cdiv((long long)N * Cout, 256), B);
  if (Cout == 64)
    EC_BG(64);
  else if (Cout == 128)
    EC_BG(128);
  else if (Cout == 256)
    EC_BG(256);
  else if (Cout == 512)
    EC_BG(512);
  else
    hipLaunchKernelGGL(pn_edgeconv_bwd_gather_generic_kernel, g2, dim3(256), 0, stream, PQ, off, rev, t, argk,
                       mean, rstd, c1c2, N, k, Cout, Cg, per_sample, dense, dPQ);
#undef EC_BG
  hipLaunchKernelGGL(pn_edgeconv_bwd_point_kernel, g2, dim3(256), 0, stream, t, s1, mean, rstd, c1c2, N, k, Cout,
                     Cg, per_sample, dPQ);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

extern "C" int pn_edgeconv_bwd_f32(const float* PQ, const int64_t* idx, const float* t,
                                   const float* s1, const uint8_t* argk, const float* mean,
                                   const float* rstd, const float* c1c2, int B, int N, int k,
                                   int Cout, int groups, int per_sample, int dense, float* dPQ,
                                   void* workspace, size_t workspace_bytes, void* stream_) {
  return edgeconv_bwd<int64_t>(PQ, idx, t, s1, argk, mean, rstd, c1c2, B, N, k, Cout, groups, per_sample, dense,
                               dPQ, workspace, workspace_bytes, stream_);
}
// The transposed graph of idx into ``workspace`` (pn_edgeconv_bwd_workspace(B, N, k) bytes), for
// pn_edgeconv_bwd_prebuilt_*; idx_is_i32 selects the index width.
extern "C" int pn_edgeconv_csr_build(const void* idx, int idx_is_i32, int B, int N, int k, void* workspace,
                                     size_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PN_CHECK_ARG(idx && B > 0 && N > 0 && k > 0, "pn_edgeconv_csr_build: bad arguments");
  const int* off = nullptr;
  const uint32_t* rev = nullptr;
  PN_PROF("edgeconv_bwd_csr", stream);
  return idx_is_i32 ? pn_build_rev_csr<int32_t>((const int32_t*)idx, B, N, k, workspace, workspace_bytes, stream, &off, &rev)
                    : pn_build_rev_csr<int64_t>((const int64_t*)idx, B, N, k, workspace, workspace_bytes, stream, &off, &rev);
}

extern "C" int pn_edgeconv_bwd_prebuilt(const float* PQ, const void* idx, int idx_is_i32, const float* t, const float* s1,
                                        const uint8_t* argk, const float* mean, const float* rstd, const float* c1c2,
                                        int B, int N, int k, int Cout, int groups, int per_sample, int dense, float* dPQ,
                                        void* workspace, size_t workspace_bytes, void* stream_) {
  return idx_is_i32 ? edgeconv_bwd<int32_t>(PQ, (const int32_t*)idx, t, s1, argk, mean, rstd, c1c2, B, N, k, Cout, groups,
                                            per_sample, dense, dPQ, workspace, workspace_bytes, stream_, 1)
                    : edgeconv_bwd<int64_t>(PQ, (const int64_t*)idx, t, s1, argk, mean, rstd, c1c2, B, N, k, Cout, groups,
                                            per_sample, dense, dPQ, workspace, workspace_bytes, stream_, 1);
}

// the same on the library's int32 graph (pn_knn_graph_i32)
extern "C" int pn_edgeconv_bwd_i32(const float* PQ, const int32_t* idx, const float* t,
                                   const float* s1, const uint8_t* argk, const float* mean,
                                   const float* rstd, const float* c1c2, int B, int N, int k,
                                   int Cout, int groups, int per_sample, int dense, float* dPQ,
                                   void* workspace, size_t workspace_bytes, void* stream_) {
  return edgeconv_bwd<int32_t>(PQ, idx, t, s1, argk, mean, rstd, c1c2, B, N, k, Cout, groups, per_sample, dense,
                               dPQ, workspace, workspace_bytes, stream_);
}

// ------------------------------------------------------------------------------------
// backward of the API form of get_graph_feature (gathering over the transposed graph):
//   gxt[b,j,:] = sum_{(i,slot) -> j} g[b,i,slot,0:C]  +  sum_kk (g[b,j,kk,C:2C] - g[b,j,kk,0:C])
// one wave per target point, lanes over the channels; list order = (source, slot) ascending.
// ------------------------------------------------------------------------------------
// CW = lanes per row (the power of two >= min(C, 64)); the 64 / CW lane groups of a wave take the list
// entries (and the neighbour slots of the centre term) t = group, group + 64 / CW, ... in ascending order,
// two row loads in flight per lane, and are added by a fixed xor tree at the end: a fixed summation order.
template <int CW>
__global__ __launch_bounds__(256) void pn_edge_feature_bwd_kernel(
    const float* __restrict__ g, const int* __restrict__ off, const uint32_t* __restrict__ rev, int N, int k,
    int C, float* __restrict__ gxt) {
  constexpr int G = 64 / CW;
  const int b = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j = blockIdx.x * 4 + wave;
  if (j >= N) return;
  const int grp = lane / CW, cl = lane - grp * CW;
  const float* __restrict__ gb = g + (size_t)b * N * k * 2 * C;
  const int* __restrict__ ob = off + (size_t)b * (N + 1);
  const uint32_t* __restrict__ rb = rev + (size_t)b * N * k;
  const int e0 = ob[j], e1 = ob[j + 1];
  const float* __restrict__ gj = gb + (size_t)j * k * 2 * C;
  for (int c0 = 0; c0 < C; c0 += CW) {
    const int c = c0 + cl;
    const bool cin = c < C;
    const int cc = cin ? c : 0;
    float acc = 0.f;
    // centre term: sum_kk (g[j,kk,C+c] - g[j,kk,c])
    for (int kk = grp; kk < k; kk += 2 * G) {
      const int kb = kk + G;
      const float a1 = gj[(size_t)kk * 2 * C + C + cc], a0 = gj[(size_t)kk * 2 * C + cc];
      const bool ob2 = kb < k;
      const float b1 = ob2 ? gj[(size_t)kb * 2 * C + C + cc] : 0.f, b0 = ob2 ? gj[(size_t)kb * 2 * C + cc] : 0.f;
      acc += a1 - a0;
      if (ob2) acc += b1 - b0;
    }
    // incoming edges: 64 list entries at a time in registers
    for (int base = e0; base < e1; base += 64) {
      const uint32_t mine = base + lane < e1 ? rb[base + lane] : 0u;
      const int cnt = e1 - base < 64 ? e1 - base : 64;
      for (int t0 = 0; t0 < cnt; t0 += 2 * G) {          // (wave-uniform trips: the shuffles are executed by every lane)
        const int ta = t0 + grp, tb = ta + G;
        const uint32_t sa = (uint32_t)__shfl((int)mine, ta & 63, 64), sb = (uint32_t)__shfl((int)mine, tb & 63, 64);
        const bool oa2 = ta < cnt, ob2 = tb < cnt;
        const float va = oa2 ? gb[((size_t)(sa >> 8) * k + (sa & 255u)) * 2 * C + cc] : 0.f;
        const float vb = ob2 ? gb[((size_t)(sb >> 8) * k + (sb & 255u)) * 2 * C + cc] : 0.f;
        if (oa2) acc += va;
        if (ob2) acc += vb;
      }
    }
#pragma unroll
    for (int o = CW; o < 64; o <<= 1) acc += __shfl_xor(acc, o, 64);
    if (grp == 0 && cin) gxt[((size_t)b * N + j) * C + c] = acc;
  }
}

extern "C" int pn_edge_feature_bwd_f32(const float* gfeat, const int64_t* idx, int B, int N, int k,
                                       int C, float* gxt, void* workspace, size_t workspace_bytes,
                                       void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PN_CHECK_ARG(gfeat && idx && gxt, "pn_edge_feature_bwd_f32: null pointer");
  PN_CHECK_ARG(B > 0 && N > 0 && k > 0 && C > 0, "pn_edge_feature_bwd_f32: empty input");
  const int* off = nullptr;
  const uint32_t* rev = nullptr;
  PN_PROF("edge_feature_bwd", stream);
  const int rc = pn_build_rev_csr<int64_t>(idx, B, N, k, workspace, workspace_bytes, stream, &off, &rev);
  if (rc != PN_OK) return rc;
  dim3 grid(pn_cdiv(N, 4), B);
#define EF_BWD(CW_) \
  hipLaunchKernelGGL(pn_edge_feature_bwd_kernel<CW_>, grid, dim3(256), 0, stream, gfeat, off, rev, N, k, C, gxt)
  if (C <= 4)
    EF_BWD(4);
  else if (C <= 8)
    EF_BWD(8);
  else if (C <= 16)
    EF_BWD(16);
  else if (C <= 32)
    EF_BWD(32);
  else
    EF_BWD(64);
#undef EF_BWD
  PN_CHECK_LAUNCH();
  return PN_OK;
}
