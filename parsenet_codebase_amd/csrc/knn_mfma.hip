// Exact k-best selection on the fp32 matrix cores (gfx950): kNN graphs, bandwidth quantiles,
// nearest-centre assignment.
//
// Replaces, without ever materialising an N x N matrix,
//   src/model.py:9-22, src/PointNet.py:9-26, :29-69   (N x N GEMM + torch.topk -> kNN graph)
//   src/mean_shift.py:125-137   (2 - 2 X X^T + topk(K, largest=False) -> K-th distance per row)
//   src/mean_shift.py:146-149   (argmin over centres of 2 - 2 C X^T -> membership)
//
// The values are k-ordered fp32 fma chains; v_mfma_f32_32x32x2_f32 evaluates exactly such a
// chain (D = fma(a_k1, b_k1, fma(a_k0, b_k0, C)), one rounding per product, no wider
// accumulation), so the matrix-core result is bit-identical to the oracle's scalar loop.
// A 32(candidates) x 32(queries) tile is accumulated per wave with the query operands resident
// in VGPRs; candidate operands stream from L2 with one dword per lane per k-step (two 128-byte
// segments per load).
//
// Selection never sorts more than a handful of values per query:
//   K0  gather the candidates into a decorrelated order (golden-ratio affine bijection; tile
//       statistics then do not depend on how the caller ordered the points), pad to
//       multiples of 64 points / 2 channels, compute squared norms.
//   K1  pass 1: for every query and every group of 16 candidates keep only the maximum
//       value ("tile maximum").  The k-th largest tile maximum T_k is a valid threshold:
//       at least k distinct candidates have value >= T_k.
//   K2  wave-per-query bisection on the register-resident tile maxima -> tau
//       (count(tilemax >= tau) >= k, as close to k as the bisection gets).  With N/16 tiles
//       the expected number of candidates >= tau is ~1.07 k for k << N/16.
//   K3  pass 2: recompute the values (same arithmetic), append the survivors (v >= tau) as
//       64-bit (value, index) keys to a sub-list private to (query, slice, half-wave): no
//       atomics, the fill count lives in a register.
//   K4  one wave per query: gather the sub-lists, select/sort, emit either the k best
//       indices (best first, ties -> smaller original index) or the k-th best value.
//   Lists that overflow (degenerate inputs: masses of exactly equal values) are flagged; for
//   kNN graphs they are recomputed on the device by the generic scan kernel (knn.hip),
//   gated on the flags; the other entry points hand the flags to the caller.
//
// Value semantics (each operation rounded to fp32, in the reference's order):
//   MODE 0:  v = (-xx[j] - (-2*dot)) - xx[i]                      (model.py:14-16)
//   MODE 1:  p = (xxp[j] - 2*dot_p) + xxp[i]; n = 2 - 2*dot_n;
//            v = -(p * (1 + n))                                     (PointNet.py:41-59)
//   MODE 2:  v = dot(q_i, c_j)     (mean_shift.py: 2 - 2*dot is a decreasing, exact map)
#include "knn_common.h"
#include "split_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

size_t pn_knn_v1_workspace(int B, int C, int N, int k, bool gated);
int pn_knn_v1_launch(int mode, const float* x, int B, int C, int N, int k, KnnIdxOut idx,
                     void* workspace, size_t workspace_bytes, hipStream_t stream,
                     const int* gate, const int* gate_any = nullptr);

struct KnnPerm {
  uint32_t a, c, mask;
  int n;
};

// bijection on [0, n): affine map modulo 2^m (odd multiplier near 2^m / golden ratio) with
// cycle walking.  Consecutive positions land ~0.618 * 2^m apart.  a = 1, c = 0: identity.
__host__ __device__ static inline int knn_perm(const KnnPerm& p, int i) {
  uint32_t v = (uint32_t)i;
  do {
    v = (v * p.a + p.c) & p.mask;
  } while (v >= (uint32_t)p.n);
  return (int)v;
}

static KnnPerm knn_make_perm(int N, bool identity) {
  KnnPerm p;
  int m = 1;
  while ((1u << m) < (uint32_t)N) ++m;
  p.mask = (m >= 32) ? 0xffffffffu : ((1u << m) - 1u);
  p.a = identity ? 1u : (((uint32_t)(0.6180339887498949 * (double)(1ull << m))) | 1u);
  p.c = identity ? 0u : ((0x9E3779B9u >> (32 - m)) | 1u);
  p.n = N;
  return p;
}

// K0: xp (B, Cp, Np) zero padded, permuted columns; xxp (B, Np) squared norms (fma chain over
// the first `cnorm` source channels).  MODE 1 channel map: [p0 p1 p2 0 n0 n1 n2 0].
// Source layout: channel-first (B,C,N) or point-major (B,N,C).
__global__ void pn_knn_prep_kernel(const float* __restrict__ x, int C, int N, int Cp, int Np,
                                   int mode, int point_major, KnnPerm perm,
                                   float* __restrict__ xp, float* __restrict__ xxp) {
  const int b = blockIdx.y;
  const int jp = blockIdx.x * blockDim.x + threadIdx.x;
  if (jp >= Np) return;
  const float* xb = x + (size_t)b * C * N;
  float* xpb = xp + (size_t)b * Cp * Np;
  float acc = 0.f;
  if (jp < N) {
    const int j = knn_perm(perm, jp);
    const int cnorm = mode == 1 ? 3 : C;
    // (eight channels at a time: independent loads ahead of the norm's fma chain)
    int c = 0;
    for (; c + 8 <= C; c += 8) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = point_major ? xb[(size_t)j * C + c + e] : xb[(size_t)(c + e) * N + j];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int cd = (mode == 1 && c + e >= 3) ? c + e + 1 : c + e;
        xpb[(size_t)cd * Np + jp] = v[e];
        if (c + e < cnorm) acc = __builtin_fmaf(v[e], v[e], acc);
      }
    }
    for (; c < C; ++c) {
      const float v = point_major ? xb[(size_t)j * C + c] : xb[(size_t)c * N + j];
      const int cd = (mode == 1 && c >= 3) ? c + 1 : c;
      xpb[(size_t)cd * Np + jp] = v;
      if (c < cnorm) acc = __builtin_fmaf(v, v, acc);
    }
    if (mode == 1) {
      xpb[(size_t)3 * Np + jp] = 0.f;
      xpb[(size_t)7 * Np + jp] = 0.f;
    } else {
      for (int c = C; c < Cp; ++c) xpb[(size_t)c * Np + jp] = 0.f;
    }
  } else {
    for (int c = 0; c < Cp; ++c) xpb[(size_t)c * Np + jp] = 0.f;
  }
  xxp[(size_t)b * Np + jp] = acc;
}

// K0 for point-major sources (B,N,C) with a permutation (the mean-shift bandwidth / membership
// queries: C = 128): the generic kernel above reads one 4-byte element per lane from 64 different
// rows (250 us at N = 10 000).  Here a workgroup takes 64 output columns, reads their source rows
// with 16-byte lanes along the row, turns the tile in LDS (row stride C + 1: conflict free) and
// writes the channel-first rows 256 bytes at a time.  Norms stay a per-point fma chain in channel
// order (their bits decide the ranking).
__global__ __launch_bounds__(256) void pn_knn_prep_pm_kernel(const float* __restrict__ x, int C, int N,
                                                            int Cp, int Np, KnnPerm perm,
                                                            float* __restrict__ xp,
                                                            float* __restrict__ xxp) {
  extern __shared__ float prep_tile[];  // [64][C + 1]
  const int b = blockIdx.y;
  const int jp0 = blockIdx.x * 64;
  const int t = threadIdx.x;
  const float* xb = x + (size_t)b * C * N;
  float* xpb = xp + (size_t)b * Cp * Np;
  const int C4 = C >> 2, ld = C + 1;
  for (int e = t; e < 64 * C4; e += 256) {
    const int p = e / C4, q = e - p * C4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (jp0 + p < N) v = *reinterpret_cast<const float4*>(xb + (size_t)knn_perm(perm, jp0 + p) * C + 4 * q);
    float* d = prep_tile + p * ld + 4 * q;
    d[0] = v.x;
    d[1] = v.y;
    d[2] = v.z;
    d[3] = v.w;
  }
  __syncthreads();
  const int p = t & 63, part = t >> 6;
  if (jp0 + p < Np) {
    for (int c = part; c < Cp; c += 4) xpb[(size_t)c * Np + jp0 + p] = c < C ? prep_tile[p * ld + c] : 0.f;
    if (part == 0) {
      float acc = 0.f;
      for (int c = 0; c < C; ++c) {
        const float v = prep_tile[p * ld + c];
        acc = __builtin_fmaf(v, v, acc);
      }
      xxp[(size_t)b * Np + jp0 + p] = acc;
    }
  }
}

static void knn_prep_launch(hipStream_t stream, const float* x, int B, int C, int N, int Cp, int Np, int mode,
                            int point_major, KnnPerm perm, float* xp, float* xxp) {
  if (point_major && mode != 1 && (C & 3) == 0 && C <= 224) {  // 64 x (C+1) floats of LDS
    hipLaunchKernelGGL(pn_knn_prep_pm_kernel, dim3(pn_cdiv(Np, 64), B), dim3(256), 64 * (C + 1) * sizeof(float),
                       stream, x, C, N, Cp, Np, perm, xp, xxp);
  } else {
    // (64-thread workgroups: a thread walks all channels of its column; 4 x 10 000 columns are 628 workgroups, not 160)
    hipLaunchKernelGGL(pn_knn_prep_kernel, dim3(pn_cdiv(Np, 64), B), dim3(64), 0, stream, x, C, N, Cp, Np,
                       mode, point_major, perm, xp, xxp);
  }
}

// K1 / K3.  KSTEPS = Cp / 2 (MODE 1: 4 = two steps xyz + two steps normals), QSETS = number of
// 32-query column blocks per wave.  Queries (xq, Nq valid of Nqp padded) and candidates
// (xc, Nc of Ncp) may be the same array.
// KIND 0: tile maxima (pass 1); 1: collect survivors (pass 2); 2: single-pass arg-max (k = 1,
// index only): every lane keeps the best key of its (query, slice, half) — value first, then the
// smaller ORIGINAL candidate index — and writes it to lists[((b*Nqp+q)*S+slice)*2+h].
template <int KSTEPS, int QSETS, int MODE, int KIND>
__global__ __launch_bounds__(256) void pn_knn_mfma_kernel(
    const float* __restrict__ xq, const float* __restrict__ xxq_, int Nq, int Nqp,
    const float* __restrict__ xc, const float* __restrict__ xxc_, int Nc, int Ncp,
    int tiles_per_slice, float* __restrict__ tilemax, const float* __restrict__ tau,
    u64* __restrict__ lists, int* __restrict__ counts, int subcap, KnnPerm perm_c) {
  constexpr bool COLLECT = KIND == 1;
  constexpr bool ARGMAX = KIND == 2;
  // grid: (slices, query blocks, B): consecutive workgroups differ in the slice, so each of the
  // 8 XCDs streams only its share of the candidates (kept in its private L2).
  // The candidate tile [CP channels][32 candidates] (+ the 32 squared norms) is staged through
  // LDS once per block, double buffered, and shared by the four waves (different queries).
  constexpr int CP = 2 * KSTEPS;
  constexpr int TILE = CP * 32;            // floats
  __shared__ __attribute__((aligned(16))) float lds[2][TILE + 32];
  const int b = blockIdx.z;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int col = lane & 31, h = lane >> 5;
  const int q0 = (blockIdx.y * 4 + wave) * (32 * QSETS);
  const bool wave_on = q0 < Nq;
  const float* __restrict__ xqb = xq + (size_t)b * CP * Nqp;
  const float* __restrict__ xcb = xc + (size_t)b * CP * Ncp;
  const float* __restrict__ xxqb = xxq_ + (size_t)b * Nqp;
  const float* __restrict__ xxcb = xxc_ + (size_t)b * Ncp;
  const int ntiles = Ncp / 32;
  const int S = gridDim.x, slice = blockIdx.x;
  const int t_begin = slice * tiles_per_slice;
  const int t_end = min(ntiles, t_begin + tiles_per_slice);
  const int T16 = Ncp / 16;

  // resident query operands: B[k = lane>>5][j = lane&31] of every k-step
  float bq[QSETS][KSTEPS];
  float xxq[QSETS], tq[QSETS];
  int mycnt[QSETS];
  u64* sub[QSETS];
  int bestj[QSETS];
  float bestv[QSETS];
#pragma unroll
  for (int s = 0; s < QSETS; ++s) {
    const int q = q0 + 32 * s + col;
    const int qcl = q < Nqp ? q : Nqp - 1;
#pragma unroll
    for (int m = 0; m < KSTEPS; ++m) bq[s][m] = xqb[(size_t)(2 * m + h) * Nqp + qcl];
    xxq[s] = xxqb[qcl];
    tq[s] = (COLLECT && q < Nq) ? tau[(size_t)b * Nqp + qcl] : __builtin_inff();
    mycnt[s] = 0;
    bestj[s] = 0;
    bestv[s] = -__builtin_inff();
    // sub-list of (query, slice, half): (((b*Nqp + q)*S + slice)*2 + h) * subcap
    sub[s] = lists + ((((size_t)b * Nqp + qcl) * S + slice) * 2 + h) * (size_t)subcap;
  }

  // stage with the LDS DMA: one wave instruction copies a 1 KiB chunk (8 channel rows x 32
  // candidates) from per-lane global addresses into the lane-linear LDS image; no staging VGPRs
  typedef const __attribute__((address_space(1))) void* km_gptr;
  typedef __attribute__((address_space(3))) void* km_lptr;
  constexpr int NCHUNK = (TILE + 255) / 256;
#define KM_STAGE(MT, BUF)                                                                        \
  {                                                                                              \
    const int j0s = (MT) * 32;                                                                   \
    _Pragma("unroll") for (int q = 0; q < NCHUNK; ++q) {                                         \
      if ((q & 3) == wave) {                                                                     \
        const int row = q * 8 + (lane >> 3);                                                     \
        if (row < CP)                                                                            \
          __builtin_amdgcn_global_load_lds((km_gptr)(xcb + (size_t)row * Ncp + j0s + ((lane & 7) << 2)), \
                                           (km_lptr)(&lds[BUF][q * 256]), 16, 0, 0);             \
      }                                                                                          \
    }                                                                                            \
    if (MODE != 2 && wave == 3 && lane < 32)                                                     \
      __builtin_amdgcn_global_load_lds((km_gptr)(xxcb + j0s + lane), (km_lptr)(&lds[BUF][TILE]), 4, 0, 0); \
  }
  int cur = 0;
  if (t_begin < t_end) KM_STAGE(t_begin, 0);
  __syncthreads();
  for (int mt = t_begin; mt < t_end; ++mt) {
    const int j0 = mt * 32;
    const bool has_next = mt + 1 < t_end;
    if (has_next) KM_STAGE(mt + 1, cur ^ 1);  // DMA in flight while this tile is computed
    if (wave_on) {
      const float* __restrict__ lx = lds[cur];
      f32x16 acc[QSETS], accn[MODE == 1 ? QSETS : 1];
#pragma unroll
      for (int s = 0; s < QSETS; ++s) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;
        if (MODE == 1) {
#pragma unroll
          for (int r = 0; r < 16; ++r) accn[s][r] = 0.f;
        }
      }
      // A[i = lane&31][k = lane>>5]: candidate j0+col, channel 2m+h
#pragma unroll
      for (int m = 0; m < KSTEPS; ++m) {
        const float a = lx[(2 * m + h) * 32 + col];
#pragma unroll
        for (int s = 0; s < QSETS; ++s) {
          if (MODE == 1 && m >= 2)
            accn[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bq[s][m], accn[s], 0, 0, 0);
          else
            acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bq[s][m], acc[s], 0, 0, 0);
        }
      }
      // D[i][j]: lane holds column j = col (query), rows i = (r&3) + 8*(r>>2) + 4*h (candidates)
      float xxj[16];
      if (MODE != 2) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 t4 = *reinterpret_cast<const float4*>(&lx[TILE + 8 * g + 4 * h]);
          xxj[4 * g + 0] = t4.x;
          xxj[4 * g + 1] = t4.y;
          xxj[4 * g + 2] = t4.z;
          xxj[4 * g + 3] = t4.w;
        }
      }
#pragma unroll
      for (int s = 0; s < QSETS; ++s) {
        float tm = -__builtin_inff();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
          float v;
          if (MODE == 0) {
            // (-xx[j] - (-2*dot)) - xx[i]; 2*dot is exact, so the fma rounds once like the
            // reference's subtraction
            const float t = __builtin_fmaf(2.0f, acc[s][r], -xxj[r]);
            v = __fsub_rn(t, xxq[s]);
          } else if (MODE == 1) {
            const float t = __builtin_fmaf(-2.0f, acc[s][r], xxj[r]);  // xx[j] - 2*dot_p
            const float pp = __fadd_rn(t, xxq[s]);
            const float pn = __builtin_fmaf(-2.0f, accn[s][r], 2.0f);  // 2 - 2*dot_n
            v = -__fmul_rn(pp, __fadd_rn(1.0f, pn));
          } else {
            v = acc[s][r];
          }
          const bool ok = (j0 + row) < Nc;
          if (ARGMAX) {
            // candidates come in ORIGINAL order in this mode (identity permutation) and a lane
            // visits its rows in increasing index order: a strict > keeps the smallest index
            // among exact ties, which near-coincident mean-shift modes produce by the thousand
            if (ok && v > bestv[s]) {
              bestj[s] = j0 + row;
              bestv[s] = v;
            }
          } else if (!COLLECT) {
            tm = fmaxf(tm, ok ? v : -__builtin_inff());
          } else {
            if (ok && v >= tq[s]) {
              // the key carries the PERMUTED index here; K4 rewrites it to the original one
              if (mycnt[s] < subcap) sub[s][mycnt[s]] = knn_key(v, j0 + row);
              ++mycnt[s];
            }
          }
        }
        if (KIND == 0) {
          const int q = q0 + 32 * s + col;
          if (q < Nqp) tilemax[((size_t)b * Nqp + q) * T16 + (2 * mt + h)] = tm;
        }
      }
    }
    __syncthreads();
    cur ^= 1;
  }
#undef KM_STAGE
  if (COLLECT && wave_on) {
#pragma unroll
    for (int s = 0; s < QSETS; ++s) {
      const int q = q0 + 32 * s + col;
      if (q < Nqp) counts[(((size_t)b * Nqp + q) * S + slice) * 2 + h] = mycnt[s];
    }
  }
  if (ARGMAX && wave_on) {
#pragma unroll
    for (int s = 0; s < QSETS; ++s) {
      const int q = q0 + 32 * s + col;
      if (q < Nqp)
        lists[(((size_t)b * Nqp + q) * S + slice) * 2 + h] =
            t_begin < t_end ? knn_key(bestv[s], knn_perm(perm_c, bestj[s])) : 0;
    }
  }
}

// arg-max epilogue: one thread per query, maximum of its 2 S partial keys
__global__ void pn_knn_argmax_final_kernel(const u64* __restrict__ lists, int Nq, int Nqp, int S,
                                           KnnPerm perm_q, KnnIdxOut out_idx) {
  const int b = blockIdx.y;
  const int qp = blockIdx.x * blockDim.x + threadIdx.x;
  if (qp >= Nq) return;
  const u64* l = lists + ((size_t)b * Nqp + qp) * S * 2;
  u64 best = 0;
  for (int i = 0; i < 2 * S; ++i) best = l[i] > best ? l[i] : best;
  out_idx.put((size_t)b * Nq + knn_perm(perm_q, qp), knn_key_index(best));
}

// K2: one wave per query; the T = Ncp/16 tile maxima of the query are contiguous and live in
// registers (16 per lane cover Nc <= 16384; the rare remainder is re-read each round).
// Bisection on order-preserving uint keys for tau with count(tilemax >= tau) >= k, stopping
// as soon as the count is within a small slack of k.
#define KM_TR 16
__global__ __launch_bounds__(256) void pn_knn_tau_kernel(const float* __restrict__ tilemax, int Nq,
                                                         int Nqp, int T, int k,
                                                         float* __restrict__ tau) {
  const int b = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int q = blockIdx.x * 4 + wave;
  if (q >= Nq) return;
  const float* __restrict__ row = tilemax + ((size_t)b * Nqp + q) * T;
  float v[KM_TR];
  float vmin = __builtin_inff(), vmax = -__builtin_inff();
#pragma unroll
  for (int e = 0; e < KM_TR; ++e) {
    const int t = e * 64 + lane;
    v[e] = t < T ? row[t] : -__builtin_inff();
    if (v[e] > -__builtin_inff()) vmin = fminf(vmin, v[e]);
    vmax = fmaxf(vmax, v[e]);
  }
  for (int t = KM_TR * 64 + lane; t < T; t += 64) {
    const float x = row[t];
    if (x > -__builtin_inff()) vmin = fminf(vmin, x);
    vmax = fmaxf(vmax, x);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    vmin = fminf(vmin, __shfl_xor(vmin, o, 64));
    vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
  }
  uint32_t lo = pn_f2ord(vmin);       // count(>= lo) = #finite tiles >= k (host checked)
  uint32_t hi = pn_f2ord(vmax) + 1u;  // count(>= hi) = 0 < k
  const int slack = k / 16 + 1;
  while (hi - lo > 1u) {
    const uint32_t mid = lo + ((hi - lo) >> 1);
    const float fm = pn_ord2f(mid);
    int c = 0;
#pragma unroll
    for (int e = 0; e < KM_TR; ++e) c += __builtin_popcountll(__ballot(v[e] >= fm));
    for (int t0 = KM_TR * 64; t0 < T; t0 += 64) {
      const int t = t0 + lane;
      c += __builtin_popcountll(__ballot(t < T && row[t] >= fm));
    }
    if (c >= k) {
      lo = mid;
      if (c <= k + slack) break;
    } else {
      hi = mid;
    }
  }
  if (lane == 0) tau[(size_t)b * Nqp + q] = pn_ord2f(lo);
}

// K4: one wave per query: gather its 2*S sub-lists into LDS, select/sort, emit.
//   out_idx != null: the k best original candidate indices (k <= 128), row perm_q(q).
//   out_val != null: the value of the k-th best (k <= KNN_CAP), row perm_q(q).
__global__ __launch_bounds__(256) void pn_knn_final_kernel(
    const u64* __restrict__ lists, const int* __restrict__ counts, int Nq, int Nqp, int k, int S,
    int subcap, KnnPerm perm_q, KnnPerm perm_c, KnnIdxOut out_idx,
    float* __restrict__ out_val, int* __restrict__ flags, int* __restrict__ anyflag) {
  __shared__ __attribute__((aligned(16))) uint32_t s_hist[4][256];
  __shared__ u64 s_keys[4][KNN_CAP];
  const int b = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int qp = blockIdx.x * 4 + wave;
  if (qp >= Nq) return;
  const size_t ql = (size_t)b * Nqp + qp;
  const int qo = knn_perm(perm_q, qp);
  u64* keys = s_keys[wave];
  // gather: lane s copies sub-list s — all 2 S sub-lists at once (a loop over the lists with one
  // dependent load each cost ~1 us per list and query)
  const int nsub = 2 * S;  // <= 64
  const int myc = lane < nsub ? counts[ql * nsub + lane] : 0;
  int inc = myc;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  const int n = __builtin_amdgcn_readlane(inc, 63);
  const bool bad = __ballot(myc > subcap) != 0 || n > KNN_CAP;
  if (!bad) {
    int cmax = myc;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cmax = max(cmax, __shfl_xor(cmax, o, 64));
    const int off = inc - myc;
    if (cmax <= nsub) {   // short lists (kNN graphs): a lane per list
      const u64* lp = lists + (ql * nsub + lane) * (size_t)subcap;
      // (eight loads in flight, then their stores: the one-by-one form paid a memory latency per entry)
      for (int e0 = 0; e0 < cmax; e0 += 8) {
        u64 kk[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) kk[u] = e0 + u < myc ? lp[e0 + u] : 0ull;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (e0 + u < myc) {
            // candidate index: permuted -> original, so that ties order by the caller's indices
            const int jp = (int)(0xffffffffu - (uint32_t)(kk[u] & 0xffffffffu));
            keys[off + e0 + u] = (kk[u] & 0xffffffff00000000ull) |
                                 (u64)(0xffffffffu - (uint32_t)knn_perm(perm_c, jp));
          }
        }
      }
    } else {              // long lists (K-th value of hundreds): the lanes over the entries of a list
      for (int s = 0; s < nsub; ++s) {
        const int c = __builtin_amdgcn_readlane(myc, s), o = __builtin_amdgcn_readlane(off, s);
        const u64* lp = lists + (ql * nsub + s) * (size_t)subcap;
        for (int e = lane; e < c; e += 64) {
          const u64 key = lp[e];
          const int jp = (int)(0xffffffffu - (uint32_t)(key & 0xffffffffu));
          keys[o + e] = (key & 0xffffffff00000000ull) |
                        (u64)(0xffffffffu - (uint32_t)knn_perm(perm_c, jp));
        }
      }
    }
  }
  if (bad || n < k) {
    // overflow (or NaNs): the caller recomputes flagged queries
    if (lane == 0) flags[(size_t)b * Nq + qo] = 1;
    if (lane == 0 && anyflag) anyflag[b] = 1;   // lets the fallback of an item without flags exit at once
    if (out_val && lane == 0) out_val[(size_t)b * Nq + qo] = __builtin_nanf("");
    return;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
  if (out_val) {
    u64 kth;
    if (n > k) {
      kth = knn_wave_select(keys, n, k, s_hist[wave]);
    } else {  // n == k: the k-th best is the minimum
      u64 m = ~0ull;
      for (int e = lane; e < n; e += 64) m = keys[e] < m ? keys[e] : m;
      kth = pn_wave_min_u64(m);
    }
    if (lane == 0) out_val[(size_t)b * Nq + qo] = pn_ord2f((uint32_t)(kth >> 32));
    return;
  }
  if (n > 128) knn_wave_select(keys, n, k, s_hist[wave]);
  const int m = n > 128 ? k : n;
  u64 k0 = lane < m ? keys[lane] : 0ull;
  u64 k1 = lane + 64 < m ? keys[lane + 64] : 0ull;
  knn_wave_sort128(k0, k1);
  const size_t o = ((size_t)b * Nq + qo) * k;
  if (lane < k) out_idx.put(o + lane, knn_key_index(k0));
  if (lane + 64 < k) out_idx.put(o + lane + 64, knn_key_index(k1));
}

// ---------------------------------------------------------------------------------------
#define KX_FUSED_K 10       // one-pass form of the bf16 x 3 graph (knn_x3.h, KIND 2): neighbours at most
#define KX_SEED_TILES 16    // ... tiles whose group maxima seed its running threshold before anything is collected
// PN_KNN_FUSED (read at every call): 1 = kNN graphs with k <= 10 at 64 / 128 / 256 channels and >= 2 048 points take the
// one-pass bf16 x 3 form instead of the exact one-pass kernel of knn_smallk.h; 0 (default until measured): they do not.
// PN_KNN_FUSED_NP = 6: six piece products instead of three.
static bool knn_fused_on() {
  const char* e = getenv("PN_KNN_FUSED");
  return e && atoi(e) != 0;
}

struct KnnPlan {
  bool fast, fused;
  int ksteps, qsets, Cp, Nqp, Ncp, S, tiles_per_slice, subcap, B, k;
};

static KnnPlan knn_mfma_plan(int mode, int B, int C, int Nq, int Nc, int k, bool want_value) {
  KnnPlan p;
  memset(&p, 0, sizeof(p));
  p.Nqp = (int)pn_align_up(Nq, 64);
  p.Ncp = (int)pn_align_up(Nc, 64);
  p.B = B;
  p.k = k;
  if (mode == 1) {
    p.ksteps = 4;
  } else if (C <= 4) {
    p.ksteps = 2;
  } else if (C <= 8) {
    p.ksteps = 4;
  } else if (C <= 64) {
    p.ksteps = 32;
  } else if (C <= 128) {
    p.ksteps = 64;
  } else if (C <= 256) {
    p.ksteps = 128;
  } else {
    p.ksteps = 0;
  }
  p.qsets = p.ksteps <= 32 ? 2 : 1;
  p.Cp = 2 * p.ksteps;
  // the threshold needs at least 2k tile maxima to be tight; other shapes use the scan path
  const int kmax = want_value ? KNN_CAP / 2 : KNN_MAXK;
  p.fast = p.ksteps > 0 && (Nc / 16) >= 2 * k && k <= kmax;
  if (p.fast) {
    const long long waves_q = (long long)B * pn_cdiv(p.Nqp, 32 * p.qsets);
    int S = (int)(8192 / (waves_q > 0 ? waves_q : 1));
    const int ntiles = p.Ncp / 32;
    if (S >= 8 && ntiles >= 64) S = 8;   // one slice per XCD (see pn_knn_mfma_kernel)
    if (S > 8) S = 8;
    if (S > ntiles) S = ntiles;
    if (S < 1) S = 1;
    p.tiles_per_slice = pn_cdiv(ntiles, S);
    if (S != 8) S = pn_cdiv(ntiles, p.tiles_per_slice);
    p.S = S;
    // expected survivors per sub-list ~ 1.1-1.3 k / (2 S); leave generous head-room
    p.subcap = (int)pn_align_up(3 * k / (2 * p.S) + 16, 8);
    if (2 * p.S * p.subcap > KNN_CAP) p.subcap = KNN_CAP / (2 * p.S);
    p.fused = mode == 0 && !want_value && k <= KX_FUSED_K && p.ksteps >= 32 && p.Ncp >= 2048 &&
              ntiles >= 4 * KX_SEED_TILES && Nq == Nc && knn_fused_on();
  }
  return p;
}

struct KnnWs {
  size_t xq, xxq, xc, xxc, tilemax, tau, cnt, flags, lists, v1, img, xxmax, xpm, xxo, total;
  size_t mu, xpc, xxcc, xxoc, xxmaxc;     // centred passes of the kNN graph (knn_x3.h)
};

#include "knn_x3.h"

// Passes on the bf16 matrix cores (knn_x3.h).  PN_KNN_X3 (read at every call) =
//   0: none;
//   1: the threshold pass of the feature metric (64 or 128 padded channels, enough candidates for
//      the split to pay for its image pass); the deciding pass stays exact fp32;
//   2 (default): also the collecting pass of the kNN graph of one set; the final sort works on
//      approximate keys and re-evaluates exactly what they cannot decide (pn_knn_final_x3_kernel).
//      The error of an approximate value scales with |q||c|, the gaps between near neighbours do
//      not: in the 64-channel layers of a network in training hundreds of candidates can fall
//      inside the 2-eps window of the k-th one.  Such queries are NOT flagged (the first version
//      did, and spent 4.7 ms per cfg4 step in the scan kernel: 9.6 -> 15.3 ms per step): all
//      candidates of the window get their exact value and the exact selection decides; the lists
//      have the full capacity of 1024 keys per query.  cfg4: 9.55 -> 9.27 ms per step;
//   3: also the threshold pass of the dot-product selections (a cliff of the same kind on a
//      converged embedding: thousands of dot products within 1e-6 of the K-th one overflow the
//      lists, measured 42 -> 55 ms per cfg5 step; off).
static int knn_x3_level() {
  const char* e = getenv("PN_KNN_X3");
  return e ? atoi(e) : 2;
}
// Piece products of the THRESHOLD pass (knn_x3.h): 3 for the dot-product form (the bandwidth's K-th neighbour on
// unit rows: 0.51 -> 0.31 ms per launch of 4 shapes, cfg5 172.5 -> 174.7 shapes/s), 6 for the squared-distance form of
// the kNN graphs.  Measured with 3 there too (tools/jobs/r6e.sh, profiles/r06_knn_p1_ab.txt): 0.32 -> 0.18 ms per
// launch on the pre-trained network of cfg5, but on cfg4's network — early in training, flat regions of a shape
// carry near-identical features — the four times wider window (A1 / A = 3.9 at 64 channels) sends so many rows over
// their list capacity that the gated fallback scan costs 1.7 ms per layer: 513 -> 423 shapes/s.
// PN_KNN_X3_P1 = 3 / 6 forces one form for both (developer A/B).
// With CENTRED passes (below) the graphs' window shrinks by the ratio of the centred to the original norms and the
// three-product threshold pass is the default for them as well.
static int knn_x3_p1_products(int mode, bool centred) {
  const char* e = getenv("PN_KNN_X3_P1");
  if (e && (atoi(e) == 6 || atoi(e) == 3)) return atoi(e);
  return (mode == 2 || centred) ? 3 : 6;
}
// PN_KNN_X3_CENTRE=0: the approximate passes of a kNN graph on the rows as they are (round 5); default: on the rows
// minus their per-channel mean (knn_x3.h)
static bool knn_x3_centre() {
  const char* e = getenv("PN_KNN_X3_CENTRE");
  return !(e && atoi(e) == 0);
}
// 256 channels (ksteps 128; round 4): the squared-distance form only — kNN graphs of the widest edge-conv
// layers (closed SplineNet) —, and only when the 128-query workgroups (one per CU: the resident queries take
// 192 registers) fill the chip: measured (tools/kbench.py knnwide, N = 2 500, k = 10) 12 segments 0.85 ms
// against 0.96 ms on the fp32 engine, 6 segments (120 workgroups) 0.73 against 0.55 ms.
static bool knn_x3_pass1(const KnnPlan& p, int mode, bool dot_form = false) {
  const int on = knn_x3_level();
  const bool wide256 = p.ksteps == 128 && mode == 0 && !dot_form && (long long)p.B * pn_cdiv(p.Nqp, 128) >= 192;
  // 128 channels with a small k (the SplineNets' graphs, k = 10): the final sort on approximate keys costs
  // more than the passes save (12 segments of 2 500 points: 0.61 against 0.52 ms on the fp32 engine)
  const bool wide128 = p.ksteps == 64 && (mode != 0 || dot_form || p.k >= 32);
  if (p.fused && mode == 0 && !dot_form && on >= 2) return true;      // the one-pass form (any of the three widths)
  return on && p.fast && (mode == 0 || (mode == 2 && on >= 3)) && (p.ksteps == 32 || wide128 || wide256) &&
         p.Ncp >= 2048;
}
#define KX_MAX_SLICES 16   // of the collecting pass (2 x 16 sub-lists per query)

static KnnWs knn_mfma_ws(const KnnPlan& p, int B, int C, int Nq, int k, bool self, bool v1) {
  KnnWs w;
  size_t o = 0;
  auto take = [&](size_t bytes) {
    size_t at = o;
    o += pn_align_up(bytes, 256);
    return at;
  };
  w.xc = take((size_t)B * p.Cp * p.Ncp * 4);
  w.xxc = take((size_t)B * p.Ncp * 4);
  if (self) {
    w.xq = w.xc;
    w.xxq = w.xxc;
  } else {
    w.xq = take((size_t)B * p.Cp * p.Nqp * 4);
    w.xxq = take((size_t)B * p.Nqp * 4);
  }
  w.tilemax = take((size_t)B * p.Nqp * (p.Ncp / 16) * 4);
  w.tau = take((size_t)B * p.Nqp * 4);
  const bool x3ws = knn_x3_pass1(p, 0);
  w.cnt = take((size_t)B * p.Nqp * 2 * (x3ws && p.S < KX_MAX_SLICES ? KX_MAX_SLICES : p.S) * 4);
  w.flags = take(((size_t)B * Nq + B) * 4);   // + one summary word per item
  // (the collecting pass on approximate values gathers a wider window: full capacity per query)
  w.lists = take((size_t)B * p.Nqp * (x3ws && self ? KNN_CAP : 2 * p.S * p.subcap) * 8);
  w.v1 = take(v1 ? pn_knn_v1_workspace(B, C, Nq, k, true) : 0);
  // candidate images + per-item largest squared norm of the bf16 x 3 pass 1 (both metrics qualify)
  const bool x3 = x3ws;
  w.img = take(x3 ? (size_t)B * p.Ncp * p.Cp * 6 : 0);
  w.xxmax = take(x3 ? (size_t)B * 8 : 0);      // B maxima of the original norms, then B of the centred ones: one memset
  // point-major fp32 rows + norms in original order: the exact repairs of the approximate final sort
  w.xpm = take(x3 && self ? (size_t)B * p.Ncp * p.Cp * 4 : 0);
  w.xxo = take(x3 && self ? (size_t)B * p.Ncp * 4 : 0);
  w.mu = take(x3 && self ? (size_t)B * p.Cp * 4 : 0);
  w.xpc = take(x3 && self ? (size_t)B * p.Ncp * p.Cp * 4 : 0);
  w.xxcc = take(x3 && self ? (size_t)B * p.Ncp * 4 : 0);
  w.xxoc = take(x3 && self ? (size_t)B * p.Ncp * 4 : 0);
  w.xxmaxc = w.xxmax + (size_t)B * 4;
  w.total = o;
  return w;
}

template <int KSTEPS, int QSETS, int MODE>
static void knn_mfma_launch_pass(int kind, dim3 grid, hipStream_t stream, const float* xq,
                                 const float* xxq, int Nq, int Nqp, const float* xc,
                                 const float* xxc, int Nc, int Ncp, int tps, float* tilemax,
                                 const float* tau, u64* lists, int* cnt, int subcap, KnnPerm perm_c) {
  if (kind == 0)
    hipLaunchKernelGGL((pn_knn_mfma_kernel<KSTEPS, QSETS, MODE, 0>), grid, dim3(256), 0, stream, xq,
                       xxq, Nq, Nqp, xc, xxc, Nc, Ncp, tps, tilemax, tau, lists, cnt, subcap, perm_c);
  else if (kind == 1)
    hipLaunchKernelGGL((pn_knn_mfma_kernel<KSTEPS, QSETS, MODE, 1>), grid, dim3(256), 0, stream, xq,
                       xxq, Nq, Nqp, xc, xxc, Nc, Ncp, tps, tilemax, tau, lists, cnt, subcap, perm_c);
  else if (MODE == 2)  // arg-max exists for the dot-product queries only
    hipLaunchKernelGGL((pn_knn_mfma_kernel<KSTEPS, QSETS, MODE, (MODE == 2 ? 2 : 0)>), grid,
                       dim3(256), 0, stream, xq, xxq, Nq, Nqp, xc, xxc, Nc, Ncp, tps, tilemax, tau,
                       lists, cnt, subcap, perm_c);
}

// The shared engine.  self: queries use the candidates' permuted copy (kNN graph of one set).
// approx_value: the K-th VALUE in the arithmetic of the bf16 x 3 passes (both passes on the matrix
// cores, no margins: thresholds and collected values come from the same arithmetic) — for
// statistics that need fp32-grade, not chain-exact, dot products (pn_dot_kth_x3_f32).
static int select_run(const KnnPlan& p, const KnnWs& w, int mode, bool self, const float* q,
                      int q_pm, int Nq, const float* c, int c_pm, int Nc, int B, int C, int k,
                      KnnIdxOut out_idx, float* out_val, int* flags_out, char* base,
                      hipStream_t stream, bool approx_value = false) {
  float* xc = (float*)(base + w.xc);
  float* xxc = (float*)(base + w.xxc);
  float* xq = (float*)(base + w.xq);
  float* xxq = (float*)(base + w.xxq);
  float* tilemax = (float*)(base + w.tilemax);
  float* tau = (float*)(base + w.tau);
  int* cnt = (int*)(base + w.cnt);
  int* flags = flags_out ? flags_out : (int*)(base + w.flags);
  int* anyflag = flags_out ? nullptr : flags + (size_t)B * Nq;   // one summary word per item behind the own flags
  u64* lists = (u64*)(base + w.lists);
  // k = 1 without the value (nearest-centre membership): one pass, no threshold, no lists, and
  // no candidate permutation (it only serves the tile-maxima threshold)
  const bool argmax = mode == 2 && k == 1 && out_val == nullptr && out_idx.p != nullptr && !self;
  const KnnPerm perm_c = knn_make_perm(Nc, argmax);
  const KnnPerm perm_q = self ? perm_c : knn_make_perm(Nq, true);

  PN_CHECK_HIP(hipMemsetAsync(flags, 0, ((size_t)B * Nq + (flags_out ? 0 : B)) * 4, stream));
  {
    PN_PROF("knn_prep", stream);
    knn_prep_launch(stream, c, B, C, Nc, p.Cp, p.Ncp, mode, c_pm, perm_c, xc, xxc);
    if (!self) knn_prep_launch(stream, q, B, C, Nq, p.Cp, p.Nqp, mode, q_pm, perm_q, xq, xxq);
  }
  PN_CHECK_LAUNCH();
  const bool x3p1 = !argmax && (approx_value ? knn_x3_pass1(p, 0, true) : knn_x3_pass1(p, mode));
  if (approx_value && !(x3p1 && mode == 2 && out_val && !out_idx.p)) {
    pn_set_error("select_run: the bf16 x 3 value selection needs C <= 128 and Nc >= 2048");
    return PN_ERR_UNSUPPORTED;
  }
  // collecting pass + approximate final: kNN graph of one set, indices only
  const bool x3p2 = x3p1 && knn_x3_level() >= 2 && self && mode == 0 && out_idx.p && !out_val;
  u32x4* img = (u32x4*)(base + w.img);
  unsigned* xxmax = (unsigned*)(base + w.xxmax);
  float* xpm = x3p2 ? (float*)(base + w.xpm) : nullptr;
  float* xxo = x3p2 ? (float*)(base + w.xxo) : nullptr;
  const float x3A = 4.0f * (float)(p.Cp + 2) * 0x1p-24f;
  // centred passes: the kNN graph of one set with both passes approximate (the exact engine never sees the centred rows)
  const bool centre = x3p2 && knn_x3_centre();
  // (needed: 2 Ao |q||c| >= 2 C 2^-24 |q||c| for the oracle's dot-product chain + 2^-23 * 2 |q||c| for its two closing
  //  roundings, i.e. Ao >= (C + 2) 2^-24; the norms' share of those roundings sits in knx_eps's 2^-21 term)
  const float x3Ao = centre ? (float)(p.Cp + 4) * 0x1p-24f : 0.f;
  const int np1 = knn_x3_p1_products(mode, centre);
  const float x3A1 = np1 == 3 ? x3A + 3.1f * 0x1p-16f : x3A;
  // The collecting pass keeps six products.  On three (PN_KNN_X3_P2=3, centred graphs only; bit-exact like the rest)
  // it is 0.39 -> 0.31 ms per launch and cfg5 gains 0.4 %, but during cfg4's first training steps its wider window
  // sends enough rows over their sub-list capacity for the gated fallback scan to cost 1.3 ms per step: 515 -> 440
  // shapes/s (tools/jobs/r6h.sh, profiles/r06_knn_centre_ab.txt) — a cliff not worth 0.4 %.
  const int np2 = (centre && np1 == 3 && getenv("PN_KNN_X3_P2") && atoi(getenv("PN_KNN_X3_P2")) == 3) ? 3 : 6;
  const float x3A2 = np2 == 3 ? x3A1 : x3A;      // error constant of the collected keys
  float* mu = centre ? (float*)(base + w.mu) : nullptr;
  float* xpc = centre ? (float*)(base + w.xpc) : nullptr;
  float* xxcc = centre ? (float*)(base + w.xxcc) : nullptr;
  float* xxoc = centre ? (float*)(base + w.xxoc) : nullptr;
  unsigned* xxmaxc = centre ? (unsigned*)(base + w.xxmaxc) : nullptr;
  if (x3p1) {
    PN_PROF("knn_x3_image", stream);
    PN_CHECK_HIP(hipMemsetAsync(xxmax, 0, (size_t)B * 8, stream));
    if (centre) {
      hipLaunchKernelGGL(pn_knn_x3_mean_kernel, dim3(p.Cp, B), dim3(256), 0, stream, (const float*)xc, p.Cp, p.Ncp, Nc, mu);
    }
    dim3 ig(p.Ncp / 32, B);
    if (p.ksteps == 32)
      hipLaunchKernelGGL(pn_knn_x3_image_kernel<8>, ig, dim3(256), 0, stream, xc, xxc, p.Ncp, img, xxmax, Nc, perm_c,
                         xpm, xxo, (const float*)mu, xpc, xxcc, xxoc, xxmaxc);
    else if (p.ksteps == 64)
      hipLaunchKernelGGL(pn_knn_x3_image_kernel<16>, ig, dim3(256), 0, stream, xc, xxc, p.Ncp, img, xxmax, Nc, perm_c,
                         xpm, xxo, (const float*)mu, xpc, xxcc, xxoc, xxmaxc);
    else
      hipLaunchKernelGGL(pn_knn_x3_image_kernel<32>, ig, dim3(256), 0, stream, xc, xxc, p.Ncp, img, xxmax, Nc, perm_c,
                         xpm, xxo, (const float*)mu, xpc, xxcc, xxoc, xxmaxc);
    PN_CHECK_LAUNCH();
  }
  // what the approximate passes read: the centred copy, its norms and their maximum — or the rows as they are
  const float* xq_a = centre ? (const float*)xpc : (const float*)xq;
  const float* xxq_a = centre ? (const float*)xxcc : (const float*)xxq;
  const float* xxc_a = centre ? (const float*)xxcc : (const float*)xxc;
  const unsigned* xxmax_a = centre ? (const unsigned*)xxmaxc : (const unsigned*)xxmax;
  if (p.fused && x3p2) {
    // ---- one pass: running threshold + collection, then the final sort with exact repairs (knn_x3.h, KIND 2) ----
    const char* ne = getenv("PN_KNN_FUSED_NP");
    const int npf = (ne && atoi(ne) == 6) ? 6 : 3;
    const float Af = npf == 3 ? x3A + 3.1f * 0x1p-16f : x3A;
    const int qpw = 128 * p.qsets, ntiles = p.Ncp / 32, subcapf = KNN_CAP / 2;
    dim3 gf(1, pn_cdiv(p.Nqp, qpw), B);
    {
      PN_PROF(p.ksteps == 32 ? "knn_x3_fused_c64" : "knn_x3_fused_wide", stream);
#define KX_GOF(NCH, QS, TPS_, NP_)                                                                                       \
  hipLaunchKernelGGL((pn_knn_x3_pass_kernel<NCH, QS, 0, TPS_, 2, NP_>), gf, dim3(256), 0, stream, xq_a, xxq_a, Nq, p.Nqp, \
                     img, xxc_a, Nc, p.Ncp, ntiles, tilemax, (const float*)tau, lists, cnt, subcapf, xxmax_a,           \
                     (const float*)xxq, (const unsigned*)xxmax, Af, x3Ao)
      if (p.ksteps == 32) {
        if (npf == 3) KX_GOF(8, 2, 2, 3); else KX_GOF(8, 2, 2, 6);
      } else if (p.ksteps == 64) {
        if (npf == 3) KX_GOF(16, 1, 1, 3); else KX_GOF(16, 1, 1, 6);
      } else {
        if (npf == 3) KX_GOF(32, 1, 1, 3); else KX_GOF(32, 1, 1, 6);
      }
#undef KX_GOF
    }
    PN_CHECK_LAUNCH();
    {
      PN_PROF("knn_final", stream);
      if (p.ksteps == 32)
        hipLaunchKernelGGL(pn_knn_final_x3_kernel<64>, dim3(pn_cdiv(Nq, 4), B), dim3(256), 0, stream, lists, cnt, Nq,
                           p.Nqp, k, 1, subcapf, perm_q, perm_c, (const float*)xpm, (const float*)xxo, xxmax, Nc,
                           Af, out_idx, flags, anyflag, (const float*)xxoc, (const unsigned*)xxmaxc, x3Ao);
      else if (p.ksteps == 64)
        hipLaunchKernelGGL(pn_knn_final_x3_kernel<128>, dim3(pn_cdiv(Nq, 4), B), dim3(256), 0, stream, lists, cnt, Nq,
                           p.Nqp, k, 1, subcapf, perm_q, perm_c, (const float*)xpm, (const float*)xxo, xxmax, Nc,
                           Af, out_idx, flags, anyflag, (const float*)xxoc, (const unsigned*)xxmaxc, x3Ao);
      else
        hipLaunchKernelGGL(pn_knn_final_x3_kernel<256>, dim3(pn_cdiv(Nq, 4), B), dim3(256), 0, stream, lists, cnt, Nq,
                           p.Nqp, k, 1, subcapf, perm_q, perm_c, (const float*)xpm, (const float*)xxo, xxmax, Nc,
                           Af, out_idx, flags, anyflag, (const float*)xxoc, (const unsigned*)xxmaxc, x3Ao);
    }
    PN_CHECK_LAUNCH();
    return PN_OK;
  }
  dim3 grid(p.S, pn_cdiv(p.Nqp, 32 * p.qsets * 4), B);
  for (int pass = 0; pass < (argmax ? 1 : 2); ++pass) {
    const int collect = argmax ? 2 : pass;
    if (pass == 0 && x3p1) {
      // workgroups of 4 waves x 32 qsets queries, two per CU; steps of tstep tiles (knn_x3.h).
      // Slices: whole rounds of the 512 slots where possible, a fixed cost of about one tile each.
      const int qpw = 128 * p.qsets, ntiles = p.Ncp / 32;
      const long long rowblocks = (long long)B * pn_cdiv(p.Nqp, qpw);
      int tps1 = ntiles;
      double best_score = -1.0;
      for (int t = 8; t <= ntiles; t += 2) {
        const double rounds = (double)(rowblocks * pn_cdiv(ntiles, t)) / (256.0 * KX_WPE(p.ksteps / 4, mode));
        const double score = rounds / (double)(long long)(rounds + 0.999999) * (double)t / ((double)t + 1.0);
        if (score > best_score) {
          best_score = score;
          tps1 = t;
        }
      }
      dim3 g1(pn_cdiv(ntiles, tps1), pn_cdiv(p.Nqp, qpw), B);
      {
        PN_PROF(mode == 2 ? "sel_x3_pass1_dot" : (p.ksteps == 32 ? "knn_x3_pass1_c64" : "knn_x3_pass1_wide"), stream);
#define KX_GO_NP(NCH, QS, MD, TPS_, KIND_, NP_, GRID, TPSL, SUBCAP)                                                  \
  hipLaunchKernelGGL((pn_knn_x3_pass_kernel<NCH, QS, MD, TPS_, KIND_, NP_>), GRID, dim3(256), 0, stream, xq_a, xxq_a,  \
                     Nq, p.Nqp, img, xxc_a, Nc, p.Ncp, TPSL, tilemax, (const float*)tau, lists, cnt, SUBCAP, xxmax_a)
#define KX_GO(NCH, QS, MD, TPS_, KIND_, GRID, TPSL, SUBCAP) KX_GO_NP(NCH, QS, MD, TPS_, KIND_, 6, GRID, TPSL, SUBCAP)
#define KX_GO1(NCH, QS, MD, TPS_)                            \
  {                                                          \
    if (np1 == 3)                                            \
      KX_GO_NP(NCH, QS, MD, TPS_, 0, 3, g1, tps1, 0);        \
    else                                                     \
      KX_GO_NP(NCH, QS, MD, TPS_, 0, 6, g1, tps1, 0);        \
  }
        if (mode == 0 && p.ksteps == 32)
          KX_GO1(8, 2, 0, 2)
        else if (mode == 0 && p.ksteps == 64)
          KX_GO1(16, 1, 0, 1)
        else if (mode == 0)
          KX_GO1(32, 1, 0, 1)
        else if (p.ksteps == 32)
          KX_GO1(8, 2, 2, 2)
        else
          KX_GO1(16, 1, 2, 1)
      }
      PN_CHECK_LAUNCH();
      {
        PN_PROF("knn_tau", stream);
        hipLaunchKernelGGL(pn_knn_tau_kernel, dim3(pn_cdiv(Nq, 4), B), dim3(256), 0, stream,
                           tilemax, Nq, p.Nqp, p.Ncp / 16, k, tau);
        // (the value selection collects in the arithmetic of a six-product threshold pass: no margin then; with the
        // three-product pass its threshold is lowered by both passes' bounds like the graph's)
        if (!approx_value || np1 == 3)
          hipLaunchKernelGGL(pn_knn_x3_margin_kernel, dim3(pn_cdiv(Nq, 256), B), dim3(256), 0, stream, tau, xxq_a, Nq,
                             p.Nqp, xxmax_a, x3p2 ? x3A2 : x3A, mode, (x3p2 || approx_value) ? 2.0f : 1.0f, x3A1,
                             (const float*)xxq, (const unsigned*)xxmax, x3Ao);
      }
      PN_CHECK_LAUNCH();
      if (approx_value) {
        // collecting pass in the same arithmetic, then the standard final selection of the value
        int tpsv = tps1;
        if (pn_cdiv(ntiles, tpsv) > KX_MAX_SLICES) tpsv = (int)pn_align_up(pn_cdiv(ntiles, KX_MAX_SLICES), 2);
        const int Sv = pn_cdiv(ntiles, tpsv);
        int subcapv = (int)pn_align_up(3 * k / (2 * Sv) + 16, 8);
        if (2 * Sv * subcapv > 2 * p.S * p.subcap) subcapv = (2 * p.S * p.subcap) / (2 * Sv);
        if (2 * Sv * subcapv > KNN_CAP) subcapv = KNN_CAP / (2 * Sv);
        dim3 gv(Sv, pn_cdiv(p.Nqp, qpw), B);
        {
          PN_PROF("sel_x3_pass2_dot", stream);
          if (p.ksteps == 32)
            KX_GO(8, 2, 2, 2, 1, gv, tpsv, subcapv);
          else
            KX_GO(16, 1, 2, 1, 1, gv, tpsv, subcapv);
        }
        PN_CHECK_LAUNCH();
        {
          PN_PROF("knn_final", stream);
          hipLaunchKernelGGL(pn_knn_final_kernel, dim3(pn_cdiv(Nq, 4), B), dim3(256), 0, stream, lists, cnt, Nq,
                             p.Nqp, k, Sv, subcapv, perm_q, perm_c, out_idx, out_val, flags, (int*)nullptr);
        }
        PN_CHECK_LAUNCH();
        return PN_OK;
      }
      if (!x3p2) continue;
      // ---- collecting pass on the approximate values + final sort with exact repairs ----
      int tps2 = tps1;
      if (pn_cdiv(ntiles, tps2) > KX_MAX_SLICES) {
        tps2 = (int)pn_align_up(pn_cdiv(ntiles, KX_MAX_SLICES), 2);
      }
      const int S2 = pn_cdiv(ntiles, tps2);
      int subcap2;
      subcap2 = KNN_CAP / (2 * S2);   // the whole capacity of a query (knn_mfma_ws)
      dim3 g2(S2, pn_cdiv(p.Nqp, qpw), B);
      {
        PN_PROF(p.ksteps == 32 ? "knn_x3_pass2_c64" : "knn_x3_pass2_wide", stream);
#define KX_GO2(NCH, QS, TPS_)                                    \
  {                                                              \
    if (np2 == 3)                                                \
      KX_GO_NP(NCH, QS, 0, TPS_, 1, 3, g2, tps2, subcap2);       \
    else                                                         \
      KX_GO_NP(NCH, QS, 0, TPS_, 1, 6, g2, tps2, subcap2);       \
  }
        if (p.ksteps == 32)
          KX_GO2(8, 2, 2)
        else if (p.ksteps == 64)
          KX_GO2(16, 1, 1)
        else
          KX_GO2(32, 1, 1)
#undef KX_GO2
      }
      PN_CHECK_LAUNCH();
      {
        PN_PROF("knn_final", stream);
        if (p.ksteps == 32)
          hipLaunchKernelGGL(pn_knn_final_x3_kernel<64>, dim3(pn_cdiv(Nq, 4), B), dim3(256), 0, stream, lists, cnt, Nq,
                             p.Nqp, k, S2, subcap2, perm_q, perm_c, (const float*)xpm, (const float*)xxo, xxmax, Nc,
                             x3A2, out_idx, flags, anyflag, (const float*)xxoc, (const unsigned*)xxmaxc, x3Ao);
        else if (p.ksteps == 64)
          hipLaunchKernelGGL(pn_knn_final_x3_kernel<128>, dim3(pn_cdiv(Nq, 4), B), dim3(256), 0, stream, lists, cnt, Nq,
                             p.Nqp, k, S2, subcap2, perm_q, perm_c, (const float*)xpm, (const float*)xxo, xxmax, Nc,
                             x3A2, out_idx, flags, anyflag, (const float*)xxoc, (const unsigned*)xxmaxc, x3Ao);
        else
          hipLaunchKernelGGL(pn_knn_final_x3_kernel<256>, dim3(pn_cdiv(Nq, 4), B), dim3(256), 0, stream, lists, cnt, Nq,
                             p.Nqp, k, S2, subcap2, perm_q, perm_c, (const float*)xpm, (const float*)xxo, xxmax, Nc,
                             x3A2, out_idx, flags, anyflag, (const float*)xxoc, (const unsigned*)xxmaxc, x3Ao);
      }
      PN_CHECK_LAUNCH();
      return PN_OK;
#undef KX_GO
    }
    static const char* const pass_names[2][5] = {
        {"knn_mfma_pass1_c4", "knn_mfma_pass1_c64", "knn_mfma_pass1_wide", "knn_mfma_pass1_pn",
         "sel_mfma_pass1_dot"},
        {"knn_mfma_pass2_c4", "knn_mfma_pass2_c64", "knn_mfma_pass2_wide", "knn_mfma_pass2_pn",
         "sel_mfma_pass2_dot"}};
    const int fam =
        mode == 2 ? 4 : (mode == 1 ? 3 : (p.ksteps <= 4 ? 0 : (p.ksteps == 32 ? 1 : 2)));
    {
      PN_PROF(argmax ? "sel_mfma_argmax_dot" : pass_names[pass][fam], stream);
#define KM_GO(KS, QS, MD)                                                                        \
  knn_mfma_launch_pass<KS, QS, MD>(collect, grid, stream, xq, xxq, Nq, p.Nqp, xc, xxc, Nc, p.Ncp, \
                                   p.tiles_per_slice, tilemax, tau, lists, cnt, p.subcap, perm_c)
      if (mode == 1)
        KM_GO(4, 2, 1);
      else if (mode == 2 && p.ksteps == 2)
        KM_GO(2, 2, 2);
      else if (mode == 2 && p.ksteps == 4)
        KM_GO(4, 2, 2);
      else if (mode == 2 && p.ksteps == 32)
        KM_GO(32, 2, 2);
      else if (mode == 2 && p.ksteps == 64)
        KM_GO(64, 1, 2);
      else if (mode == 2)
        KM_GO(128, 1, 2);
      else if (p.ksteps == 2)
        KM_GO(2, 2, 0);
      else if (p.ksteps == 4)
        KM_GO(4, 2, 0);
      else if (p.ksteps == 32)
        KM_GO(32, 2, 0);
      else if (p.ksteps == 64)
        KM_GO(64, 1, 0);
      else
        KM_GO(128, 1, 0);
#undef KM_GO
    }
    PN_CHECK_LAUNCH();
    if (collect == 0) {
      PN_PROF("knn_tau", stream);
      hipLaunchKernelGGL(pn_knn_tau_kernel, dim3(pn_cdiv(Nq, 4), B), dim3(256), 0, stream,
                         tilemax, Nq, p.Nqp, p.Ncp / 16, k, tau);
      PN_CHECK_LAUNCH();
    }
  }
  if (argmax) {
    PN_PROF("knn_final", stream);
    hipLaunchKernelGGL(pn_knn_argmax_final_kernel, dim3(pn_cdiv(Nq, 256), B), dim3(256), 0, stream, lists,
                       Nq, p.Nqp, p.S, perm_q, out_idx);
    PN_CHECK_LAUNCH();
    return PN_OK;
  }
  {
    PN_PROF("knn_final", stream);
    hipLaunchKernelGGL(pn_knn_final_kernel, dim3(pn_cdiv(Nq, 4), B), dim3(256), 0, stream, lists,
                       cnt, Nq, p.Nqp, k, p.S, p.subcap, perm_q, perm_c, out_idx, out_val, flags, anyflag);
  }
  PN_CHECK_LAUNCH();
  return PN_OK;
}

// K-th largest dot product of every query row, both distance passes in bf16 x 3 arithmetic
// (error-free operand split, six piece products, fp32 accumulate: fp32-grade values, |error| ~1e-7
// on unit vectors, not the fma chain of pn_dot_select_f32): for statistics such as the mean-shift
// bandwidth (src/mean_shift.py:125-137: the K-th nearest-neighbour distance averaged over the
// points).  Same workspace and flags as pn_dot_select_f32(out_val); PN_ERR_UNSUPPORTED outside the
// path of the split passes (C <= 128, Nc >= 2048, the fast-path shape rules).
extern "C" int pn_dot_kth_x3_f32(const float* q, int Nq, const float* c, int Nc, int B, int C, int k,
                                 float* out_val, int* flags, void* workspace, size_t workspace_bytes,
                                 void* stream) {
  PN_CHECK_ARG(q && c && out_val && flags, "pn_dot_kth_x3_f32: null pointer");
  PN_CHECK_ARG(B > 0 && C > 0 && Nq > 0 && Nc > 0 && k >= 1 && k <= Nc,
               "pn_dot_kth_x3_f32: bad sizes (B=%d C=%d Nq=%d Nc=%d k=%d)", B, C, Nq, Nc, k);
  const KnnPlan p = knn_mfma_plan(2, B, C, Nq, Nc, k, true);
  if (!p.fast || !knn_x3_pass1(p, 0, true)) {
    pn_set_error("pn_dot_kth_x3_f32: shape outside the bf16 x 3 path (C <= 128, Nc >= 2048, Nc/16 >= 2k)");
    return PN_ERR_UNSUPPORTED;
  }
  const KnnWs w = knn_mfma_ws(p, B, C, Nq, k, false, false);
  PN_CHECK_ARG(workspace && workspace_bytes >= w.total, "pn_dot_kth_x3_f32: workspace too small");
  return select_run(p, w, 2, false, q, 1, Nq, c, 1, Nc, B, C, k, KnnIdxOut{nullptr, 0}, out_val, flags, (char*)workspace,
                    (hipStream_t)stream, true);
}

// ---- K-th largest dot product between unit vectors on the fp16 matrix cores ----------------
// The bandwidth statistic of mean-shift (src/mean_shift.py:125-137) needs the VALUE of the K-th
// nearest neighbour of every point, averaged over the points, to 1e-5 — not its identity.  The two
// passes of the engine above (tile maxima, collect) are matrix-core bound at the fp32 MFMA rate;
// here they evaluate the dot products with the scaled fp16 x 2 split of meanshift_h2.h (three
// 32-cycle MFMAs per 16 channels instead of eight 64-cycle ones; |error| <= ~1e-7 on dot products
// of unit vectors, far below the tolerance of the statistic and of the same size as the error of
// an fp32 GEMM).  Candidates come as the tile images of pn_meanshift_h2_split_f32, queries as fp32
// rows split in registers; tile maxima, thresholds, survivor lists and the final selection are
// those of the exact engine (natural candidate order: identity permutation).
// KIND 0: tile maxima; 1: collect survivors.  grid (slices, blocks of 256 queries, B), 512 threads.
template <int KIND>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void pn_dotsel_h2_kernel(
    const float* __restrict__ Q, int Nq, int Nqp, const u32x4* __restrict__ PC, int Nc, int Ncp,
    int tiles_per_slice, float* __restrict__ tilemax, const float* __restrict__ tau,
    u64* __restrict__ lists, int* __restrict__ counts, int subcap) {
  __shared__ __attribute__((aligned(16))) u32x4 ldsP[2][H2_IMG_U4];
  const int b = blockIdx.z;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int col = lane & 31, h = lane >> 5;
  const int q0 = (blockIdx.y * 8 + wave) * 32;
  const bool wave_on = q0 < Nq;
  const int ntiles = Ncp / 32;
  const int S = gridDim.x, slice = blockIdx.x;
  const int t_begin = slice * tiles_per_slice;
  const int t_end = min(ntiles, t_begin + tiles_per_slice);
  const int T16 = Ncp / 16;
  const u32x4* __restrict__ PCb = PC + (size_t)b * ntiles * H2_IMG_U4;

  // resident queries as B operands: k-step s = channels 16 s + 8 h + e
  const int q = q0 + col;
  const int qcl = q < Nq ? q : Nq - 1;
  f16x8 qh[8], qm[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const float* src = Q + ((size_t)b * Nq + qcl) * 128 + 16 * s + 8 * h;
    const float4 a = *reinterpret_cast<const float4*>(src);
    const float4 c = *reinterpret_cast<const float4*>(src + 4);
    u32x4 vh, vm;
    H2_SPLIT_TO(a.x * H2_SX, a.y * H2_SX, vh, vm, 0);
    H2_SPLIT_TO(a.z * H2_SX, a.w * H2_SX, vh, vm, 1);
    H2_SPLIT_TO(c.x * H2_SX, c.y * H2_SX, vh, vm, 2);
    H2_SPLIT_TO(c.z * H2_SX, c.w * H2_SX, vh, vm, 3);
    qh[s] = h2_as_f16(vh);
    qm[s] = h2_as_f16(vm);
  }
  const float tq = (KIND == 1 && q < Nq) ? tau[(size_t)b * Nqp + qcl] : __builtin_inff();
  int mycnt = 0;
  u64* sub = lists + ((((size_t)b * Nqp + qcl) * S + slice) * 2 + h) * (size_t)subcap;

#define DS_STAGE(MT, BUF)                                                                  \
  {                                                                                        \
    _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                        \
      const int c_ = wave * 2 + u;                                                         \
      X3_GLDS16(PCb + (size_t)(MT) * H2_IMG_U4 + c_ * 64 + lane, &ldsP[BUF][c_ * 64]);     \
    }                                                                                      \
  }
  int cur = 0;
  if (t_begin < t_end) DS_STAGE(t_begin, 0);
  const int rowoff = col * 16, sw = x3_swz(col);
  for (int mt = t_begin; mt < t_end; ++mt) {
    const int j0 = mt * 32;
    __syncthreads();  // image of tile mt landed; every wave is done with tile mt - 1
    if (mt + 1 < t_end) DS_STAGE(mt + 1, cur ^ 1);
    if (wave_on) {
      f32x16 sa;
#pragma unroll
      for (int r = 0; r < 16; ++r) sa[r] = 0.f;
      const u32x4* __restrict__ lp = ldsP[cur];
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const int slot = rowoff + ((2 * s + h) ^ sw);
        const f16x8 ah = h2_as_f16(lp[slot]);
        const f16x8 am = h2_as_f16(lp[H2_PIECE_U4 + slot]);
        H2_MFMA(sa, am, qh[s]);
        H2_MFMA(sa, ah, qm[s]);
        H2_MFMA(sa, ah, qh[s]);
      }
      // D[candidate = (r&3) + 8(r>>2) + 4h][query = col]
      const bool tail = j0 + 32 > Nc;
      if (KIND == 0) {
        float tm = -__builtin_inff();
        if (!tail) {
#pragma unroll
          for (int r = 0; r < 16; ++r) tm = fmaxf(tm, sa[r]);
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            tm = fmaxf(tm, j0 + (r & 3) + 8 * (r >> 2) + 4 * h < Nc ? sa[r] : -__builtin_inff());
        }
        if (q < Nqp) tilemax[((size_t)b * Nqp + q) * T16 + (2 * mt + h)] = tm * H2_ISX2;
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
          const float v = sa[r] * H2_ISX2;
          if (j0 + row < Nc && v >= tq) {
            if (mycnt < subcap) sub[mycnt] = knn_key(v, j0 + row);
            ++mycnt;
          }
        }
      }
    }
    cur ^= 1;
  }
#undef DS_STAGE
  if (KIND == 1 && wave_on && q < Nqp) counts[(((size_t)b * Nqp + q) * S + slice) * 2 + h] = mycnt;
}

extern "C" int pn_dot_kth_unit_h2_f32(const float* q, int Nq, const void* img_c, int Nc, int B, int D,
                                      int k, float* out_val, int* flags, void* workspace,
                                      size_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PN_CHECK_ARG(q && img_c && out_val && flags, "pn_dot_kth_unit_h2_f32: null pointer");
  PN_CHECK_ARG(D == 128, "pn_dot_kth_unit_h2_f32: embedding size %d unsupported (built for 128)", D);
  PN_CHECK_ARG(B > 0 && Nq > 0 && Nc > 0 && k >= 1 && k <= Nc,
               "pn_dot_kth_unit_h2_f32: bad sizes (B=%d Nq=%d Nc=%d k=%d)", B, Nq, Nc, k);
  const KnnPlan p = knn_mfma_plan(2, B, D, Nq, Nc, k, true);
  if (!p.fast) {
    pn_set_error("pn_dot_kth_unit_h2_f32: shape outside the fast path (Nc/16 >= 2k, k <= %d)", KNN_CAP / 2);
    return PN_ERR_UNSUPPORTED;
  }
  const KnnWs w = knn_mfma_ws(p, B, D, Nq, k, false, false);
  PN_CHECK_ARG(workspace && workspace_bytes >= w.total, "pn_dot_kth_unit_h2_f32: workspace too small");
  char* base = (char*)workspace;
  float* tilemax = (float*)(base + w.tilemax);
  float* tau = (float*)(base + w.tau);
  int* cnt = (int*)(base + w.cnt);
  u64* lists = (u64*)(base + w.lists);
  const KnnPerm ident_q = knn_make_perm(Nq, true), ident_c = knn_make_perm(Nc, true);
  PN_CHECK_HIP(hipMemsetAsync(flags, 0, (size_t)B * Nq * 4, stream));
  // rows Nq .. Nqp-1 of tilemax / counts are never read (the tau and final kernels stop at Nq).
  // Workgroups hold 256 queries (one per CU): fewer, longer slices than the exact engine's plan so
  // that the grid fits one round of the 256 CUs; the sub-lists grow accordingly, inside the same
  // allocation.
  const int ntiles = p.Ncp / 32;
  const long long rowblocks = (long long)B * pn_cdiv(Nq, 256);
  int S = p.S;
  if (rowblocks * S > 256 && rowblocks <= 256) S = (int)(256 / rowblocks);
  S = S < 1 ? 1 : S;
  const int tps = pn_cdiv(ntiles, S);
  S = pn_cdiv(ntiles, tps);
  int subcap = (int)pn_align_up(3 * k / (2 * S) + 16, 8);
  if (2 * S * subcap > 2 * p.S * p.subcap) subcap = (2 * p.S * p.subcap) / (2 * S);
  if (2 * S * subcap > KNN_CAP) subcap = KNN_CAP / (2 * S);
  dim3 grid(S, pn_cdiv(Nq, 256), B);
  {
    PN_PROF("sel_h2_pass1_dot", stream);
    hipLaunchKernelGGL(pn_dotsel_h2_kernel<0>, grid, dim3(512), 0, stream, q, Nq, p.Nqp, (const u32x4*)img_c,
                       Nc, p.Ncp, tps, tilemax, (const float*)tau, lists, cnt, subcap);
  }
  PN_CHECK_LAUNCH();
  {
    PN_PROF("knn_tau", stream);
    hipLaunchKernelGGL(pn_knn_tau_kernel, dim3(pn_cdiv(Nq, 4), B), dim3(256), 0, stream, tilemax, Nq, p.Nqp,
                       p.Ncp / 16, k, tau);
  }
  PN_CHECK_LAUNCH();
  {
    PN_PROF("sel_h2_pass2_dot", stream);
    hipLaunchKernelGGL(pn_dotsel_h2_kernel<1>, grid, dim3(512), 0, stream, q, Nq, p.Nqp, (const u32x4*)img_c,
                       Nc, p.Ncp, tps, tilemax, (const float*)tau, lists, cnt, subcap);
  }
  PN_CHECK_LAUNCH();
  {
    PN_PROF("knn_final", stream);
    hipLaunchKernelGGL(pn_knn_final_kernel, dim3(pn_cdiv(Nq, 4), B), dim3(256), 0, stream, lists, cnt, Nq,
                       p.Nqp, k, S, subcap, ident_q, ident_c, KnnIdxOut{nullptr, 0}, out_val, flags, (int*)nullptr);
  }
  PN_CHECK_LAUNCH();
  return PN_OK;
}

#include "knn_smallk.h"

// ---- kNN graph entry points ---------------------------------------------------------------
extern "C" size_t pn_knn_workspace(int B, int C, int N, int k) {
  size_t best = 0;
  {
    const KskPlan sk = ksk_plan(0, B, C, N, k);
    if (sk.ok) best = sk.total;
  }
  for (int mode = 0; mode < 2; ++mode) {
    if (mode == 1 && C != 6) continue;
    KnnPlan p = knn_mfma_plan(mode, B, C, N, N, k, false);
    size_t a = p.fast ? knn_mfma_ws(p, B, C, N, k, true, true).total
                      : pn_knn_v1_workspace(B, C, N, k, false);
    if (a > best) best = a;
  }
  return best;
}

static int knn_dispatch(int mode, const float* x, int B, int C, int N, int k, KnnIdxOut idx,
                        void* workspace, size_t workspace_bytes, hipStream_t stream) {
  PN_CHECK_ARG(x && idx.p, "pn_knn: null pointer");
  PN_CHECK_ARG(B > 0 && C > 0 && N > 0, "pn_knn: empty input (B=%d C=%d N=%d)", B, C, N);
  PN_CHECK_ARG(k >= 1 && k <= KNN_MAXK, "pn_knn: k=%d unsupported (1..%d)", k, KNN_MAXK);
  PN_CHECK_ARG(k <= N, "pn_knn: k=%d exceeds the number of points N=%d", k, N);
  PN_CHECK_ARG(mode == 0 || C == 6, "pn_knn_pn: points+normals metric needs C=6, got %d", C);
  PN_CHECK_ARG(workspace && workspace_bytes >= pn_knn_workspace(B, C, N, k),
               "pn_knn: workspace too small");
  const KnnPlan p = knn_mfma_plan(mode, B, C, N, N, k, false);
  if (!(p.fused && knn_x3_level() >= 2)) {
    // small k (the SplineNets' graphs): one distance pass, the k best of a lane in registers (knn_smallk.h)
    const KskPlan sk = ksk_plan(mode, B, C, N, k);
    if (sk.ok) return ksk_run(sk, x, B, C, N, k, idx.p, idx.is32, (char*)workspace, stream);
  }
  if (!p.fast)
    return pn_knn_v1_launch(mode, x, B, C, N, k, idx, workspace, workspace_bytes, stream, nullptr);
  const KnnWs w = knn_mfma_ws(p, B, C, N, k, true, true);
  char* base = (char*)workspace;
  int rc = select_run(p, w, mode, true, x, 0, N, x, 0, N, B, C, k, idx, nullptr, nullptr, base,
                      stream);
  if (rc) return rc;
  // degenerate queries (flagged) are redone by the generic scan kernel; waves without a
  // flagged query exit immediately
  PN_PROF("knn_fallback_gate", stream);
  return pn_knn_v1_launch(mode, x, B, C, N, k, idx, base + w.v1,
                          pn_knn_v1_workspace(B, C, N, k, true), stream, (int*)(base + w.flags),
                          (int*)(base + w.flags) + (size_t)B * N);
}

extern "C" int pn_knn_f32(const float* x, int B, int C, int N, int k, int64_t* idx,
                          void* workspace, size_t workspace_bytes, void* stream) {
  return knn_dispatch(0, x, B, C, N, k, KnnIdxOut{idx, 0}, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int pn_knn_pn_f32(const float* x6, int B, int N, int k, int64_t* idx, void* workspace,
                             size_t workspace_bytes, void* stream) {
  return knn_dispatch(1, x6, B, 6, N, k, KnnIdxOut{idx, 0}, workspace, workspace_bytes, (hipStream_t)stream);
}

// The graph as the library's own kernels take it (pn_edgeconv_reduce_fwd_i32, pn_edgeconv_bwd_i32): int32
// indices, same values as pn_knn_f32 / pn_knn_pn_f32 (metric 0: feature space, 1: points + normals, C = 6).
extern "C" int pn_knn_graph_i32(const float* x, int B, int C, int N, int k, int metric, int32_t* idx,
                                void* workspace, size_t workspace_bytes, void* stream) {
  PN_CHECK_ARG(metric == 0 || metric == 1, "pn_knn_graph_i32: metric %d", metric);
  return knn_dispatch(metric, x, B, C, N, k, KnnIdxOut{idx, 1}, workspace, workspace_bytes, (hipStream_t)stream);
}

// ---- dot-product selection between two point-major sets ------------------------------------
// q (B,Nq,C), c (B,Nc,C) point-major.  For every query either the indices of the k candidates
// with the largest dot product (out_idx (B,Nq,k), best first, ties -> smaller index) or the
// k-th largest dot product (out_val (B,Nq)).  flags (B,Nq) int32: 1 where the result of a query
// is NOT valid (survivor list overflow on massively tied data) — the caller must recompute
// those rows.  Returns PN_ERR_UNSUPPORTED when the shape is outside the fast path
// (needs Nc/16 >= 2k, C <= 256).
extern "C" size_t pn_dot_select_workspace(int B, int C, int Nq, int Nc, int k, int want_value) {
  KnnPlan p = knn_mfma_plan(2, B, C, Nq, Nc, k, want_value != 0);
  if (!p.fast) return 0;
  return knn_mfma_ws(p, B, C, Nq, k, false, false).total;
}

extern "C" int pn_dot_select_f32(const float* q, int Nq, const float* c, int Nc, int B, int C,
                                 int k, int64_t* out_idx, float* out_val, int* flags,
                                 void* workspace, size_t workspace_bytes, void* stream) {
  PN_CHECK_ARG(q && c && flags && (out_idx || out_val), "pn_dot_select_f32: null pointer");
  PN_CHECK_ARG(!(out_idx && out_val), "pn_dot_select_f32: ask for indices or the value, not both");
  PN_CHECK_ARG(B > 0 && C > 0 && Nq > 0 && Nc > 0 && k >= 1 && k <= Nc,
               "pn_dot_select_f32: bad sizes (B=%d C=%d Nq=%d Nc=%d k=%d)", B, C, Nq, Nc, k);
  const KnnPlan p = knn_mfma_plan(2, B, C, Nq, Nc, k, out_val != nullptr);
  if (!p.fast) {
    pn_set_error("pn_dot_select_f32: shape outside the fast path (Nc/16 >= 2k, C <= 256, k <= %d)",
                 out_val ? KNN_CAP / 2 : KNN_MAXK);
    return PN_ERR_UNSUPPORTED;
  }
  const KnnWs w = knn_mfma_ws(p, B, C, Nq, k, false, false);
  PN_CHECK_ARG(workspace && workspace_bytes >= w.total, "pn_dot_select_f32: workspace too small");
  return select_run(p, w, 2, false, q, 1, Nq, c, 1, Nc, B, C, k, KnnIdxOut{out_idx, 0}, out_val, flags,
                    (char*)workspace, (hipStream_t)stream);
}
