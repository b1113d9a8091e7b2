// kNN graph build, fast path: exact two-pass selection on the fp32 matrix cores (gfx950).
//
// Replaces src/model.py:9-22, src/PointNet.py:9-26 and :29-69 (N x N GEMM + torch.topk).
//
// The neighbour values are k-ordered fp32 fma chains; v_mfma_f32_32x32x2_f32 evaluates
// exactly such a chain (D = fma(a_k1, b_k1, fma(a_k0, b_k0, C)), one rounding per product,
// no wider accumulation), so the matrix-core result is bit-identical to the oracle's
// scalar loop.  A 32(candidates) x 32(queries) tile is accumulated per wave with the query
// operands resident in VGPRs; candidate operands stream from L2 with one dword per lane per
// k-step (two 128-byte segments per load).
//
// Selection never sorts more than a handful of values per query:
//   K0  gather x into a decorrelated candidate order (golden-ratio affine bijection; tile
//       statistics then do not depend on how the caller ordered the points), pad to
//       multiples of 64 points / 2 channels, compute squared norms.
//   K1  pass 1: for every query and every group of 16 candidates keep only the maximum
//       value ("tile maximum").  The k-th largest tile maximum T_k is a valid threshold:
//       at least k distinct candidates have value >= T_k.
//   K2  wave-per-query bisection on the register-resident tile maxima -> tau
//       (count(tilemax >= tau) >= k, as close to k as the bisection gets).  With N/16 tiles
//       the expected number of candidates >= tau is ~1.07 k.
//   K3  pass 2: recompute the values (same arithmetic), append the survivors (v >= tau) as
//       64-bit (value, index) keys to a sub-list private to (query, slice, half-wave): no
//       atomics, the fill count lives in a register.
//   K4  one wave per query: gather the sub-lists, bitonic sort of <= 128 keys (select first
//       if more), emit the k best indices, best first, ties -> smaller original index.
//   Lists that overflow (degenerate inputs: masses of exactly equal values) are flagged
//   and recomputed by the generic scan kernel (knn.hip), gated on the flags.
#include "knn_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));


size_t pn_knn_v1_workspace(int B, int C, int N, int k, bool gated);
int pn_knn_v1_launch(int mode, const float* x, int B, int C, int N, int k, int64_t* idx,
                     void* workspace, size_t workspace_bytes, hipStream_t stream,
                     const int* gate);

struct KnnPerm {
  uint32_t a, c, mask;
  int n;
};

// bijection on [0, n): affine map modulo 2^m (odd multiplier near 2^m / golden ratio) with
// cycle walking.  Consecutive positions land ~0.618 * 2^m apart.
__host__ __device__ static inline int knn_perm(const KnnPerm& p, int i) {
  uint32_t v = (uint32_t)i;
  do {
    v = (v * p.a + p.c) & p.mask;
  } while (v >= (uint32_t)p.n);
  return (int)v;
}

static KnnPerm knn_make_perm(int N) {
  KnnPerm p;
  int m = 1;
  while ((1u << m) < (uint32_t)N) ++m;
  p.mask = (m >= 32) ? 0xffffffffu : ((1u << m) - 1u);
  p.a = ((uint32_t)(0.6180339887498949 * (double)(1ull << m))) | 1u;
  p.c = (0x9E3779B9u >> (32 - m)) | 1u;
  p.n = N;
  return p;
}

// K0: xp (B, Cp, Np) zero padded, permuted columns; xxp (B, Np) squared norms (fma chain over
// the first `cnorm` source channels).  MODE 1 channel map: [p0 p1 p2 0 n0 n1 n2 0].
__global__ void pn_knn_prep_kernel(const float* __restrict__ x, int C, int N, int Cp, int Np,
                                   int mode, KnnPerm perm, float* __restrict__ xp,
                                   float* __restrict__ xxp) {
  const int b = blockIdx.y;
  const int jp = blockIdx.x * blockDim.x + threadIdx.x;
  if (jp >= Np) return;
  const float* xb = x + (size_t)b * C * N;
  float* xpb = xp + (size_t)b * Cp * Np;
  float acc = 0.f;
  if (jp < N) {
    const int j = knn_perm(perm, jp);
    const int cnorm = mode == 0 ? C : 3;
    for (int c = 0; c < C; ++c) {
      const float v = xb[(size_t)c * N + j];
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

// K1 / K3.  KSTEPS = Cp / 2 (MODE 1: 4 = two steps xyz + two steps normals), QSETS = number of
// 32-query column blocks per wave.
template <int KSTEPS, int QSETS, int MODE, bool COLLECT>
__global__ __launch_bounds__(256) void pn_knn_mfma_kernel(
    const float* __restrict__ xp, const float* __restrict__ xxp, int N, int Np,
    int tiles_per_slice, float* __restrict__ tilemax, const float* __restrict__ tau,
    u64* __restrict__ lists, int* __restrict__ counts, int subcap, KnnPerm perm) {
  const int b = blockIdx.z;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int col = lane & 31, h = lane >> 5;
  const int q0 = (blockIdx.x * 4 + wave) * (32 * QSETS);
  if (q0 >= N) return;  // wave-uniform
  constexpr int CP = 2 * KSTEPS;
  const float* __restrict__ xb = xp + (size_t)b * CP * Np;
  const float* __restrict__ xxb = xxp + (size_t)b * Np;
  const int ntiles = Np / 32;
  const int S = gridDim.y, slice = blockIdx.y;
  const int t_begin = slice * tiles_per_slice;
  const int t_end = min(ntiles, t_begin + tiles_per_slice);
  const int T16 = Np / 16;

  // resident query operands: B[k = lane>>5][j = lane&31] of every k-step
  float bq[QSETS][KSTEPS];
  float xxq[QSETS], tq[QSETS];
  bool qok[QSETS];
  int mycnt[QSETS];
  u64* sub[QSETS];
#pragma unroll
  for (int s = 0; s < QSETS; ++s) {
    const int q = q0 + 32 * s + col;
    const int qcl = q < Np ? q : Np - 1;
#pragma unroll
    for (int m = 0; m < KSTEPS; ++m) bq[s][m] = xb[(size_t)(2 * m + h) * Np + qcl];
    xxq[s] = xxb[qcl];
    qok[s] = q < N;
    tq[s] = (COLLECT && qok[s]) ? tau[(size_t)b * Np + qcl] : __builtin_inff();
    mycnt[s] = 0;
    // sub-list of (query, slice, half): (((b*Np + q)*S + slice)*2 + h) * subcap
    sub[s] = lists + ((((size_t)b * Np + qcl) * S + slice) * 2 + h) * (size_t)subcap;
  }

  for (int mt = t_begin; mt < t_end; ++mt) {
    const int j0 = mt * 32;
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
    // A[i = lane&31][k = lane>>5]: candidate j0+col, channel 2m+h; streamed in chunks of at
    // most 32 k-steps so that the operand window stays small for wide features
    constexpr int KCH = KSTEPS < 32 ? KSTEPS : 32;
#pragma unroll
    for (int m0 = 0; m0 < KSTEPS; m0 += KCH) {
      float av[KCH];
#pragma unroll
      for (int m = 0; m < KCH; ++m) av[m] = xb[(size_t)(2 * (m0 + m) + h) * Np + j0 + col];
#pragma unroll
      for (int m = 0; m < KCH; ++m) {
#pragma unroll
        for (int s = 0; s < QSETS; ++s) {
          if (MODE == 1 && m0 + m >= 2)
            accn[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bq[s][m0 + m], accn[s], 0, 0, 0);
          else
            acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bq[s][m0 + m], acc[s], 0, 0, 0);
        }
      }
    }
    // D[i][j]: lane holds column j = col (query), rows i = (r&3) + 8*(r>>2) + 4*h (candidates)
    float xxj[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 t4 = *reinterpret_cast<const float4*>(&xxb[j0 + 8 * g + 4 * h]);
      xxj[4 * g + 0] = t4.x;
      xxj[4 * g + 1] = t4.y;
      xxj[4 * g + 2] = t4.z;
      xxj[4 * g + 3] = t4.w;
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
        } else {
          const float t = __builtin_fmaf(-2.0f, acc[s][r], xxj[r]);  // xx[j] - 2*dot_p
          const float pp = __fadd_rn(t, xxq[s]);
          const float pn = __builtin_fmaf(-2.0f, accn[s][r], 2.0f);  // 2 - 2*dot_n
          v = -__fmul_rn(pp, __fadd_rn(1.0f, pn));
        }
        const bool ok = (j0 + row) < N;
        if (!COLLECT) {
          tm = fmaxf(tm, ok ? v : -__builtin_inff());
        } else {
          if (ok && v >= tq[s]) {
            // the key carries the PERMUTED index here; K4 rewrites it to the original one
            if (mycnt[s] < subcap) sub[s][mycnt[s]] = knn_key(v, j0 + row);
            ++mycnt[s];
          }
        }
      }
      if (!COLLECT) {
        const int q = q0 + 32 * s + col;
        if (q < Np) tilemax[((size_t)b * Np + q) * T16 + (2 * mt + h)] = tm;
      }
    }
  }
  if (COLLECT) {
#pragma unroll
    for (int s = 0; s < QSETS; ++s) {
      const int q = q0 + 32 * s + col;
      if (q < Np) counts[(((size_t)b * Np + q) * S + slice) * 2 + h] = mycnt[s];
    }
  }
}

// K2: one wave per query; the T = Np/16 tile maxima of the query are contiguous and live in
// registers (16 per lane cover N <= 16384; the rare remainder is re-read each round).
// Bisection on order-preserving uint keys for tau with count(tilemax >= tau) >= k, stopping
// as soon as the count is within a small slack of k.
#define KM_TR 16
__global__ __launch_bounds__(256) void pn_knn_tau_kernel(const float* __restrict__ tilemax, int N,
                                                         int Np, int k, float* __restrict__ tau) {
  const int b = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int q = blockIdx.x * 4 + wave;
  if (q >= N) return;
  const int T = Np / 16;
  const float* __restrict__ row = tilemax + ((size_t)b * Np + q) * T;
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
  uint32_t lo = pn_f2ord(vmin);       // count(>= lo) = #finite tiles >= 2k (host checked)
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
  if (lane == 0) tau[(size_t)b * Np + q] = pn_ord2f(lo);
}

// K4: one wave per (permuted) query: gather its 2*S sub-lists into LDS, sort, emit.
__global__ __launch_bounds__(256) void pn_knn_final_kernel(const u64* __restrict__ lists,
                                                           const int* __restrict__ counts, int N,
                                                           int Np, int k, int S, int subcap,
                                                           KnnPerm perm,
                                                           int64_t* __restrict__ out,
                                                           int* __restrict__ flags) {
  __shared__ __attribute__((aligned(16))) uint32_t s_hist[4][256];
  __shared__ u64 s_keys[4][KNN_CAP];
  const int b = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int qp = blockIdx.x * 4 + wave;
  if (qp >= N) return;
  const size_t ql = (size_t)b * Np + qp;
  const int qo = knn_perm(perm, qp);
  u64* keys = s_keys[wave];
  // all 2S fill counts with one coalesced load, then register-only bookkeeping
  const int nsub = 2 * S;  // <= 16
  const int myc = lane < nsub ? counts[ql * nsub + lane] : 0;
  int n = 0;
  bool bad = false;
  for (int s = 0; s < nsub; ++s) {
    const int c = __builtin_amdgcn_readlane(myc, s);
    if (c > subcap || n + c > KNN_CAP) {
      bad = true;
      break;
    }
    const u64* lp = lists + (ql * nsub + s) * (size_t)subcap;
    for (int e = lane; e < c; e += 64) {
      const u64 key = lp[e];
      // candidate index: permuted -> original, so that ties order by the caller's indices
      const int jp = (int)(0xffffffffu - (uint32_t)(key & 0xffffffffu));
      keys[n + e] = (key & 0xffffffff00000000ull) |
                    (u64)(0xffffffffu - (uint32_t)knn_perm(perm, jp));
    }
    n += c;
  }
  if (bad || n < k) {
    // overflow (or NaNs): recomputed by the gated generic kernel
    if (lane == 0) flags[(size_t)b * N + qo] = 1;
    return;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
  if (n > 128) knn_wave_select(keys, n, k, s_hist[wave]);
  const int m = n > 128 ? k : n;
  u64 k0 = lane < m ? keys[lane] : 0ull;
  u64 k1 = lane + 64 < m ? keys[lane + 64] : 0ull;
  knn_wave_sort128(k0, k1);
  int64_t* o = out + ((size_t)b * N + qo) * k;
  if (lane < k) o[lane] = knn_key_index(k0);
  if (lane + 64 < k) o[lane + 64] = knn_key_index(k1);
}

// ---------------------------------------------------------------------------------------
struct KnnPlan {
  bool fast;
  int ksteps, qsets, Cp, Np, S, tiles_per_slice, subcap;
};

static KnnPlan knn_mfma_plan(int mode, int B, int C, int N, int k) {
  KnnPlan p;
  memset(&p, 0, sizeof(p));
  p.Np = (int)pn_align_up(N, 64);
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
  // the threshold needs at least 2k tile maxima to be tight; small clouds use the scan path
  p.fast = p.ksteps > 0 && (N / 16) >= 2 * k && k <= KNN_MAXK;
  if (p.fast) {
    const long long waves_q = (long long)B * pn_cdiv(p.Np, 32 * p.qsets);
    int S = (int)(8192 / (waves_q > 0 ? waves_q : 1));
    const int ntiles = p.Np / 32;
    if (S > 8) S = 8;
    if (S > ntiles) S = ntiles;
    if (S < 1) S = 1;
    p.tiles_per_slice = pn_cdiv(ntiles, S);
    p.S = pn_cdiv(ntiles, p.tiles_per_slice);
    // expected survivors per sub-list ~ 1.1 k / (2 S); leave generous head-room
    p.subcap = (int)pn_align_up(3 * k / (2 * p.S) + 16, 8);
    if (2 * p.S * p.subcap > KNN_CAP) p.subcap = KNN_CAP / (2 * p.S);
  }
  return p;
}

struct KnnWs {
  size_t xp, xxp, tilemax, tau, cnt, flags, lists, v1, total;
};

static KnnWs knn_mfma_ws(const KnnPlan& p, int B, int C, int N, int k) {
  KnnWs w;
  size_t o = 0;
  auto take = [&](size_t bytes) {
    size_t at = o;
    o += pn_align_up(bytes, 256);
    return at;
  };
  w.xp = take((size_t)B * p.Cp * p.Np * 4);
  w.xxp = take((size_t)B * p.Np * 4);
  w.tilemax = take((size_t)B * (p.Np / 16) * p.Np * 4);
  w.tau = take((size_t)B * p.Np * 4);
  w.cnt = take((size_t)B * p.Np * 2 * p.S * 4);
  w.flags = take((size_t)B * N * 4);
  w.lists = take((size_t)B * p.Np * 2 * p.S * p.subcap * 8);
  w.v1 = take(pn_knn_v1_workspace(B, C, N, k, true));
  w.total = o;
  return w;
}

extern "C" size_t pn_knn_workspace(int B, int C, int N, int k) {
  // mode 1 (points+normals) has C = 6 and plans like mode 0 with ksteps = 4
  KnnPlan p = knn_mfma_plan(C == 6 ? 1 : 0, B, C, N, k);
  KnnPlan p0 = knn_mfma_plan(0, B, C, N, k);
  size_t a = p.fast ? knn_mfma_ws(p, B, C, N, k).total : pn_knn_v1_workspace(B, C, N, k, false);
  size_t b = p0.fast ? knn_mfma_ws(p0, B, C, N, k).total : pn_knn_v1_workspace(B, C, N, k, false);
  return a > b ? a : b;
}

template <int KSTEPS, int QSETS, int MODE>
static void knn_mfma_launch_pass(bool collect, dim3 grid, hipStream_t stream, const float* xp,
                                 const float* xxp, int N, int Np, int tps, float* tilemax,
                                 const float* tau, u64* lists, int* cnt, int subcap,
                                 KnnPerm perm) {
  if (!collect)
    hipLaunchKernelGGL((pn_knn_mfma_kernel<KSTEPS, QSETS, MODE, false>), grid, dim3(256), 0,
                       stream, xp, xxp, N, Np, tps, tilemax, tau, lists, cnt, subcap, perm);
  else
    hipLaunchKernelGGL((pn_knn_mfma_kernel<KSTEPS, QSETS, MODE, true>), grid, dim3(256), 0,
                       stream, xp, xxp, N, Np, tps, tilemax, tau, lists, cnt, subcap, perm);
}

static int knn_dispatch(int mode, const float* x, int B, int C, int N, int k, int64_t* idx,
                        void* workspace, size_t workspace_bytes, hipStream_t stream) {
  PN_CHECK_ARG(x && idx, "pn_knn: null pointer");
  PN_CHECK_ARG(B > 0 && C > 0 && N > 0, "pn_knn: empty input (B=%d C=%d N=%d)", B, C, N);
  PN_CHECK_ARG(k >= 1 && k <= KNN_MAXK, "pn_knn: k=%d unsupported (1..%d)", k, KNN_MAXK);
  PN_CHECK_ARG(k <= N, "pn_knn: k=%d exceeds the number of points N=%d", k, N);
  PN_CHECK_ARG(mode == 0 || C == 6, "pn_knn_pn: points+normals metric needs C=6, got %d", C);
  PN_CHECK_ARG(workspace && workspace_bytes >= pn_knn_workspace(B, C, N, k),
               "pn_knn: workspace too small");
  const KnnPlan p = knn_mfma_plan(mode, B, C, N, k);
  if (!p.fast)
    return pn_knn_v1_launch(mode, x, B, C, N, k, idx, workspace, workspace_bytes, stream, nullptr);

  const KnnWs w = knn_mfma_ws(p, B, C, N, k);
  char* base = (char*)workspace;
  float* xp = (float*)(base + w.xp);
  float* xxp = (float*)(base + w.xxp);
  float* tilemax = (float*)(base + w.tilemax);
  float* tau = (float*)(base + w.tau);
  int* cnt = (int*)(base + w.cnt);
  int* flags = (int*)(base + w.flags);
  u64* lists = (u64*)(base + w.lists);
  const KnnPerm perm = knn_make_perm(N);

  PN_CHECK_HIP(hipMemsetAsync(flags, 0, (size_t)B * N * 4, stream));
  {
    PN_PROF("knn_prep", stream);
    hipLaunchKernelGGL(pn_knn_prep_kernel, dim3(pn_cdiv(p.Np, 256), B), dim3(256), 0, stream, x,
                     C, N, p.Cp, p.Np, mode, perm, xp, xxp);
  }
  PN_CHECK_LAUNCH();
  dim3 grid(pn_cdiv(p.Np, 32 * p.qsets * 4), p.S, B);
  for (int pass = 0; pass < 2; ++pass) {
    const bool collect = pass == 1;
    static const char* const pass_names[2][4] = {
        {"knn_mfma_pass1_c4", "knn_mfma_pass1_c64", "knn_mfma_pass1_wide", "knn_mfma_pass1_pn"},
        {"knn_mfma_pass2_c4", "knn_mfma_pass2_c64", "knn_mfma_pass2_wide", "knn_mfma_pass2_pn"}};
    const int fam = mode == 1 ? 3 : (p.ksteps <= 4 ? 0 : (p.ksteps == 32 ? 1 : 2));
    {
    PN_PROF(pass_names[pass][fam], stream);
#define KM_GO(KS, QS, MD)                                                                     \
  knn_mfma_launch_pass<KS, QS, MD>(collect, grid, stream, xp, xxp, N, p.Np, p.tiles_per_slice, \
                                   tilemax, tau, lists, cnt, p.subcap, perm)
    if (mode == 1)
      KM_GO(4, 2, 1);
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
    if (!collect) {
      PN_PROF("knn_tau", stream);
      hipLaunchKernelGGL(pn_knn_tau_kernel, dim3(pn_cdiv(N, 4), B), dim3(256), 0, stream,
                         tilemax, N, p.Np, k, tau);
      PN_CHECK_LAUNCH();
    }
  }
  {
    PN_PROF("knn_final", stream);
    hipLaunchKernelGGL(pn_knn_final_kernel, dim3(pn_cdiv(N, 4), B), dim3(256), 0, stream, lists,
                       cnt, N, p.Np, k, p.S, p.subcap, perm, idx, flags);
  }
  PN_CHECK_LAUNCH();
  PN_PROF("knn_fallback_gate", stream);
  // degenerate queries (flagged) are redone by the generic scan kernel; waves without a
  // flagged query exit immediately
  return pn_knn_v1_launch(mode, x, B, C, N, k, idx, base + w.v1,
                          pn_knn_v1_workspace(B, C, N, k, true), stream, flags);
}

extern "C" int pn_knn_f32(const float* x, int B, int C, int N, int k, int64_t* idx,
                          void* workspace, size_t workspace_bytes, void* stream) {
  return knn_dispatch(0, x, B, C, N, k, idx, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int pn_knn_pn_f32(const float* x6, int B, int N, int k, int64_t* idx, void* workspace,
                             size_t workspace_bytes, void* stream) {
  return knn_dispatch(1, x6, B, 6, N, k, idx, workspace, workspace_bytes, (hipStream_t)stream);
}
