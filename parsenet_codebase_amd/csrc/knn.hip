// kNN graph build for gfx950 — replaces the per-batch-item N x N GEMM + torch.topk of
//   src/model.py:9-22        knn(x, k)
//   src/PointNet.py:9-26     knn(x, k1, k2)
//   src/PointNet.py:29-69    knn_points_normals(x, k1, k2)
// without ever materialising the N x N matrix.
//
// Mapping (wave64, no inter-wave communication at all):
//   * one QUERY per lane: the lane streams its own feature x[b,c,q] from HBM/L2
//     (coalesced across the wave for every channel c),
//   * a tile of 32 CANDIDATES per step, wave-uniform: their features are read with
//     scalar loads (s_load_dwordx16) and fed to v_fma as SGPR operands, so each
//     candidate dword is fetched once per wave and reused by 64 queries,
//   * 32 fp32 accumulators per lane hold the k-ordered fmaf chains
//       dot(i,j) = fma(x[C-1,i], x[C-1,j], ... fma(x[0,i], x[0,j], 0)),
//     which is bit-for-bit what the oracle (oracle/c/pn_oracle.c: pno_knn) evaluates,
//   * selection: a lane appends (value, index) keys that beat its running threshold to
//     its private list in the workspace; when a list fills up, the WAVE cooperatively
//     radix-selects its k best (LDS histogram, 8 bit digits, early exit) and tightens the
//     lane's threshold; at the end the k survivors are bitonic-sorted in registers.
//
// Value semantics follow the reference exactly, including the order of the roundings:
//   MODE 0:  v = (-xx[j] - (-2*dot)) - xx[i]                      (model.py:14-16)
//   MODE 1:  p = (xxp[j] - 2*dot_p) + xxp[i]; n = 2 - 2*dot_n;
//            v = -(p * (1 + n))                                     (PointNet.py:41-59)
// The k LARGEST v are returned, best first; ties in v resolve to the smaller index
// (torch.topk leaves ties unspecified).
#include "common.h"

#define KNN_TC 32        // candidates per step (accumulators per lane)
#define KNN_CAP 1024     // list capacity per query (keys)
#define KNN_CAP0 256     // first compaction after this many keys (tightens tau early)
#define KNN_MAXK 128
#define KNN_EPL (KNN_CAP / 64)  // list entries per lane during a wave-wide select

typedef unsigned long long u64;

__device__ static inline u64 knn_key(float v, int j) {
  return ((u64)pn_f2ord(v) << 32) | (u64)(0xffffffffu - (uint32_t)j);
}

__device__ static inline u64 readlane_u64(u64 v, int l) {
  uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, l);
  uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(v >> 32), l);
  return ((u64)hi << 32) | lo;
}

// Wave-cooperative: among the n keys at lp[0..n) keep the k largest (compacted to
// lp[0..k), unordered) and return the k-th largest key.  Requires k <= n <= KNN_CAP.
__device__ static u64 knn_wave_select(u64* __restrict__ lp, int n, int k,
                                      uint32_t* __restrict__ hist) {
  const int lane = threadIdx.x & 63;
  u64 key[KNN_EPL];
#pragma unroll
  for (int e = 0; e < KNN_EPL; ++e) {
    int s = e * 64 + lane;
    key[e] = (s < n) ? lp[s] : 0ull;
  }
  u64 prefix = 0, pmask = 0, kth = 0;
  int rem = k;
  bool done = false;
  for (int p = 7; p >= 0 && !done; --p) {
    const int sh = p * 8;
    reinterpret_cast<uint4*>(hist)[lane] = make_uint4(0, 0, 0, 0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int e = 0; e < KNN_EPL; ++e) {
      if (e * 64 + lane < n && (key[e] & pmask) == prefix)
        atomicAdd(&hist[(uint32_t)(key[e] >> sh) & 255u], 1u);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    uint4 h = reinterpret_cast<uint4*>(hist)[lane];
    int hb[4] = {(int)h.x, (int)h.y, (int)h.z, (int)h.w};
    int tot = hb[0] + hb[1] + hb[2] + hb[3];
    // inclusive prefix over lanes, then exclusive suffix (bins above this lane's)
    int inc = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      int t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    int total = __builtin_amdgcn_readlane(inc, 63);
    int cum = total - inc;
    int found = -1, newrem = 0, fcount = 0;
#pragma unroll
    for (int bb = 3; bb >= 0; --bb) {
      int c = hb[bb];
      if (found < 0 && cum < rem && cum + c >= rem) {
        found = lane * 4 + bb;
        newrem = rem - cum;
        fcount = c;
      }
      cum += c;
    }
    u64 fm = __ballot(found >= 0);
    int src = __builtin_ctzll(fm);
    int bin = __builtin_amdgcn_readlane(found, src);
    rem = __builtin_amdgcn_readlane(newrem, src);
    int bc = __builtin_amdgcn_readlane(fcount, src);
    prefix |= (u64)bin << sh;
    pmask |= 0xffull << sh;
    if (bc == 1) {
      // the k-th key is the only one with this prefix: fetch it and stop early
      u64 cand = 0;
#pragma unroll
      for (int e = 0; e < KNN_EPL; ++e)
        if (e * 64 + lane < n && (key[e] & pmask) == prefix) cand = key[e];
      u64 cm = __ballot(cand != 0);
      kth = readlane_u64(cand, __builtin_ctzll(cm));
      done = true;
    }
  }
  if (!done) kth = prefix;
  // compact: exactly k keys are >= kth because keys are pairwise distinct
  int mine = 0;
#pragma unroll
  for (int e = 0; e < KNN_EPL; ++e) mine += (e * 64 + lane < n && key[e] >= kth) ? 1 : 0;
  int inc = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  int off = inc - mine;
#pragma unroll
  for (int e = 0; e < KNN_EPL; ++e)
    if (e * 64 + lane < n && key[e] >= kth) lp[off++] = key[e];
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
  return kth;
}

// Bitonic sort (descending) of 128 keys held as 2 per lane: element e = r*64 + lane.
__device__ static inline void knn_wave_sort128(u64& k0, u64& k1) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int size = 2; size <= 128; size <<= 1) {
#pragma unroll
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      if (stride == 64) {
        u64 a = k0 > k1 ? k0 : k1, b = k0 > k1 ? k1 : k0;
        k0 = a;
        k1 = b;
      } else {
        const bool lower = (lane & stride) == 0;
        {
          u64 pv = __shfl_xor(k0, stride, 64);
          const bool up = ((lane & size) == 0);  // e = lane (bit 6 clear)
          const bool keep_max = (lower == up);
          k0 = keep_max ? (k0 > pv ? k0 : pv) : (k0 < pv ? k0 : pv);
        }
        {
          u64 pv = __shfl_xor(k1, stride, 64);
          const bool up = (((lane + 64) & size) == 0);
          const bool keep_max = (lower == up);
          k1 = keep_max ? (k1 > pv ? k1 : pv) : (k1 < pv ? k1 : pv);
        }
      }
    }
  }
}

// xx[b,j] = fma chain over channels [c0, c1) of x[b,c,j]^2
__global__ void pn_knn_sqnorm_kernel(const float* __restrict__ x, int C, int N, int c0, int c1,
                                     float* __restrict__ xx) {
  const int b = blockIdx.y;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= N) return;
  const float* xb = x + (size_t)b * C * N;
  float acc = 0.f;
  for (int c = c0; c < c1; ++c) {
    float v = xb[(size_t)c * N + j];
    acc = __builtin_fmaf(v, v, acc);
  }
  xx[(size_t)b * N + j] = acc;
}

template <int MODE, bool TAIL>
__device__ static inline void knn_tile(const float* __restrict__ xb, const float* __restrict__ xxb,
                                       int C, int N, int j0, int qc, float xxq, float tau,
                                       int& cnt, u64* __restrict__ mylist) {
  float acc[KNN_TC];
  float accn[MODE == 1 ? KNN_TC : 1];
#pragma unroll
  for (int t = 0; t < KNN_TC; ++t) acc[t] = 0.f;
  if (MODE == 1) {
#pragma unroll
    for (int t = 0; t < KNN_TC; ++t) accn[t] = 0.f;
  }
  if (MODE == 0) {
    for (int c = 0; c < C; ++c) {
      const float xq = xb[(size_t)c * N + qc];
      const float* __restrict__ row = xb + (size_t)c * N;  // wave-uniform
#pragma unroll
      for (int t = 0; t < KNN_TC; ++t) {
        const int j = TAIL ? min(j0 + t, N - 1) : j0 + t;
        acc[t] = __builtin_fmaf(xq, row[j], acc[t]);
      }
    }
  } else {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float xq = xb[(size_t)c * N + qc];
      const float nq = xb[(size_t)(c + 3) * N + qc];
      const float* __restrict__ row = xb + (size_t)c * N;
      const float* __restrict__ rown = xb + (size_t)(c + 3) * N;
#pragma unroll
      for (int t = 0; t < KNN_TC; ++t) {
        const int j = TAIL ? min(j0 + t, N - 1) : j0 + t;
        acc[t] = __builtin_fmaf(xq, row[j], acc[t]);
        accn[t] = __builtin_fmaf(nq, rown[j], accn[t]);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < KNN_TC; ++t) {
    const int j = TAIL ? min(j0 + t, N - 1) : j0 + t;
    const float xxj = xxb[j];  // wave-uniform
    float v;
    if (MODE == 0) {
      const float inner = __fmul_rn(-2.0f, acc[t]);
      v = __fsub_rn(__fsub_rn(-xxj, inner), xxq);
    } else {
      const float inner = __fmul_rn(2.0f, acc[t]);
      const float pp = __fadd_rn(__fsub_rn(xxj, inner), xxq);
      const float pn = __fsub_rn(2.0f, __fmul_rn(2.0f, accn[t]));
      v = -__fmul_rn(pp, __fadd_rn(1.0f, pn));
    }
    const bool ok = TAIL ? (j0 + t < N) : true;
    if (ok && v >= tau) {
      mylist[cnt] = knn_key(v, j0 + t);
      ++cnt;
    }
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void pn_knn_kernel(const float* __restrict__ x,
                                                     const float* __restrict__ xx, int C, int N,
                                                     int k, u64* __restrict__ lists,
                                                     int64_t* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) uint32_t s_hist[4][256];
  const int b = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int qbase = (blockIdx.x * 4 + wave) * 64;
  if (qbase >= N) return;  // wave-uniform
  const int q = qbase + lane;
  const bool qvalid = q < N;
  const int qc = qvalid ? q : N - 1;
  const float* __restrict__ xb = x + (size_t)b * C * N;
  const float* __restrict__ xxb = xx + (size_t)b * N;
  u64* __restrict__ wlists = lists + ((size_t)b * N + qbase) * KNN_CAP;  // wave-uniform
  u64* __restrict__ mylist = wlists + (size_t)lane * KNN_CAP;
  uint32_t* hist = s_hist[wave];
  const float xxq = xxb[qc];
  // invalid lanes never pass the filter (comparison with NaN is false)
  float tau = qvalid ? -__builtin_inff() : __builtin_nanf("");
  int cnt = 0;
  int trig = KNN_CAP0 - KNN_TC;
  if (trig < k) trig = k;  // never compact below k entries

  for (int j0 = 0; j0 < N; j0 += KNN_TC) {
    if (j0 + KNN_TC <= N)
      knn_tile<MODE, false>(xb, xxb, C, N, j0, qc, xxq, tau, cnt, mylist);
    else
      knn_tile<MODE, true>(xb, xxb, C, N, j0, qc, xxq, tau, cnt, mylist);
    u64 m = __ballot(cnt > trig);
    if (m) {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      while (m) {
        const int L = __builtin_ctzll(m);
        m &= m - 1;
        const int n = __builtin_amdgcn_readlane(cnt, L);
        const u64 kth = knn_wave_select(wlists + (size_t)L * KNN_CAP, n, k, hist);
        if (lane == L) {
          cnt = k;
          tau = pn_ord2f((uint32_t)(kth >> 32));
          trig = KNN_CAP - KNN_TC;
        }
      }
    }
  }
  // final: reduce every list to its k best, sort, emit indices (best first)
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  const int nvalid = min(64, N - qbase);
  for (int L = 0; L < nvalid; ++L) {
    const int n = __builtin_amdgcn_readlane(cnt, L);
    u64* lp = wlists + (size_t)L * KNN_CAP;
    if (n > k) knn_wave_select(lp, n, k, hist);
    const int m = n < k ? n : k;  // n < k only if the input held NaNs
    u64 k0 = lane < m ? lp[lane] : 0ull;
    u64 k1 = lane + 64 < m ? lp[lane + 64] : 0ull;
    knn_wave_sort128(k0, k1);
    int64_t* o = out + ((size_t)b * N + qbase + L) * k;
    if (lane < k) o[lane] = k0 ? (int64_t)(0xffffffffu - (uint32_t)(k0 & 0xffffffffu)) : 0;
    if (lane + 64 < k) o[lane + 64] = k1 ? (int64_t)(0xffffffffu - (uint32_t)(k1 & 0xffffffffu)) : 0;
  }
}

extern "C" size_t pn_knn_workspace(int B, int C, int N, int k) {
  (void)C;
  (void)k;
  return pn_align_up((size_t)B * N * sizeof(float), 256) +
         pn_align_up((size_t)B * pn_align_up(N, 64) * KNN_CAP * sizeof(u64), 256);
}

static int knn_launch(int mode, const float* x, int B, int C, int N, int k, int64_t* idx,
                      void* workspace, size_t workspace_bytes, hipStream_t stream) {
  PN_CHECK_ARG(x && idx, "pn_knn: null pointer");
  PN_CHECK_ARG(B > 0 && C > 0 && N > 0, "pn_knn: empty input (B=%d C=%d N=%d)", B, C, N);
  PN_CHECK_ARG(k >= 1 && k <= KNN_MAXK, "pn_knn: k=%d unsupported (1..%d)", k, KNN_MAXK);
  PN_CHECK_ARG(k <= N, "pn_knn: k=%d exceeds the number of points N=%d", k, N);
  PN_CHECK_ARG(mode == 0 || C == 6, "pn_knn_pn: points+normals metric needs C=6, got %d", C);
  PN_CHECK_ARG(workspace && workspace_bytes >= pn_knn_workspace(B, C, N, k),
               "pn_knn: workspace too small");
  float* xx = (float*)workspace;
  u64* lists = (u64*)((char*)workspace + pn_align_up((size_t)B * N * sizeof(float), 256));
  dim3 g1(pn_cdiv(N, 256), B);
  hipLaunchKernelGGL(pn_knn_sqnorm_kernel, g1, dim3(256), 0, stream, x, C, N, 0,
                     mode == 0 ? C : 3, xx);
  PN_CHECK_LAUNCH();
  dim3 grid(pn_cdiv(N, 256), B);
  if (mode == 0)
    hipLaunchKernelGGL(pn_knn_kernel<0>, grid, dim3(256), 0, stream, x, xx, C, N, k, lists, idx);
  else
    hipLaunchKernelGGL(pn_knn_kernel<1>, grid, dim3(256), 0, stream, x, xx, C, N, k, lists, idx);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

extern "C" int pn_knn_f32(const float* x, int B, int C, int N, int k, int64_t* idx,
                          void* workspace, size_t workspace_bytes, void* stream) {
  return knn_launch(0, x, B, C, N, k, idx, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int pn_knn_pn_f32(const float* x6, int B, int N, int k, int64_t* idx, void* workspace,
                             size_t workspace_bytes, void* stream) {
  return knn_launch(1, x6, B, 6, N, k, idx, workspace, workspace_bytes, (hipStream_t)stream);
}
