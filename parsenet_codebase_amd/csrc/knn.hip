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
#include "knn_common.h"

// xx[b,j] = fma chain over channels [c0, c1) of x[b,c,j]^2
__global__ void pn_knn_sqnorm_kernel(const float* __restrict__ x, int C, int N, int c0, int c1,
                                     float* __restrict__ xx, const int* __restrict__ gate_any) {
  const int b = blockIdx.y;
  if (gate_any && !gate_any[b]) return;   // gated fallback: no query of this item was flagged
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
                                       int C, int Nend, int j0, int qc, int N, float xxq,
                                       float tau, int& cnt, u64* __restrict__ mylist) {
  // N: row stride of x (points per item); Nend: one past the last candidate of this slice
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
        const int j = TAIL ? min(j0 + t, Nend - 1) : j0 + t;
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
        const int j = TAIL ? min(j0 + t, Nend - 1) : j0 + t;
        acc[t] = __builtin_fmaf(xq, row[j], acc[t]);
        accn[t] = __builtin_fmaf(nq, rown[j], accn[t]);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < KNN_TC; ++t) {
    const int j = TAIL ? min(j0 + t, Nend - 1) : j0 + t;
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
    const bool ok = TAIL ? (j0 + t < Nend) : true;
    if (ok && v >= tau) {
      mylist[cnt] = knn_key(v, j0 + t);
      ++cnt;
    }
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void pn_knn_scan_kernel(const float* __restrict__ x,
                                                          const float* __restrict__ xx, int C,
                                                          int N, int k, int S, int slice_len,
                                                          u64* __restrict__ lists,
                                                          int* __restrict__ counts,
                                                          KnnIdxOut out,
                                                          const int* __restrict__ gate) {
  // grid: (ceil(N/256), S, B).  Wave = 64 queries x one slice of the candidates.
  // S == 1: the sorted result goes straight to `out`; S > 1: every (query, slice) list is
  // reduced to its k best and pn_knn_merge_kernel finishes the job.
  __shared__ __attribute__((aligned(16))) uint32_t s_hist[4][256];
  const int b = blockIdx.z;
  const int slice = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int qbase = (blockIdx.x * 4 + wave) * 64;
  if (qbase >= N) return;  // wave-uniform
  const int q = qbase + lane;
  const bool qvalid = q < N;
  const int qc = qvalid ? q : N - 1;
  const int Np = (N + 63) & ~63;
  // gated mode (fallback of the MFMA path): only waves owning a flagged query do any work
  if (gate && !__any(qvalid && gate[(size_t)b * N + qc] != 0)) return;
  const float* __restrict__ xb = x + (size_t)b * C * N;
  const float* __restrict__ xxb = xx + (size_t)b * N;
  // list of (b, q, slice): ((b*Np + q)*S + slice)*CAP
  u64* __restrict__ wlists = lists + (((size_t)b * Np + qbase) * S + slice) * KNN_CAP;
  const size_t lstride = (size_t)S * KNN_CAP;  // between consecutive queries
  u64* __restrict__ mylist = wlists + (size_t)lane * lstride;
  uint32_t* hist = s_hist[wave];
  const float xxq = xxb[qc];
  // invalid lanes never pass the filter (comparison with NaN is false)
  float tau = qvalid ? -__builtin_inff() : __builtin_nanf("");
  int cnt = 0;
  int trig = KNN_CAP0 - KNN_TC;
  if (trig < k) trig = k;  // never compact below k entries

  const int j_begin = slice * slice_len;
  const int j_end = min(N, j_begin + slice_len);
  for (int j0 = j_begin; j0 < j_end; j0 += KNN_TC) {
    if (j0 + KNN_TC <= j_end)
      knn_tile<MODE, false>(xb, xxb, C, j_end, j0, qc, N, xxq, tau, cnt, mylist);
    else
      knn_tile<MODE, true>(xb, xxb, C, j_end, j0, qc, N, xxq, tau, cnt, mylist);
    u64 m = __ballot(cnt > trig);
    if (m) {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      while (m) {
        const int L = __builtin_ctzll(m);
        m &= m - 1;
        const int n = __builtin_amdgcn_readlane(cnt, L);
        const u64 kth = knn_wave_select(wlists + (size_t)L * lstride, n, k, hist);
        if (lane == L) {
          cnt = k;
          tau = pn_ord2f((uint32_t)(kth >> 32));
          trig = KNN_CAP - KNN_TC;
        }
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  const int nvalid = min(64, N - qbase);
  for (int L = 0; L < nvalid; ++L) {
    const int n = __builtin_amdgcn_readlane(cnt, L);
    u64* lp = wlists + (size_t)L * lstride;
    if (n > k) knn_wave_select(lp, n, k, hist);
    const int m = n < k ? n : k;  // n < k: short slice (S > 1) or NaNs in the input
    if (S > 1) {
      if (lane == 0) counts[((size_t)b * Np + qbase + L) * S + slice] = m;
      continue;
    }
    u64 k0 = lane < m ? lp[lane] : 0ull;
    u64 k1 = lane + 64 < m ? lp[lane + 64] : 0ull;
    knn_wave_sort128(k0, k1);
    const size_t o = ((size_t)b * N + qbase + L) * k;
    if (lane < k) out.put(o + lane, knn_key_index(k0));
    if (lane + 64 < k) out.put(o + lane + 64, knn_key_index(k1));
  }
}

// One wave per query: gather the <= S*k survivors of its slices into LDS, select the k best,
// sort, emit.  S*k <= KNN_CAP is guaranteed by the launcher.
__global__ __launch_bounds__(256) void pn_knn_merge_kernel(const u64* __restrict__ lists,
                                                           const int* __restrict__ counts, int N,
                                                           int k, int S, long long nq_total,
                                                           KnnIdxOut out) {
  __shared__ __attribute__((aligned(16))) uint32_t s_hist[4][256];
  __shared__ u64 s_keys[4][KNN_CAP];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long qi = (long long)blockIdx.x * 4 + wave;  // index over B*N real queries
  if (qi >= nq_total) return;
  const int Np = (N + 63) & ~63;
  const long long b = qi / N;
  const int q = (int)(qi - b * N);
  const size_t qslot = (size_t)b * Np + q;
  u64* keys = s_keys[wave];
  int n = 0;
  for (int s = 0; s < S; ++s) {
    const int c = counts[qslot * S + s];
    const u64* lp = lists + (qslot * S + s) * KNN_CAP;
    for (int e = lane; e < c; e += 64) keys[n + e] = lp[e];
    n += c;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
  if (n > k) knn_wave_select(keys, n, k, s_hist[wave]);
  const int m = n < k ? n : k;
  u64 k0 = lane < m ? keys[lane] : 0ull;
  u64 k1 = lane + 64 < m ? keys[lane + 64] : 0ull;
  knn_wave_sort128(k0, k1);
  const size_t o = (size_t)qi * k;
  if (lane < k) out.put(o + lane, knn_key_index(k0));
  if (lane + 64 < k) out.put(o + lane + 64, knn_key_index(k1));
}

static void knn_plan(int B, int N, int k, bool single, int* S_out, int* slice_len_out) {
  if (single) {
    *S_out = 1;
    *slice_len_out = (int)pn_align_up(N, KNN_TC);
    return;
  }
  const long long waves_q = (long long)B * pn_cdiv(N, 64);
  int S = (int)(6144 / (waves_q > 0 ? waves_q : 1));
  if (S > 8) S = 8;
  if (S * k > KNN_CAP) S = KNN_CAP / k;
  if (S < 1) S = 1;
  int len = (int)pn_align_up(pn_cdiv(N, S), KNN_TC);
  if (len < 128) len = 128;  // keep slices long enough to be worth a wave
  if (len < k) len = (int)pn_align_up(k, KNN_TC);
  S = pn_cdiv(N, len);
  *S_out = S;
  *slice_len_out = len;
}

size_t pn_knn_v1_workspace(int B, int C, int N, int k, bool gated) {
  (void)C;
  int S, len;
  knn_plan(B, N, k, gated, &S, &len);
  const size_t Np = pn_align_up(N, 64);
  return pn_align_up((size_t)B * N * sizeof(float), 256) +
         pn_align_up((size_t)B * Np * S * sizeof(int), 256) +
         pn_align_up((size_t)B * Np * S * KNN_CAP * sizeof(u64), 256);
}

int pn_knn_v1_launch(int mode, const float* x, int B, int C, int N, int k, KnnIdxOut idx,
                     void* workspace, size_t workspace_bytes, hipStream_t stream,
                     const int* gate, const int* gate_any) {
  PN_CHECK_ARG(x && idx.p, "pn_knn: null pointer");
  PN_CHECK_ARG(B > 0 && C > 0 && N > 0, "pn_knn: empty input (B=%d C=%d N=%d)", B, C, N);
  PN_CHECK_ARG(k >= 1 && k <= KNN_MAXK, "pn_knn: k=%d unsupported (1..%d)", k, KNN_MAXK);
  PN_CHECK_ARG(k <= N, "pn_knn: k=%d exceeds the number of points N=%d", k, N);
  PN_CHECK_ARG(mode == 0 || C == 6, "pn_knn_pn: points+normals metric needs C=6, got %d", C);
  PN_CHECK_ARG(workspace && workspace_bytes >= pn_knn_v1_workspace(B, C, N, k, gate != nullptr),
               "pn_knn: workspace too small");
  int S, slice_len;
  knn_plan(B, N, k, gate != nullptr, &S, &slice_len);
  const size_t Np = pn_align_up(N, 64);
  char* w = (char*)workspace;
  float* xx = (float*)w;
  w += pn_align_up((size_t)B * N * sizeof(float), 256);
  int* counts = (int*)w;
  w += pn_align_up((size_t)B * Np * S * sizeof(int), 256);
  u64* lists = (u64*)w;
  dim3 g1(pn_cdiv(N, 256), B);
  hipLaunchKernelGGL(pn_knn_sqnorm_kernel, g1, dim3(256), 0, stream, x, C, N, 0,
                     mode == 0 ? C : 3, xx, gate ? gate_any : nullptr);
  PN_CHECK_LAUNCH();
  dim3 grid(pn_cdiv(N, 256), S, B);
  PN_PROF(gate ? "knn_scan_gated" : "knn_scan", stream);
  if (mode == 0)
    hipLaunchKernelGGL(pn_knn_scan_kernel<0>, grid, dim3(256), 0, stream, x, xx, C, N, k, S,
                       slice_len, lists, counts, idx, gate);
  else
    hipLaunchKernelGGL(pn_knn_scan_kernel<1>, grid, dim3(256), 0, stream, x, xx, C, N, k, S,
                       slice_len, lists, counts, idx, gate);
  PN_CHECK_LAUNCH();
  if (S > 1) {
    const long long nq = (long long)B * N;
    hipLaunchKernelGGL(pn_knn_merge_kernel, dim3(pn_cdiv(nq, 4)), dim3(256), 0, stream, lists,
                       counts, N, k, S, nq, idx);
    PN_CHECK_LAUNCH();
  }
  return PN_OK;
}

