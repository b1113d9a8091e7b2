// k nearest neighbours of 3-D points from coordinate DIFFERENCES, for a ragged batch of small clouds:
// the neighbour searches of the evaluation-mode fitting path,
//   src/fitting_utils.py:150-164, 202-237  up_sample_points_torch(_in_range): centroid of the 4 nearest
//                                           neighbours of every point (k = 5 with the point itself),
//   src/fitting_utils.py:704-710           remove_outliers -> open3d 0.9 remove_statistical_outlier(20, 0.5):
//                                           mean distance to the 20 nearest neighbours, in float64,
// on the segments of one shape (a few hundred to a few thousand points each).
//
// Why not the kNN engine of knn_mfma.hip: that one evaluates the reference's GEMM form
// |x|^2 + |y|^2 - 2 x.y (src/model.py:9-22), whose rounding (~1e-7 absolute) reorders neighbours once the
// spacing of the points reaches 1e-3 — which up-sampled segments do.  The reference's up-sampling
// (a broadcast difference, fitting_utils.py:155-158) and open3d's KD-tree work on differences:
//   d(i, j) = ((x_i - x_j)^2 + (y_i - y_j)^2) + (z_i - z_j)^2,   every operation rounded once, no fma.
//
// One WAVE per query.  A lane evaluates the candidates lane, lane + 64, ... of the query's segment and keeps
// their negated distances in registers (E per lane, n <= 64 E); the k-th largest value is found by bisection
// over the order-preserving integer image of the values (32 or 64 rounds of compare + ballot + popcount: no
// sorting of n values, no lists); the k survivors — ties at the k-th value go to the smaller index — are
// compacted through LDS and sorted by one 128-key bitonic network (k <= 64).
#include "knn_common.h"

template <typename T>
struct Knn3Ord;
template <>
struct Knn3Ord<float> {
  typedef uint32_t U;
  static constexpr int BITS = 32;
  __device__ static inline U ord(float f) { return pn_f2ord(f); }
  __device__ static inline float val(U o) { return pn_ord2f(o); }
};
template <>
struct Knn3Ord<double> {
  typedef unsigned long long U;
  static constexpr int BITS = 64;
  __device__ static inline U ord(double f) {
    const U u = (U)__double_as_longlong(f);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
  }
  __device__ static inline double val(U o) {
    const U u = (o >> 63) ? (o & 0x7fffffffffffffffull) : ~o;
    return __longlong_as_double((long long)u);
  }
};

// pts (total, 3) fp32, off (S + 1) segment offsets into the rows of pts; idx (total, k) int32 LOCAL indices
// (into the segment), nearest first, the point itself first (distance 0; equal distances -> smaller index);
// dist (total, k) of type T or null: the distances sqrt(d) of those neighbours (remove_outliers' statistic).
// grid (ceil(max n / 4), S), 4 waves = 4 queries per workgroup.
template <typename T, int E>
__global__ __launch_bounds__(256) void pn_knn3_kernel(const float* __restrict__ pts, const int* __restrict__ off,
                                                      int k, int* __restrict__ idx, T* __restrict__ dist) {
  typedef typename Knn3Ord<T>::U U;
  __shared__ u64 s_keys[4][64];
  const int s = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int o0 = off[s], n = off[s + 1] - o0;
  const int q = blockIdx.x * 4 + wave;
  if (q >= n) return;
  const float* __restrict__ p = pts + (size_t)o0 * 3;
  const T qx = (T)p[3 * q], qy = (T)p[3 * q + 1], qz = (T)p[3 * q + 2];
  U v[E];     // order-preserving integer images of the negated distances (larger = nearer)
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int j = e * 64 + lane;
    T d = (T)__builtin_inff();
    if (j < n) {
      const T dx = qx - (T)p[3 * j], dy = qy - (T)p[3 * j + 1], dz = qz - (T)p[3 * j + 2];
      d = (dx * dx + dy * dy) + dz * dz;      // (-ffp-contract=off: three products and two sums, each rounded once)
    }
    v[e] = Knn3Ord<T>::ord(-d);
  }
  const int kk = k < n ? k : n;
  // largest t with count(ord(v) >= t) >= kk: bit by bit from the top
  U t = 0;
#pragma unroll 1
  for (int bit = Knn3Ord<T>::BITS - 1; bit >= 0; --bit) {
    const U cand = t | ((U)1 << bit);
    int c = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) c += __builtin_popcountll(__ballot(v[e] >= cand));
    if (c >= kk) t = cand;
  }
  int above = 0;
#pragma unroll
  for (int e = 0; e < E; ++e) above += __builtin_popcountll(__ballot(v[e] > t));
  int need_eq = kk - above;      // ties at the k-th value: the first need_eq in index order (e major, lane minor)
  int base = 0;
  u64* keys = s_keys[wave];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const U o = v[e];
    const bool valid = e * 64 + lane < n;
    const u64 meq = __ballot(valid && o == t);
    const bool take_eq = valid && o == t && pn_mbcnt(meq) < need_eq;
    const bool sel = (valid && o > t) || take_eq;
    const u64 msel = __ballot(sel);
    if (sel) {
      // sort key: the fp32 image of the value orders the survivors; survivors closer than fp32 resolves
      // (float64 mode) fall back on the index, like exact ties
      keys[base + pn_mbcnt(msel)] = knn_key((float)Knn3Ord<T>::val(o), e * 64 + lane);
    }
    base += __builtin_popcountll(msel);
    const int eqc = __builtin_popcountll(meq);
    need_eq -= eqc < need_eq ? eqc : need_eq;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
  u64 k0 = lane < kk ? keys[lane] : 0ull, k1 = 0ull;
  knn_wave_sort128(k0, k1);
  if (lane < k) {
    const size_t o = ((size_t)o0 + q) * k + lane;
    const int j = lane < kk ? (int)knn_key_index(k0) : q;     // fewer than k points in the segment: pad with self
    idx[o] = j;
    if (dist) {
      const T dx = qx - (T)p[3 * j], dy = qy - (T)p[3 * j + 1], dz = qz - (T)p[3 * j + 2];
      dist[o] = (T)sqrt((double)((dx * dx + dy * dy) + dz * dz));
    }
  }
}

template <typename T>
static int knn3_launch(const float* pts, const int* off, int S, int max_n, int k, int* idx, T* dist,
                       hipStream_t stream) {
  dim3 grid(pn_cdiv(max_n, 4), S);
#define KNN3_GO(E_) hipLaunchKernelGGL((pn_knn3_kernel<T, E_>), grid, dim3(256), 0, stream, pts, off, k, idx, dist)
  if (max_n <= 64 * 16)
    KNN3_GO(16);
  else if (max_n <= 64 * 40)
    KNN3_GO(40);
  else if (max_n <= 64 * 80)
    KNN3_GO(80);
  else if (max_n <= 64 * 160)
    // (float64: 320 registers of values per lane — one wave per SIMD and 38 registers in scratch; a rare size, but a
    // segment that is most of a 10 000-point shape must not abort the whole batch's evaluation)
    KNN3_GO(160);
  else {
    pn_set_error("pn_knn3_ragged: segments of up to %d points (max %d)", max_n, 64 * 160);
    return PN_ERR_UNSUPPORTED;
  }
#undef KNN3_GO
  PN_CHECK_LAUNCH();
  return PN_OK;
}

// max_n: an upper bound of the segment sizes (the host knows them: it built the offsets).  f64: distances and
// selection in float64 (open3d's arithmetic), else fp32 (the reference's torch arithmetic).  dist may be null;
// its element type follows f64.
extern "C" int pn_knn3_ragged(const float* pts, const int* off, int S, int max_n, int k, int f64, int* idx,
                              void* dist, void* stream) {
  PN_CHECK_ARG(pts && off && idx, "pn_knn3_ragged: null pointer");
  PN_CHECK_ARG(S > 0 && max_n > 0 && k >= 1 && k <= 64, "pn_knn3_ragged: bad sizes (S=%d max_n=%d k=%d; k <= 64)", S,
               max_n, k);
  PN_PROF("knn3_ragged", (hipStream_t)stream);
  if (f64) return knn3_launch<double>(pts, off, S, max_n, k, idx, (double*)dist, (hipStream_t)stream);
  return knn3_launch<float>(pts, off, S, max_n, k, idx, (float*)dist, (hipStream_t)stream);
}
