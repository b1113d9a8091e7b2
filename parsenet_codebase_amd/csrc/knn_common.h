// Selection primitives shared by the kNN kernels (knn.hip: generic scan path,
// knn_mfma.hip: two-pass MFMA path).
#pragma once
#include "common.h"

#define KNN_TC 32        // candidates per step (accumulators per lane)
#define KNN_CAP 1024     // list capacity per (query, slice) in keys
#define KNN_CAP0 256     // first compaction after this many keys (tightens tau early)
#define KNN_MAXK 128
#define KNN_EPL (KNN_CAP / 64)  // list entries per lane during a wave-wide select

typedef unsigned long long u64;

// Where a graph's indices go: int64 at the API (torch's index dtype; src/model.py:19 returns topk's indices),
// int32 inside the library (the edge-conv kernels read half the bytes).  p == null: no index output.
struct KnnIdxOut {
  void* p;
  int is32;
  __host__ __device__ explicit operator bool() const { return p != nullptr; }
  __device__ inline void put(size_t pos, int64_t v) const {
    if (is32)
      static_cast<int*>(p)[pos] = (int)v;
    else
      static_cast<int64_t*>(p)[pos] = v;
  }
};

// larger key = better neighbour: larger value first, then smaller index.  -0.0 is folded
// into +0.0 so that key order agrees with the float comparison the oracle uses.
__device__ static inline u64 knn_key(float v, int j) {
  if (v == 0.0f) v = 0.0f;
  return ((u64)pn_f2ord(v) << 32) | (u64)(0xffffffffu - (uint32_t)j);
}
__device__ static inline int64_t knn_key_index(u64 key) {
  return key ? (int64_t)(0xffffffffu - (uint32_t)(key & 0xffffffffu)) : 0;
}

__device__ static inline u64 readlane_u64(u64 v, int l) {
  uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, l);
  uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(v >> 32), l);
  return ((u64)hi << 32) | lo;
}

// Wave-cooperative: among the n keys at lp[0..n) find the k-th largest key and keep (compacted
// to lp[0..m), unordered) every key whose VALUE is at least value(k-th) - slack; slack = 0 keeps
// exactly the k largest.  Returns the k-th largest key, m through ``kept``.
// Requires k <= n <= KNN_CAP.
__device__ static u64 knn_wave_select_ge(u64* __restrict__ lp, int n, int k, uint32_t* __restrict__ hist,
                                         float slack, int* __restrict__ kept) {
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
  // compact: exactly k keys are >= kth because keys are pairwise distinct (slack = 0)
  const u64 kth_found = kth;
  if (slack > 0.f) kth = (u64)pn_f2ord(pn_ord2f((uint32_t)(kth >> 32)) - slack) << 32;
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
  *kept = __builtin_amdgcn_readlane(inc, 63);
#pragma unroll
  for (int e = 0; e < KNN_EPL; ++e)
    if (e * 64 + lane < n && key[e] >= kth) lp[off++] = key[e];
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
  return kth_found;
}

__device__ static inline u64 knn_wave_select(u64* __restrict__ lp, int n, int k, uint32_t* __restrict__ hist) {
  int kept;
  return knn_wave_select_ge(lp, n, k, hist, 0.f, &kept);
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

