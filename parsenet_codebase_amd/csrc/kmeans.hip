// Spherical k-means steps for the locality order of the block-sparse mean-shift iterations
// (mean_shift.locality_order; no counterpart in the reference: the order is free — mean-shift is
// permutation-equivariant, src/mean_shift.py:45-79 — it only decides how tight the 32-point tiles are).
//
// Until round 5 a Lloyd step was a dozen tensor-library launches (a GEMM and an arg-max for the assignment;
// one-hot, a GEMM, norm, clamp, divide, where for the centres): 66 launches and ~0.6 ms per end-to-end step
// for the two k-means of a call.  Two kernels here:
//   pn_kmeans_assign_f32   label of every point = arg-max over the centres of the dot product
//   pn_kmeans_centres_f32  centre of every cell = normalised sum of its points
// Both are deterministic (fixed summation orders, ties -> the smaller centre index): the order they produce
// — and with it the summation order of every later launch — is the same from run to run.
#include "common.h"

typedef float km_f32x16 __attribute__((ext_vector_type(16)));
#define KM_D 128
#define KM_CH 96           // centres per LDS chunk (three tiles of 32: 50 KiB)
#define KM_LD 131          // floats per staged centre row (the two k halves of a row 65 floats apart)

// x (B,N,128), cen (B,K,128) -> lab (B,N) int32.  grid (ceil(N / 128), B), 4 waves, 32 points each.
// D[i = centre][j = point] on v_mfma_f32_32x32x2_f32: k-step m = channels (m, m + 64), a lane reads 64
// contiguous floats of its point once (the resident operand) and one float of a staged centre per k-step.
__global__ __launch_bounds__(256) void pn_kmeans_assign_kernel(const float* __restrict__ x, const float* __restrict__ cen,
                                                               int N, int K, int* __restrict__ lab) {
  __shared__ float cs[KM_CH * KM_LD];
  const int b = blockIdx.y, tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, col = lane & 31, h = lane >> 5;
  const int p = (blockIdx.x * 4 + wave) * 32 + col;
  const int pc = p < N ? p : N - 1;
  const float4* xr = reinterpret_cast<const float4*>(x + ((size_t)b * N + pc) * KM_D + 64 * h);
  float bq[64];
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const float4 v = xr[e];
    bq[4 * e] = v.x;
    bq[4 * e + 1] = v.y;
    bq[4 * e + 2] = v.z;
    bq[4 * e + 3] = v.w;
  }
  float best = -__builtin_inff();
  int bi = 0;
  const float* cb = cen + (size_t)b * K * KM_D;
  for (int c0 = 0; c0 < K; c0 += KM_CH) {
    const int nc = min(KM_CH, K - c0);
    __syncthreads();
    // stage nc centres: thread t copies float t & 127 of rows t >> 7, + 2, ...
    for (int e = tid; e < nc * KM_D; e += 256) {
      const int r = e >> 7, ch = e & 127;
      cs[r * KM_LD + (ch & 63) + 65 * (ch >> 6)] = cb[(size_t)(c0 + r) * KM_D + ch];
    }
    __syncthreads();
    for (int t0 = 0; t0 < nc; t0 += 32) {
      const int r = min(t0 + col, nc - 1);
      const float* ar = cs + r * KM_LD + 65 * h;
      km_f32x16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
      for (int m = 0; m < 64; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[m], bq[m], acc, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int ci = t0 + (i & 3) + 8 * (i >> 2) + 4 * h;      // ascending in i: strict > keeps the smaller index
        if (ci < nc && acc[i] > best) {
          best = acc[i];
          bi = c0 + ci;
        }
      }
    }
  }
  // the two lanes of a point
  const float ob = __shfl_xor(best, 32, 64);
  const int oi = __shfl_xor(bi, 32, 64);
  if (ob > best || (ob == best && oi < bi)) bi = oi;
  if (h == 0 && p < N) lab[(size_t)b * N + p] = bi;
}

// x (B,N,128), lab (B,N), old (B,K,128) -> cen (B,K,128): the normalised sum of the cell's points, or the old
// centre when the cell is empty (norm <= 1e-6).  grid (K, B), 4 waves: wave w adds the members among points
// [w N/4, (w+1) N/4) in index order (lane = channels lane, lane + 64), the four partial sums are added in wave order.
__global__ __launch_bounds__(256) void pn_kmeans_centres_kernel(const float* __restrict__ x, const int* __restrict__ lab,
                                                                const float* __restrict__ old, int N, int K,
                                                                float* __restrict__ cen) {
  __shared__ float part[4][KM_D];
  const int b = blockIdx.y, c = blockIdx.x, tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int per = ((N + 3) / 4 + 63) / 64 * 64;
  const int j0 = wave * per, j1 = min(N, j0 + per);
  const int* lb = lab + (size_t)b * N;
  const float* xb = x + (size_t)b * N * KM_D;
  float s0 = 0.f, s1 = 0.f;
  for (int base = j0; base < j1; base += 256) {
    // four chunks of labels in flight
    int l[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = base + 64 * u + lane;
      l[u] = j < j1 ? lb[j] : -1;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      unsigned long long m = __ballot(l[u] == c);
      while (m) {
        const int j = base + 64 * u + __ffsll((long long)m) - 1;
        m &= m - 1ull;
        s0 += xb[(size_t)j * KM_D + lane];
        s1 += xb[(size_t)j * KM_D + lane + 64];
      }
    }
  }
  part[wave][lane] = s0;
  part[wave][lane + 64] = s1;
  __syncthreads();
  if (wave == 0) {
    const float t0 = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
    const float t1 = ((part[0][lane + 64] + part[1][lane + 64]) + part[2][lane + 64]) + part[3][lane + 64];
    float nn = t0 * t0 + t1 * t1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nn += __shfl_xor(nn, o, 64);
    nn = sqrtf(nn);
    const size_t o = ((size_t)b * K + c) * KM_D;
    const bool ok = nn > 1e-6f;
    cen[o + lane] = ok ? t0 / nn : old[o + lane];
    cen[o + lane + 64] = ok ? t1 / nn : old[o + lane + 64];
  }
}

extern "C" int pn_kmeans_assign_f32(const float* x, const float* cen, int B, int N, int D, int K, int* lab,
                                    void* stream) {
  PN_CHECK_ARG(x && cen && lab && B > 0 && N > 0 && K > 0, "pn_kmeans_assign_f32: bad arguments");
  PN_CHECK_ARG(D == KM_D, "pn_kmeans_assign_f32: embedding size %d unsupported (built for %d)", D, KM_D);
  PN_PROF("kmeans_assign", (hipStream_t)stream);
  hipLaunchKernelGGL(pn_kmeans_assign_kernel, dim3(pn_cdiv(N, 128), B), dim3(256), 0, (hipStream_t)stream, x, cen, N, K,
                     lab);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

extern "C" int pn_kmeans_centres_f32(const float* x, const int* lab, const float* old, int B, int N, int D, int K,
                                     float* cen, void* stream) {
  PN_CHECK_ARG(x && lab && old && cen && B > 0 && N > 0 && K > 0, "pn_kmeans_centres_f32: bad arguments");
  PN_CHECK_ARG(D == KM_D, "pn_kmeans_centres_f32: embedding size %d unsupported (built for %d)", D, KM_D);
  PN_PROF("kmeans_centres", (hipStream_t)stream);
  hipLaunchKernelGGL(pn_kmeans_centres_kernel, dim3(K, B), dim3(256), 0, (hipStream_t)stream, x, lab, old, N, K, cen);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

// ---- the order itself: a stable counting sort by cell ---------------------------------------------------------
// Points are filed by key = rank[home[fine]] * F + fine (mean_shift.locality_order): every fine cell has ONE key,
// so the stable argsort of the keys is a counting sort over the F cells taken in key order.  One workgroup of
// 16 waves per cloud: cell positions (F^2 / 1024 comparisons a thread), per-wave histograms of contiguous
// element ranges, a scan over (cell position, wave), then every wave files its range in index order — inside a
// group of 64 elements a lane's slot is the number of equal cells on lower lanes (ballots), so the result is THE
// stable permutation, identical to the tensor library's argsort(stable=True) (which took 16 launches).
#define KO_WAVES 16
#define KO_MAXF 768

__global__ __launch_bounds__(KO_WAVES * 64) void pn_cell_order_kernel(const int* __restrict__ rank, const int* __restrict__ home,
                                                                      const int* __restrict__ fine, int N, int P, int F,
                                                                      long long* __restrict__ perm) {
  extern __shared__ int ko_lds[];
  int* key = ko_lds;                    // [F] key of a cell
  int* pos = key + F;                   // [F] position of a cell in key order
  int* at = pos + F;                    // [F] cell at a position
  int* start = at + F;                  // [F] first slot of the cell at a position (exclusive scan of the counts)
  volatile int* hist = start + F;       // [KO_WAVES][F] count, then next free slot, of (wave, cell)
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int* rk = rank + (size_t)b * P;
  const int* hm = home + (size_t)b * F;
  const int* fn = fine + (size_t)b * N;
  long long* out = perm + (size_t)b * N;
  for (int f = tid; f < F; f += blockDim.x) key[f] = rk[hm[f]] * F + f;
  for (int i = tid; i < KO_WAVES * F; i += blockDim.x) hist[i] = 0;
  __syncthreads();
  for (int f = tid; f < F; f += blockDim.x) {
    const int kf = key[f];
    int p = 0;
    for (int g = 0; g < F; ++g) p += key[g] < kf;
    pos[f] = p;
    at[p] = f;
  }
  const int per = (N + KO_WAVES - 1) / KO_WAVES;
  const int lo = w * per, hi = min(N, lo + per);
  for (int i = lo + lane; i < hi; i += 64) atomicAdd((int*)&hist[w * F + fn[i]], 1);
  __syncthreads();
  // slots of a cell: its waves in order; cells in key order.  One thread per position sums its waves, a
  // single wave scans the F totals (F <= 768: twelve values a lane).
  for (int p = tid; p < F; p += blockDim.x) {
    const int f = at[p];
    int s = 0;
    for (int v = 0; v < KO_WAVES; ++v) s += hist[v * F + f];
    start[p] = s;
  }
  __syncthreads();
  if (w == 0) {
    const int each = (F + 63) / 64;
    int s = 0;
    for (int j = 0; j < each; ++j) { const int p = lane * each + j; if (p < F) s += start[p]; }
    int incl = s;
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
    int run = incl - s;
    for (int j = 0; j < each; ++j) {
      const int p = lane * each + j;
      if (p < F) { const int c = start[p]; start[p] = run; run += c; }
    }
  }
  __syncthreads();
  for (int p = tid; p < F; p += blockDim.x) {
    const int f = at[p];
    int run = start[p];
    for (int v = 0; v < KO_WAVES; ++v) { const int c = hist[v * F + f]; hist[v * F + f] = run; run += c; }
  }
  __syncthreads();
  volatile int* mine = hist + w * F;
  for (int i0 = lo; i0 < hi; i0 += 64) {
    const int i = i0 + lane;
    const bool valid = i < hi;
    const int f = valid ? fn[i] : -1;
    unsigned long long left = __ballot(valid);
    int dst = 0;
    while (left) {
      const int leader = __ffsll((long long)left) - 1;
      const int lf = __shfl(f, leader);
      const unsigned long long m = __ballot(valid && f == lf);
      if (valid && f == lf) dst = mine[lf] + __popcll(m & ((1ull << lane) - 1ull));
      __builtin_amdgcn_wave_barrier();
      if (lane == leader) mine[lf] = mine[lf] + __popcll(m);
      __builtin_amdgcn_wave_barrier();
      left &= ~m;
    }
    if (valid) out[dst] = i;
  }
}

extern "C" int pn_cell_order_i32(const int* rank, const int* home, const int* fine, int B, int N, int P, int F,
                                 long long* perm, void* stream) {
  PN_CHECK_ARG(rank && home && fine && perm && B > 0 && N > 0 && P > 0, "pn_cell_order_i32: bad arguments");
  PN_CHECK_ARG(F > 0 && F <= KO_MAXF, "pn_cell_order_i32: %d cells unsupported (at most %d)", F, KO_MAXF);
  PN_CHECK_ARG((long long)P * F < (1ll << 31), "pn_cell_order_i32: keys of %d x %d cells do not fit 32 bits", P, F);
  PN_PROF("cell_order", (hipStream_t)stream);
  const size_t lds = (size_t)(4 + KO_WAVES) * F * sizeof(int);
  hipLaunchKernelGGL(pn_cell_order_kernel, dim3(B), dim3(KO_WAVES * 64), lds, (hipStream_t)stream, rank, home, fine, N, P, F,
                     perm);
  PN_CHECK_LAUNCH();
  return PN_OK;
}
