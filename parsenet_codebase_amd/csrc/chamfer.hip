// Chamfer nearest-neighbour search for gfx950.
//
// Replaces the (M,N,3) broadcast + min of the reference
//   src/utils.py:286-296 (chamfer_distance), :313-323 (one_side),
//   :338-358 (single_shape)
// with a tiled search that never materialises the M x N distance matrix.
//
// Bound: fp32 VALU issue.  A pair costs 8 arithmetic instructions that cannot be fused
// (d = ((dx*dx + dy*dy) + dz*dz), every operation rounded to fp32 exactly like the reference's
// elementwise path, so minima and arg-mins are bit-identical to the oracle) plus the selection.
//   * Q query points per lane (coordinates in VGPRs); the candidates of a group are the same for
//     every lane, so they are fetched with SCALAR loads (wave-uniform address -> s_load_dwordx8
//     through the scalar cache) and enter the VALU instructions as SGPR operands: no LDS tile, no
//     staging pass, no barrier — the one-query-per-lane LDS version issued three ds_read_b32
//     per 11 VALU instructions and was bound by the LDS port;
//     (round 5: the same eight operations for two queries of a lane on v_pk_add_f32 / v_pk_mul_f32 — 4.5 instead
//     of 8.5 issue slots per pair, bit-identical results — measured NO faster: 0.065 against 0.061 ms at 10k x 10k
//     with two queries per lane, 0.071 with four (tools/jobs/r5m.sh); the waves wait for the scalar loads of the
//     next group (SQ_WAIT_ANY 42 %), not for issue slots; not kept)
//   * selection per GROUP of 8 candidates: v_min3 chain + one compare/select pair per group
//     instead of per candidate (8.9 instead of 11 VALU instructions per pair); the arg-min is
//     the first candidate of the winning group that reproduces the minimum, resolved once per
//     query at the end — the same "strictly smaller wins, first index on ties" rule;
//   * candidate range split over blockIdx.y so that small clouds still fill the chip, merged
//     with a 64-bit packed (distance, index) atomicMin; a single split writes its result directly;
//   * ragged batches: item i owns rows [off[i], off[i+1]) of a concatenated cloud (the spline
//     segments of a step have different numbers of ground-truth points).
#include "common.h"
#include <stdlib.h>

#define CH_THREADS 128
#define CH_GROUP 8

__device__ static inline float ch_dist(float qx, float qy, float qz, float cx, float cy, float cz) {
  const float dx = __fsub_rn(qx, cx);
  const float dy = __fsub_rn(qy, cy);
  const float dz = __fsub_rn(qz, cz);
  return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

template <int Q>
__global__ __launch_bounds__(CH_THREADS) void pn_chamfer_nn_kernel(
    const float* __restrict__ q, const int* __restrict__ qoff, int Nq_uniform,
    const float* __restrict__ c, const int* __restrict__ coff, int Nc_uniform, int chunk, int direct,
    unsigned long long* __restrict__ packed, float* __restrict__ mind, int64_t* __restrict__ arg) {
  const int b = blockIdx.z;
  const int q0 = qoff ? qoff[b] : b * Nq_uniform;
  const int Nq = qoff ? qoff[b + 1] - q0 : Nq_uniform;
  const int c0 = coff ? coff[b] : b * Nc_uniform;
  const int Nc = coff ? coff[b + 1] - c0 : Nc_uniform;
  const int qbase = blockIdx.x * (CH_THREADS * Q);
  if (qbase >= Nq) return;   // block-uniform (ragged items are shorter than the grid)
  const float* qb = q + (size_t)q0 * 3;
  const float* cb = c + (size_t)c0 * 3;
  float qx[Q], qy[Q], qz[Q], best[Q];
  int bestg[Q];
#pragma unroll
  for (int r = 0; r < Q; ++r) {
    const int i = qbase + r * CH_THREADS + threadIdx.x;
    qx[r] = qy[r] = qz[r] = 0.f;
    if (i < Nq) {
      qx[r] = qb[3 * (size_t)i + 0];
      qy[r] = qb[3 * (size_t)i + 1];
      qz[r] = qb[3 * (size_t)i + 2];
    }
    best[r] = __builtin_inff();
    bestg[r] = 0x7fffffff;
  }
  const int j_begin = blockIdx.y * chunk;
  const int j_end = min(Nc, j_begin + chunk);
  int j = j_begin;
  for (; j + CH_GROUP <= j_end; j += CH_GROUP) {
    // (a ping-pong of two scalar register sets — the loads of group g + 1 issued before the arithmetic
    // of group g — was measured in round 3: 0.083 instead of 0.061 ms at 10k x 10k; scalar loads return
    // out of order, so every wait is for ALL outstanding loads and the prefetch only lengthens it)
    const float* cj = cb + 3 * (size_t)j;   // wave-uniform: scalar loads
    float cc[3 * CH_GROUP];
#pragma unroll
    for (int t = 0; t < 3 * CH_GROUP; ++t) cc[t] = cj[t];
#pragma unroll
    for (int r = 0; r < Q; ++r) {
      float m = ch_dist(qx[r], qy[r], qz[r], cc[0], cc[1], cc[2]);
#pragma unroll
      for (int p = 1; p < CH_GROUP; ++p)
        m = fminf(m, ch_dist(qx[r], qy[r], qz[r], cc[3 * p], cc[3 * p + 1], cc[3 * p + 2]));
      if (m < best[r]) {   // strict: the first group wins
        best[r] = m;
        bestg[r] = j;
      }
    }
  }
  if (j < j_end) {   // last, partial group
#pragma unroll
    for (int r = 0; r < Q; ++r) {
      float m = __builtin_inff();
      for (int p = j; p < j_end; ++p)
        m = fminf(m, ch_dist(qx[r], qy[r], qz[r], cb[3 * (size_t)p], cb[3 * (size_t)p + 1], cb[3 * (size_t)p + 2]));
      if (m < best[r]) {
        best[r] = m;
        bestg[r] = j;
      }
    }
  }
  // resolve the arg-min inside the winning group: first candidate that reproduces the minimum
#pragma unroll
  for (int r = 0; r < Q; ++r) {
    const int i = qbase + r * CH_THREADS + threadIdx.x;
    if (i >= Nq) continue;
    if (bestg[r] == 0x7fffffff) {   // no finite distance (inf / NaN coordinates): the "empty" key
      if (direct) {
        if (mind) mind[(size_t)q0 + i] = __uint_as_float(0xffffffffu);
        if (arg) arg[(size_t)q0 + i] = (int64_t)0xffffffffu;
      }
      continue;
    }
    int besti = bestg[r];
    const int gend = min(j_end, bestg[r] + CH_GROUP);
    for (int p = bestg[r]; p < gend; ++p) {
      const float d = ch_dist(qx[r], qy[r], qz[r], cb[3 * (size_t)p + 0], cb[3 * (size_t)p + 1], cb[3 * (size_t)p + 2]);
      if (d == best[r]) {
        besti = p;
        break;
      }
    }
    if (direct) {
      if (mind) mind[(size_t)q0 + i] = best[r];
      if (arg) arg[(size_t)q0 + i] = besti;
    } else {
      // d >= 0 so the raw bit pattern is already order preserving
      const unsigned long long key = ((unsigned long long)__float_as_uint(best[r]) << 32) | (unsigned)besti;
      atomicMin(&packed[(size_t)q0 + i], key);
    }
  }
}

__global__ void pn_chamfer_unpack_kernel(const unsigned long long* __restrict__ packed,
                                         long long n, float* __restrict__ mind,
                                         int64_t* __restrict__ arg) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned long long k = packed[i];
  if (mind) mind[i] = __uint_as_float((uint32_t)(k >> 32));
  if (arg) arg[i] = (int64_t)(uint32_t)(k & 0xffffffffu);
}

// one direction: queries q (rows qoff / uniform Nq) against candidates c
static int chamfer_one_side(const float* q, const int* qoff, int Nq, long long total_q, const float* c,
                            const int* coff, int Nc, int B, unsigned long long* packed, float* mind,
                            int64_t* arg, hipStream_t stream) {
  // Q queries per lane amortise the scalar candidate loads; keep >= ~2000 waves in flight
  // (PN_CHAMFER_Q overrides, for tuning)
  const long long lanes = (long long)B * Nq;
  int Q = lanes >= 262144 ? 4 : (lanes >= 65536 ? 2 : 1);
  if (const char* e = getenv("PN_CHAMFER_Q")) {
    const int v = atoi(e);
    if (v == 1 || v == 2 || v == 4) Q = v;
  }
  const int qblocks = pn_cdiv(Nq, CH_THREADS * Q);
  // enough workgroups to cover 256 CUs several times over, but never split a candidate range
  // below 256 points
  int splits = pn_cdiv(4096, (long long)qblocks * B);
  const int max_splits = pn_cdiv(Nc, 256);
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  int chunk = pn_cdiv(Nc, splits);
  chunk = (int)pn_align_up(chunk, CH_GROUP * 8);
  splits = pn_cdiv(Nc, chunk);
  const int direct = splits == 1;
  if (!direct) PN_CHECK_HIP(hipMemsetAsync(packed, 0xff, total_q * sizeof(unsigned long long), stream));
  dim3 grid(qblocks, splits, B);
  {
    PN_PROF("chamfer_nn", stream);
    if (Q == 4)
      hipLaunchKernelGGL(pn_chamfer_nn_kernel<4>, grid, dim3(CH_THREADS), 0, stream, q, qoff, Nq, c, coff, Nc,
                         chunk, direct, packed, mind, arg);
    else if (Q == 2)
      hipLaunchKernelGGL(pn_chamfer_nn_kernel<2>, grid, dim3(CH_THREADS), 0, stream, q, qoff, Nq, c, coff, Nc,
                         chunk, direct, packed, mind, arg);
    else
      hipLaunchKernelGGL(pn_chamfer_nn_kernel<1>, grid, dim3(CH_THREADS), 0, stream, q, qoff, Nq, c, coff, Nc,
                         chunk, direct, packed, mind, arg);
  }
  PN_CHECK_LAUNCH();
  if (!direct) {
    hipLaunchKernelGGL(pn_chamfer_unpack_kernel, dim3(pn_cdiv(total_q, 256)), dim3(256), 0, stream, packed,
                       total_q, mind, arg);
    PN_CHECK_LAUNCH();
  }
  return PN_OK;
}

extern "C" size_t pn_chamfer_nn_workspace(int B, int Na, int Nb) {
  return pn_align_up((size_t)B * Na * 8, 256) + pn_align_up((size_t)B * Nb * 8, 256);
}

extern "C" int pn_chamfer_nn_f32(const float* a, const float* b, int B, int Na, int Nb,
                                 float* minA, int64_t* argA, float* minB, int64_t* argB,
                                 void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PN_CHECK_ARG(a && b, "pn_chamfer_nn_f32: null input");
  PN_CHECK_ARG(B > 0 && Na > 0 && Nb > 0, "pn_chamfer_nn_f32: empty cloud (B=%d Na=%d Nb=%d)",
               B, Na, Nb);
  PN_CHECK_ARG(workspace && workspace_bytes >= pn_chamfer_nn_workspace(B, Na, Nb),
               "pn_chamfer_nn_f32: workspace too small");
  unsigned long long* pa = (unsigned long long*)workspace;
  unsigned long long* pb =
      (unsigned long long*)((char*)workspace + pn_align_up((size_t)B * Na * 8, 256));
  int rc = PN_OK;
  if (minA || argA) {
    rc = chamfer_one_side(a, nullptr, Na, (long long)B * Na, b, nullptr, Nb, B, pa, minA, argA, stream);
    if (rc) return rc;
  }
  if (minB || argB) {
    rc = chamfer_one_side(b, nullptr, Nb, (long long)B * Nb, a, nullptr, Na, B, pb, minB, argB, stream);
    if (rc) return rc;
  }
  return PN_OK;
}

// Ragged batch: item i is a[offA[i] .. offA[i+1]) against b[offB[i] .. offB[i+1]); offsets are
// device arrays of B+1 ints, maxA / maxB the largest item sizes (grid sizing), totalA / totalB the
// row counts of the concatenated clouds.  Outputs are concatenated like the inputs; indices are
// local to the item.  Items must be non-empty.
extern "C" size_t pn_chamfer_nn_ragged_workspace(int totalA, int totalB) {
  return pn_align_up((size_t)totalA * 8, 256) + pn_align_up((size_t)totalB * 8, 256);
}

extern "C" int pn_chamfer_nn_ragged_f32(const float* a, const int* offA, int totalA, int maxA, const float* b,
                                        const int* offB, int totalB, int maxB, int B, float* minA,
                                        int64_t* argA, float* minB, int64_t* argB, void* workspace,
                                        size_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PN_CHECK_ARG(a && b && offA && offB, "pn_chamfer_nn_ragged_f32: null input");
  PN_CHECK_ARG(B > 0 && maxA > 0 && maxB > 0 && totalA > 0 && totalB > 0, "pn_chamfer_nn_ragged_f32: empty batch");
  PN_CHECK_ARG(workspace && workspace_bytes >= pn_chamfer_nn_ragged_workspace(totalA, totalB),
               "pn_chamfer_nn_ragged_f32: workspace too small");
  unsigned long long* pa = (unsigned long long*)workspace;
  unsigned long long* pb = (unsigned long long*)((char*)workspace + pn_align_up((size_t)totalA * 8, 256));
  int rc = PN_OK;
  if (minA || argA) {
    rc = chamfer_one_side(a, offA, maxA, totalA, b, offB, maxB, B, pa, minA, argA, stream);
    if (rc) return rc;
  }
  if (minB || argB) {
    rc = chamfer_one_side(b, offB, maxB, totalB, a, offA, maxA, B, pb, minB, argB, stream);
    if (rc) return rc;
  }
  return PN_OK;
}

// ---- reduced two-sided distance and its backward for the ragged batch (round 4) ---------------
// chamfer_distance_single_shape (src/utils.py:326-358, two-sided, squared, reduce=True) of every
// item: out[s] = (mean_i minA + mean_j minB) / 2.  One workgroup per item, strided partial sums
// combined by a fixed tree: the same bits on every run (torch's index_add_ uses fp32 atomics).
__global__ __launch_bounds__(256) void pn_chamfer_ragged_reduce_kernel(const float* __restrict__ minA,
                                                                       const int* __restrict__ offA,
                                                                       const float* __restrict__ minB,
                                                                       const int* __restrict__ offB,
                                                                       float* __restrict__ out) {
  __shared__ float red[2][256];
  const int s = blockIdx.x, t = threadIdx.x;
  const int a0 = offA[s], a1 = offA[s + 1], b0 = offB[s], b1 = offB[s + 1];
  float sa = 0.f, sb = 0.f;
  for (int i = a0 + t; i < a1; i += 256) sa += minA[i];
  for (int j = b0 + t; j < b1; j += 256) sb += minB[j];
  red[0][t] = sa;
  red[1][t] = sb;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) {
      red[0][t] += red[0][t + o];
      red[1][t] += red[1][t + o];
    }
    __syncthreads();
  }
  if (t == 0) out[s] = (red[0][0] / (float)(a1 - a0) + red[1][0] / (float)(b1 - b0)) / 2.0f;
}

extern "C" int pn_chamfer_ragged_reduce_f32(const float* minA, const int* offA, const float* minB, const int* offB,
                                            int S, float* out, void* stream_) {
  PN_CHECK_ARG(minA && offA && minB && offB && out && S > 0, "pn_chamfer_ragged_reduce_f32: bad arguments");
  hipLaunchKernelGGL(pn_chamfer_ragged_reduce_kernel, dim3(S), dim3(256), 0, (hipStream_t)stream_, minA, offA, minB,
                     offB, out);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

// d out / d pred: every minimum routed to its pair of points,
//   gpred[i] = (pred_i - gt[argA_i]) * g_s / nA  +  sum_{j : argB_j = i} (pred_i - gt_j) * g_s / nB
// (the 2 of the square cancels the 1/2).  One wave owns 64 prediction rows of an item and walks the
// item's targets 64 at a time; the few whose nearest prediction is one of its rows are applied in
// ascending j — a gather, no atomics.
__global__ __launch_bounds__(256) void pn_chamfer_ragged_bwd_kernel(
    const float* __restrict__ pred, const int* __restrict__ offA, const float* __restrict__ gt,
    const int* __restrict__ offB, const int64_t* __restrict__ argA, const int64_t* __restrict__ argB,
    const float* __restrict__ g, float* __restrict__ gpred) {
  const int s = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int a0 = offA[s], nA = offA[s + 1] - a0, b0 = offB[s], nB = offB[s + 1] - b0;
  const int i0 = (blockIdx.x * 4 + wave) * 64;
  if (i0 >= nA) return;
  const int il = i0 + lane;
  const bool on = il < nA;
  const float gs = g[s];
  const float ga = gs / (float)nA, gb = gs / (float)nB;
  float px = 0.f, py = 0.f, pz = 0.f, ax = 0.f, ay = 0.f, az = 0.f;
  if (on) {
    const float* p = pred + (size_t)(a0 + il) * 3;
    px = p[0], py = p[1], pz = p[2];
    const float* q = gt + (size_t)(b0 + (int)argA[a0 + il]) * 3;
    ax = (px - q[0]) * ga;
    ay = (py - q[1]) * ga;
    az = (pz - q[2]) * ga;
  }
  // four chunks of 64 targets per trip: their index loads (and the coordinates of the hits) are in flight
  // together — one chunk per trip waited a memory latency per 64 targets (116 us for the step's spline segments)
  for (int j0 = 0; j0 < nB; j0 += 256) {
    int rel[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = j0 + 64 * u + lane;
      rel[u] = j < nB ? (int)argB[b0 + j] - i0 : -1;
    }
    float qx[4], qy[4], qz[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      qx[u] = qy[u] = qz[u] = 0.f;
      if (rel[u] >= 0 && rel[u] < 64) {
        const float* q = gt + (size_t)(b0 + j0 + 64 * u + lane) * 3;
        qx[u] = q[0], qy[u] = q[1], qz[u] = q[2];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      unsigned long long m = __ballot(rel[u] >= 0 && rel[u] < 64);
      while (m) {
        const int src = __builtin_ctzll(m);
        m &= m - 1;
        const int r = __shfl(rel[u], src, 64);
        const float x = __shfl(qx[u], src, 64), y = __shfl(qy[u], src, 64), z = __shfl(qz[u], src, 64);
        if (lane == r) {
          ax += (px - x) * gb;
          ay += (py - y) * gb;
          az += (pz - z) * gb;
        }
      }
    }
  }
  if (on) {
    float* o = gpred + (size_t)(a0 + il) * 3;
    o[0] = ax, o[1] = ay, o[2] = az;
  }
}

extern "C" int pn_chamfer_ragged_bwd_f32(const float* pred, const int* offA, int maxA, const float* gt,
                                         const int* offB, const int64_t* argA, const int64_t* argB, const float* g,
                                         int S, float* gpred, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PN_CHECK_ARG(pred && offA && gt && offB && argA && argB && g && gpred, "pn_chamfer_ragged_bwd_f32: null pointer");
  PN_CHECK_ARG(S > 0 && maxA > 0, "pn_chamfer_ragged_bwd_f32: empty batch");
  PN_PROF("chamfer_ragged_bwd", stream);
  hipLaunchKernelGGL(pn_chamfer_ragged_bwd_kernel, dim3(pn_cdiv(maxA, 256), S), dim3(256), 0, stream, pred, offA, gt,
                     offB, argA, argB, g, gpred);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

// ---- gradient of a row gather with repeated indices (round 4) -----------------------------------
// out[b,m,:] = src[b,idx[b,m],:] (rows of 3 floats: the nearest neighbours of a Chamfer distance).
// Backward: gsrc[b,i,:] = sum_{m : idx[b,m] = i} g[b,m,:], summed in ascending m by the wave that
// owns row i — the tensor library's scatter_add_ uses fp32 atomics, and with three or more
// contributions to a row their order changes the bits.
__global__ __launch_bounds__(256) void pn_gather_rows3_bwd_kernel(const float* __restrict__ g,
                                                                  const int64_t* __restrict__ idx, int M, int N,
                                                                  float* __restrict__ gsrc) {
  const int b = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i0 = (blockIdx.x * 4 + wave) * 64;
  if (i0 >= N) return;
  const float* __restrict__ gb = g + (size_t)b * M * 3;
  const int64_t* __restrict__ ib = idx + (size_t)b * M;
  float ax = 0.f, ay = 0.f, az = 0.f;
  for (int m0 = 0; m0 < M; m0 += 64) {
    const int m = m0 + lane;
    const int rel = m < M ? (int)ib[m] - i0 : -1;
    const bool hit = rel >= 0 && rel < 64;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (hit) {
      const float* q = gb + (size_t)m * 3;
      qx = q[0], qy = q[1], qz = q[2];
    }
    unsigned long long mask = __ballot(hit);
    while (mask) {
      const int src = __builtin_ctzll(mask);
      mask &= mask - 1;
      const int r = __shfl(rel, src, 64);
      const float x = __shfl(qx, src, 64), y = __shfl(qy, src, 64), z = __shfl(qz, src, 64);
      if (lane == r) {
        ax += x;
        ay += y;
        az += z;
      }
    }
  }
  if (i0 + lane < N) {
    float* o = gsrc + ((size_t)b * N + i0 + lane) * 3;
    o[0] = ax, o[1] = ay, o[2] = az;
  }
}

extern "C" int pn_gather_rows3_bwd_f32(const float* g, const int64_t* idx, int B, int M, int N, float* gsrc,
                                       void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PN_CHECK_ARG(g && idx && gsrc && B > 0 && M > 0 && N > 0, "pn_gather_rows3_bwd_f32: bad arguments");
  PN_PROF("gather_rows_bwd", stream);
  hipLaunchKernelGGL(pn_gather_rows3_bwd_kernel, dim3(pn_cdiv(N, 256), B), dim3(256), 0, stream, g, idx, M, N, gsrc);
  PN_CHECK_LAUNCH();
  return PN_OK;
}
