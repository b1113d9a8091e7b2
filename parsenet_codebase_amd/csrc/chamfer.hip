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
