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
//   * Q query points per lane (coordinates in VGPRs), candidate points staged in LDS as packed
//     float4 tiles: ONE broadcast ds_read_b128 serves Q x 64 pair evaluations, which takes the
//     kernel off the LDS issue port (the one-query-per-lane version issued three ds_read_b32
//     per 11 VALU instructions);
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

#define CH_THREADS 256
#define CH_TILE 512   // candidate points per LDS tile (8 KiB of float4)
#define CH_GROUP 8

template <int Q>
__global__ __launch_bounds__(CH_THREADS) void pn_chamfer_nn_kernel(
    const float* __restrict__ q, const int* __restrict__ qoff, int Nq_uniform,
    const float* __restrict__ c, const int* __restrict__ coff, int Nc_uniform, int chunk, int direct,
    unsigned long long* __restrict__ packed, float* __restrict__ mind, int64_t* __restrict__ arg) {
  __shared__ float4 tile[CH_TILE];
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
  for (int j0 = j_begin; j0 < j_end; j0 += CH_TILE) {
    const int n = min(CH_TILE, j_end - j0);
    const int npad = (n + CH_GROUP - 1) / CH_GROUP * CH_GROUP;
    __syncthreads();
    // coalesced stage: 3n consecutive floats into the xyz lanes of the float4 tile; the tail of
    // the last group is padded with +inf (distance inf is never strictly smaller)
    float* tf = (float*)tile;
    for (int t = threadIdx.x; t < 3 * n; t += CH_THREADS) {
      const float v = cb[(size_t)3 * j0 + t];
      const int p = t / 3, k = t - 3 * p;
      tf[4 * p + k] = v;
    }
    for (int p = n + threadIdx.x; p < npad; p += CH_THREADS)
      tile[p] = make_float4(__builtin_inff(), __builtin_inff(), __builtin_inff(), 0.f);
    __syncthreads();
    for (int g0 = 0; g0 < npad; g0 += CH_GROUP) {
      float m[Q];
#pragma unroll
      for (int r = 0; r < Q; ++r) m[r] = __builtin_inff();
#pragma unroll
      for (int p = 0; p < CH_GROUP; ++p) {
        const float4 cc = tile[g0 + p];
#pragma unroll
        for (int r = 0; r < Q; ++r) {
          const float dx = __fsub_rn(qx[r], cc.x);
          const float dy = __fsub_rn(qy[r], cc.y);
          const float dz = __fsub_rn(qz[r], cc.z);
          const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
          m[r] = fminf(m[r], d);
        }
      }
#pragma unroll
      for (int r = 0; r < Q; ++r)
        if (m[r] < best[r]) {   // strict: the first group wins
          best[r] = m[r];
          bestg[r] = j0 + g0;
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
    for (int j = bestg[r]; j < gend; ++j) {
      const float dx = __fsub_rn(qx[r], cb[3 * (size_t)j + 0]);
      const float dy = __fsub_rn(qy[r], cb[3 * (size_t)j + 1]);
      const float dz = __fsub_rn(qz[r], cb[3 * (size_t)j + 2]);
      const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
      if (d == best[r]) {
        besti = j;
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
  // Q queries per lane: more VALU work per LDS read, fewer workgroups — only when the batch
  // still fills the chip (PN_CHAMFER_Q overrides, for tuning)
  const long long lanes = (long long)B * Nq;
  int Q = lanes >= 32768 ? 4 : (lanes >= 8192 ? 2 : 1);
  if (const char* e = getenv("PN_CHAMFER_Q")) {
    const int v = atoi(e);
    if (v == 1 || v == 2 || v == 4) Q = v;
  }
  const int qblocks = pn_cdiv(Nq, CH_THREADS * Q);
  // enough blocks to cover 256 CUs several times over, but never split a
  // candidate range below one LDS tile
  int splits = pn_cdiv(2048, (long long)qblocks * B);
  const int max_splits = pn_cdiv(Nc, CH_TILE);
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
