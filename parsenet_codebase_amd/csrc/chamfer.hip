// Chamfer nearest-neighbour search for gfx950.
//
// Replaces the (M,N,3) broadcast + min of the reference
//   src/utils.py:286-296 (chamfer_distance), :313-323 (one_side),
//   :338-358 (single_shape)
// with a tiled search that never materialises the M x N distance matrix:
// one query point per lane (coordinates in VGPRs), candidate points staged in
// LDS as SoA tiles so every lane reads the same LDS address (broadcast, no bank
// conflict), candidate range split over blockIdx.y so that small clouds still
// fill the chip, and a 64-bit packed (distance, index) atomicMin to merge the
// splits.  Squared distances are evaluated exactly like the reference's
// elementwise path: d = ((dx*dx + dy*dy) + dz*dz), every operation rounded to
// fp32 (no FMA contraction), so minima are bit-identical to the oracle.
// Ties resolve to the smallest candidate index.
#include "common.h"

#define CH_THREADS 256
#define CH_TILE 1024  // candidate points per LDS tile (12 KiB)

__global__ __launch_bounds__(CH_THREADS) void pn_chamfer_nn_kernel(
    const float* __restrict__ q, int Nq, const float* __restrict__ c, int Nc,
    int chunk, unsigned long long* __restrict__ packed) {
  __shared__ float sx[CH_TILE], sy[CH_TILE], sz[CH_TILE];
  const int b = blockIdx.z;
  const int i = blockIdx.x * CH_THREADS + threadIdx.x;
  const float* qb = q + (size_t)b * Nq * 3;
  const float* cb = c + (size_t)b * Nc * 3;
  float qx = 0.f, qy = 0.f, qz = 0.f;
  if (i < Nq) {
    qx = qb[3 * i + 0];
    qy = qb[3 * i + 1];
    qz = qb[3 * i + 2];
  }
  const int j_begin = blockIdx.y * chunk;
  const int j_end = min(Nc, j_begin + chunk);
  float best = __builtin_inff();
  int besti = 0x7fffffff;
  for (int j0 = j_begin; j0 < j_end; j0 += CH_TILE) {
    const int n = min(CH_TILE, j_end - j0);
    __syncthreads();
    // coalesced stage: 3n consecutive floats, de-interleaved into SoA
    for (int t = threadIdx.x; t < 3 * n; t += CH_THREADS) {
      float v = cb[(size_t)3 * j0 + t];
      int p = t / 3, k = t - 3 * p;
      (k == 0 ? sx : (k == 1 ? sy : sz))[p] = v;
    }
    __syncthreads();
#pragma unroll 8
    for (int p = 0; p < n; ++p) {
      float dx = __fsub_rn(qx, sx[p]);
      float dy = __fsub_rn(qy, sy[p]);
      float dz = __fsub_rn(qz, sz[p]);
      float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)),
                          __fmul_rn(dz, dz));
      if (d < best) {  // strict: first (smallest) index wins inside a split
        best = d;
        besti = j0 + p;
      }
    }
  }
  if (i < Nq && besti != 0x7fffffff) {
    // d >= 0 so the raw bit pattern is already order preserving
    unsigned long long key =
        ((unsigned long long)__float_as_uint(best) << 32) | (unsigned)besti;
    atomicMin(&packed[(size_t)b * Nq + i], key);
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

static int chamfer_one_side(const float* q, int Nq, const float* c, int Nc, int B,
                            unsigned long long* packed, float* mind, int64_t* arg,
                            hipStream_t stream) {
  const long long n = (long long)B * Nq;
  PN_CHECK_HIP(hipMemsetAsync(packed, 0xff, n * sizeof(unsigned long long), stream));
  const int qblocks = pn_cdiv(Nq, CH_THREADS);
  // enough blocks to cover 256 CUs several times over, but never split a
  // candidate range below one LDS tile
  int splits = pn_cdiv(2048, (long long)qblocks * B);
  const int max_splits = pn_cdiv(Nc, CH_TILE);
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  int chunk = pn_cdiv(Nc, splits);
  chunk = (int)pn_align_up(chunk, 64);
  splits = pn_cdiv(Nc, chunk);
  dim3 grid(qblocks, splits, B);
  {
    PN_PROF("chamfer_nn", stream);
    hipLaunchKernelGGL(pn_chamfer_nn_kernel, grid, dim3(CH_THREADS), 0, stream, q, Nq, c,
                       Nc, chunk, packed);
  }
  PN_CHECK_LAUNCH();
  hipLaunchKernelGGL(pn_chamfer_unpack_kernel, dim3(pn_cdiv(n, 256)), dim3(256), 0, stream,
                     packed, n, mind, arg);
  PN_CHECK_LAUNCH();
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
    rc = chamfer_one_side(a, Na, b, Nb, B, pa, minA, argA, stream);
    if (rc) return rc;
  }
  if (minB || argB) {
    rc = chamfer_one_side(b, Nb, a, Na, B, pb, minB, argB, stream);
    if (rc) return rc;
  }
  return PN_OK;
}
