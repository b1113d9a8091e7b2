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
//
// Round 6 — the large searches run a PRE-FILTER on the bf16 matrix cores and decide exactly (pn_chamfer_mfma_kernel):
//   D(q, c) = |q|^2 + |c|^2 - 2 q.c; a 32 candidates x 32 queries tile of  t(q, c) = |c|^2 - 2 q.c - e(q, c)  is ONE
//   v_mfma_f32_32x32x16_bf16: both points split into two bf16 pieces (16 of the 24 bits), the four piece products
//   of the three coordinates in 12 of the 16 contraction slots, |c|^2 (lowered by 2^-18, three pieces) in three,
//   and the error budget e = 2^-13 |q| |c| (both factors rounded up) in the last one.  t is a CERTIFIED lower
//   bound of D - |q|^2 (dropped piece products <= 2^-14 |q||c|, fp32 accumulation of 16 terms <= 2^-19 |q||c| +
//   2^-20 |c|^2), so a candidate with  t > best - |q|^2 (+ 2^-20 slack for the roundings of the exact chain)
//   cannot be the minimum nor tie with it.  Everything else — a handful of candidates per query: the record lows of
//   its scan and the near ties — is evaluated with the reference's chain ((dx^2 + dy^2) + dz^2, each operation rounded
//   once) and decided on (distance, index): minima and arg-mins are bit-identical to the scalar kernel's and the
//   oracle's.  A lane holds 16 candidates of its query per tile: eight v_min3 and a compare per 16 pairs instead
//   of 8.9 instructions per pair.  The waves of a workgroup scan interleaved candidate tiles for the SAME 32 queries
//   and share their running minima through LDS, so the thresholds tighten as fast as in one long scan.
#include "split_common.h"
#include <stdlib.h>

#define CH_THREADS 128
#define CH_GROUP 8
#define CH_DEPTH 1        // tiles of prefetch in the matrix-core kernel (4 measured slower: 103 registers, 0.068 ms)

typedef float ch_f32x16 __attribute__((ext_vector_type(16)));

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

// ---- matrix-core pre-filter -------------------------------------------------------------------------------------
// images of the points, one 64-byte record per point: [candidate row 16 bf16][query row 16 bf16], contraction slots
//   candidate: chx chy chz cmx cmy cmz chx chy | chz cmx cmy cmz n0 n1 n2 cn
//   query:     -2 (qhx qhy qhz qhx qhy qhz qmx qmy | qmz qmx qmy qmz)  1 1 1  -eq
// n0 + n1 + n2 = |c|^2 (1 - 2^-18), cn >= |c|, eq >= 2^-13 |q|.
__device__ static inline float ch_bf16_rn(float x) {
  bf16x2 p = __builtin_convertvector(f32x2{x, 0.f}, bf16x2);
  return (float)p[0];
}
__device__ static inline uint32_t ch_pack(float a, float b) {   // two bf16-representable floats
  return (__float_as_uint(a) >> 16) | (__float_as_uint(b) & 0xffff0000u);
}

__global__ __launch_bounds__(256) void pn_chamfer_image_kernel(const float* __restrict__ a, long long na,
                                                               const float* __restrict__ b, long long nb,
                                                               u32x4* __restrict__ img) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= na + nb) return;
  const float* p = i < na ? a + 3 * i : b + 3 * (i - na);
  const float x = p[0], y = p[1], z = p[2];
  const float hx = ch_bf16_rn(x), hy = ch_bf16_rn(y), hz = ch_bf16_rn(z);
  const float mx = ch_bf16_rn(x - hx), my = ch_bf16_rn(y - hy), mz = ch_bf16_rn(z - hz);
  const float nn = (x * x + y * y) + z * z;
  const float nl = nn * (1.0f - 0x1p-18f);
  const float n0 = ch_bf16_rn(nl), n1 = ch_bf16_rn(nl - n0), n2 = ch_bf16_rn((nl - n0) - n1);
  const float nrm = sqrtf(nn);
  const float cn = ch_bf16_rn(nrm * (1.0f + 0x1p-6f));                 // >= |c|  (bf16 rounds by <= 2^-8)
  const float eq = ch_bf16_rn(nrm * (0x1p-13f * (1.0f + 0x1p-6f)));    // >= 2^-13 |q|
  u32x4 c0, c1, q0, q1;
  c0[0] = ch_pack(hx, hy), c0[1] = ch_pack(hz, mx), c0[2] = ch_pack(my, mz), c0[3] = ch_pack(hx, hy);
  c1[0] = ch_pack(hz, mx), c1[1] = ch_pack(my, mz), c1[2] = ch_pack(n0, n1), c1[3] = ch_pack(n2, cn);
  const float ax = -2.f * hx, ay = -2.f * hy, az = -2.f * hz, bx = -2.f * mx, by = -2.f * my, bz = -2.f * mz;
  q0[0] = ch_pack(ax, ay), q0[1] = ch_pack(az, ax), q0[2] = ch_pack(ay, az), q0[3] = ch_pack(bx, by);
  q1[0] = ch_pack(bz, bx), q1[1] = ch_pack(by, bz), q1[2] = ch_pack(1.f, 1.f), q1[3] = ch_pack(1.f, -eq);
  u32x4* o = img + 4 * i;
  o[0] = c0, o[1] = c1, o[2] = q0, o[3] = q1;
}

struct ChSide {
  const float* q;       // query coordinates (rows of 3)
  const float* c;       // candidate coordinates
  const u32x4* qimg;    // image records of the queries' cloud / the candidates' cloud
  const u32x4* cimg;
  const int* qoff;      // ragged: item offsets (B + 1) into the rows, else null
  const int* coff;
  int Nq, Nc;           // uniform item sizes (grid sizing when ragged: the largest item)
  float* mind;
  int64_t* arg;
};

// grid (ceil(max Nq / 32), B, sides); NW waves: wave w scans the candidate tiles w, w + NW, ... for the 32 queries
template <int NW>
__global__ __launch_bounds__(64 * NW) void pn_chamfer_mfma_kernel(ChSide s0, ChSide s1) {
  __shared__ unsigned s_best[32];
  __shared__ unsigned long long s_key[32];
  const ChSide& S = blockIdx.z == 0 ? s0 : s1;
  const int b = blockIdx.y;
  const int q0 = S.qoff ? S.qoff[b] : b * S.Nq;
  const int Nq = S.qoff ? S.qoff[b + 1] - q0 : S.Nq;
  const int c0 = S.coff ? S.coff[b] : b * S.Nc;
  const int Nc = S.coff ? S.coff[b + 1] - c0 : S.Nc;
  const int qbase = blockIdx.x * 32;
  if (qbase >= Nq) return;                     // block-uniform
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, col = lane & 31, h = lane >> 5;
  if (tid < 32) {
    s_best[tid] = 0x7f800000u;                 // +inf
    s_key[tid] = ~0ull;
  }
  const int i = qbase + col;
  const int iq = i < Nq ? i : Nq - 1;
  const float* qp = S.q + 3 * (size_t)(q0 + iq);
  const float qx = qp[0], qy = qp[1], qz = qp[2];
  const float nq = (qx * qx + qy * qy) + qz * qz;
  const bf16x8 bq = x3_as_bf16(S.qimg[4 * (size_t)(q0 + iq) + 2 + h]);
  const float* cb = S.c + 3 * (size_t)c0;
  const u32x4* __restrict__ ci = S.cimg + 4 * (size_t)c0;
  unsigned long long key = ~0ull;              // (bits of the exact minimum << 32) | candidate index
  float bestf = __builtin_inff();
  const int ntiles = (Nc + 31) >> 5;
  __syncthreads();
  // CH_DEPTH tiles ahead: the bf16 record of the lane's row AND its fp32 coordinates (lane l and l + 32 hold candidate
  // 32 t + (l & 31)): the exact evaluations below take their operands from a lane shuffle — a candidate that passes
  // the filter costs ~30 instructions, not a memory latency.  (First versions: operands loaded per hit — 0.7 us each,
  // serialised by the branches — and then ONE tile of prefetch: a wave's 39 tiles each waited a full L2 latency,
  // 0.046 ms at 10k x 10k where the scalar kernel takes 0.061.)
  u32x4 aq[CH_DEPTH];
  float cxq[CH_DEPTH], cyq[CH_DEPTH], czq[CH_DEPTH];
#define CH_LOAD(D_, T_)                                                          \
  {                                                                              \
    const int row_ = min(32 * (T_) + col, Nc - 1);                               \
    aq[D_] = ci[4 * (size_t)row_ + h];                                           \
    cxq[D_] = cb[3 * (size_t)row_], cyq[D_] = cb[3 * (size_t)row_ + 1], czq[D_] = cb[3 * (size_t)row_ + 2]; \
  }
#pragma unroll
  for (int d = 0; d < CH_DEPTH; ++d) {
    aq[d] = u32x4{0u, 0u, 0u, 0u};
    cxq[d] = cyq[d] = czq[d] = 0.f;
    if (wave + d * NW < ntiles) CH_LOAD(d, wave + d * NW);
  }
  for (int t0 = wave; t0 < ntiles; t0 += NW * CH_DEPTH) {
#pragma unroll
    for (int d = 0; d < CH_DEPTH; ++d) {
      const int t = t0 + d * NW;
      if (t >= ntiles) break;
      const bf16x8 a = x3_as_bf16(aq[d]);
      const float ccx = cxq[d], ccy = cyq[d], ccz = czq[d];
      if (t + NW * CH_DEPTH < ntiles) CH_LOAD(d, t + NW * CH_DEPTH);
      ch_f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bq, acc, 0, 0, 0);
      // the smallest of the lane's 16 bounds (rows past the end of the cloud repeat its last point: harmless)
      float m = fminf(fminf(acc[0], acc[1]), acc[2]);
#pragma unroll
      for (int r = 3; r < 15; r += 2) m = fminf(fminf(m, acc[r]), acc[r + 1]);
      m = fminf(m, acc[15]);
      const float bcur = fminf(bestf, __uint_as_float(s_best[col]));
      const float thr = (bcur - nq) + 0x1p-20f * (bcur + nq);       // inf while nothing is known: everything passes
      if (__ballot(!(m > thr))) {
        bool improved = false;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const bool hit = !(acc[r] > thr);
          if (__ballot(hit)) {                   // wave-uniform: the shuffles below run with every lane active
            const int rowl = (r & 3) + 8 * (r >> 2) + 4 * h;
            const float x = __shfl(ccx, rowl, 64), y = __shfl(ccy, rowl, 64), z = __shfl(ccz, rowl, 64);
            const int j = 32 * t + rowl;
            if (hit && j < Nc) {
              const float d2 = ch_dist(qx, qy, qz, x, y, z);
              if (d2 < __builtin_inff()) {        // (NaN / inf distances never win: the scalar kernel's rule)
                const unsigned long long k = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)j;
                if (k < key) {
                  key = k;
                  bestf = d2;
                  improved = true;
                }
              }
            }
          }
        }
        if (improved) atomicMin(&s_best[col], __float_as_uint(bestf));   // d >= 0: the bit pattern orders like the value
      }
    }
  }
#undef CH_LOAD
  // smallest (distance, index) over the two half waves, then over the waves
  {
    const unsigned lo = __shfl_xor((unsigned)key, 32, 64), hi = __shfl_xor((unsigned)(key >> 32), 32, 64);
    const unsigned long long other = ((unsigned long long)hi << 32) | lo;
    if (other < key) key = other;
  }
  if (h == 0 && key != ~0ull) atomicMin(&s_key[col], key);
  __syncthreads();
  if (tid < 32 && qbase + tid < Nq) {
    const unsigned long long k = s_key[tid];
    const size_t o = (size_t)q0 + qbase + tid;
    if (S.mind) S.mind[o] = __uint_as_float((uint32_t)(k >> 32));
    if (S.arg) S.arg[o] = (int64_t)(uint32_t)(k & 0xffffffffu);
  }
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
  // (ONE candidate range per query as soon as the query workgroups alone number a few hundred — no 64-bit atomic
  //  merge, no memset, no unpack launch — was measured in round 6, tools/jobs/r6k.sh: 32 x 1600 x 700 0.047 against
  //  0.037 ms with the split ranges, 0.079 from 128 workgroups on: the waves' scans get three times longer and the
  //  chip is not full; PN_CHAMFER_DIRECT_WGS=<n> re-enables it for an A/B)
  {
    const char* e = getenv("PN_CHAMFER_DIRECT_WGS");
    const long long direct_from = e ? atoll(e) : 0;
    if (direct_from > 0 && (long long)qblocks * B >= direct_from) splits = 1;
  }
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

// Measured (tools/kbench.py chamfer, profiles/r06_named_kernels_kbench.txt): 10k x 10k 0.047 against 0.061 ms for the
// scalar kernel; 32 x 1600 x 700 0.049 against 0.038; ragged items of ~900 x 2000: equal.  A wave's scan has to be
// long for its threshold to tighten (every candidate that beats the running minimum costs an exact evaluation and
// the first tiles of a scan all do), so the matrix-core kernel takes the searches with >= 4 096 points on both sides
// — the 10 000-point coverage distances of test.py:157-168.  PN_CHAMFER_MFMA=0 / 1 forces the scalar / the
// matrix-core kernel (tests run both on every size).
static bool chamfer_use_mfma(long long items, int Nq, int Nc) {
  (void)items;
  if (const char* e = getenv("PN_CHAMFER_MFMA")) return atoi(e) != 0;
  return Nq >= 4096 && Nc >= 4096;
}

static size_t chamfer_image_bytes(long long total) { return pn_align_up((size_t)total * 64, 256); }

// both clouds' images in one launch, then ONE launch for the requested sides (blockIdx.z)
static int chamfer_mfma(const float* a, const int* offA, int Na, long long totalA, const float* b, const int* offB, int Nb,
                        long long totalB, int B, float* minA, int64_t* argA, float* minB, int64_t* argB, void* img_,
                        hipStream_t stream) {
  u32x4* img = (u32x4*)img_;
  {
    PN_PROF("chamfer_image", stream);
    hipLaunchKernelGGL(pn_chamfer_image_kernel, dim3(pn_cdiv(totalA + totalB, 256)), dim3(256), 0, stream, a, totalA, b,
                       totalB, img);
  }
  PN_CHECK_LAUNCH();
  const u32x4* imgA = img;
  const u32x4* imgB = img + 4 * (size_t)totalA;
  ChSide sa = {a, b, imgA, imgB, offA, offB, Na, Nb, minA, argA};
  ChSide sb = {b, a, imgB, imgA, offB, offA, Nb, Na, minB, argB};
  const bool wantA = minA || argA, wantB = minB || argB;
  ChSide s0 = wantA ? sa : sb, s1 = sb;
  const int sides = (wantA ? 1 : 0) + (wantB ? 1 : 0);
  const int nq_max = sides == 2 ? (Na > Nb ? Na : Nb) : s0.Nq;
  const int nc_min = sides == 2 ? (Na < Nb ? Na : Nb) : s0.Nc;
  const int ntiles = pn_cdiv(nc_min, 32);
  dim3 grid(pn_cdiv(nq_max, 32), B, sides);
  {
    PN_PROF("chamfer_nn", stream);
    if (ntiles >= 64)
      hipLaunchKernelGGL(pn_chamfer_mfma_kernel<8>, grid, dim3(512), 0, stream, s0, s1);
    else if (ntiles >= 16)
      hipLaunchKernelGGL(pn_chamfer_mfma_kernel<4>, grid, dim3(256), 0, stream, s0, s1);
    else if (ntiles >= 4)
      hipLaunchKernelGGL(pn_chamfer_mfma_kernel<2>, grid, dim3(128), 0, stream, s0, s1);
    else
      hipLaunchKernelGGL(pn_chamfer_mfma_kernel<1>, grid, dim3(64), 0, stream, s0, s1);
  }
  PN_CHECK_LAUNCH();
  return PN_OK;
}

extern "C" size_t pn_chamfer_nn_workspace(int B, int Na, int Nb) {
  return pn_align_up((size_t)B * Na * 8, 256) + pn_align_up((size_t)B * Nb * 8, 256) +
         chamfer_image_bytes((long long)B * Na + (long long)B * Nb);
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
  if (chamfer_use_mfma(B, Na, Nb))
    return chamfer_mfma(a, nullptr, Na, (long long)B * Na, b, nullptr, Nb, (long long)B * Nb, B, minA, argA, minB, argB,
                        (char*)pb + pn_align_up((size_t)B * Nb * 8, 256), stream);
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
  return pn_align_up((size_t)totalA * 8, 256) + pn_align_up((size_t)totalB * 8, 256) +
         chamfer_image_bytes((long long)totalA + totalB);
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
  // (the items' sizes are on the device: decide on the mean item)
  if (chamfer_use_mfma(B, pn_cdiv(totalA, B), pn_cdiv(totalB, B)))
    return chamfer_mfma(a, offA, maxA, totalA, b, offB, maxB, totalB, B, minA, argA, minB, argB,
                        (char*)pb + pn_align_up((size_t)totalB * 8, 256), stream);
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
