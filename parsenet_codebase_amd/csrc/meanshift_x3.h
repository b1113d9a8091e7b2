// Mean-shift iterations with fp32-grade products on the bf16 matrix cores ("bf16 x 3").
//
// Same mathematics, data flow and outputs as meanshift.hip; only the two GEMMs per tile change.
// Every fp32 operand x is split without error into three bf16 pieces,
//     x = xh + xm + xl,  xh = bf16(x), xm = bf16(x - xh), xl = bf16(x - xh - xm)
// (24 significand bits = 3 x 8; the residuals are exact in fp32), and a product x*y is
// evaluated as  xh*yh + xh*ym + xm*yh + xm*ym + xh*yl + xl*yh  with fp32 accumulation on
// v_mfma_f32_32x32x16_bf16.  Each bf16 x bf16 product is exact in fp32; the dropped terms
// (xm*yl, xl*ym, xl*yl) are below 2^-25 |x*y|, i.e. below the rounding of an fp32 product.  The
// result is therefore an fp32 dot product with a different (tree) summation order — not a
// reduced-precision one; tests/test_meanshift_gpu.py measures it against the fp64 oracle next
// to the exact-fp32 path.  Six bf16 MFMAs (32 cycles each) replace eight fp32 MFMAs
// (64 cycles each) per 32x32x16 block: 2.7x less matrix-pipe time, and bf16 MFMAs overlap with
// the VALU work of the elementwise stage, which the fp32 ones do not.
//
// Streamed operands come pre-split from global memory as LDS images: one 24 KiB image per
// 32-point tile, written by pn_ms3_split_kernel and copied verbatim by the LDS DMA:
//   [piece 3][row j 32][16 chunks of 8 channels], chunk c of row j stored at c ^ swz(j),
//   swz(j) = ((j & 3) << 2) | ((j >> 2) & 3).
// The same image feeds both GEMMs.  The first one (contraction over channels) fetches its A
// operand with ds_read_b128 (row = streamed point); the second one (contraction over the
// streamed points) needs the transposed operand and gets it from the hardware transpose read
// ds_read_b64_tr_b16: within a 16-lane group lane i supplies the address of 4 contiguous bf16
// (row i>>2, columns 4(i&3)..) of a 4 x 16 block and receives column i of that block, i.e. 4
// consecutive streamed points of one feature — exactly the 4-point runs in which the D layout
// of the first GEMM hands the kernel values to the second (positions e, e+4 of a k-step are
// points (e&3) + 8(2t + (e>>2)) + 4h).  The swizzle makes both access patterns bank-conflict
// free (16-lane service groups of ds_read_b128; 4 rows x 4 chunks of the transpose read).
// (included at the end of meanshift.hip: one translation unit, shared combine kernels)

#include "split_common.h"

#define X3_IMG_U4 1536            // uint4 (16 B) units per 24 KiB tile image
#define X3_PIECE_U4 512           // per piece

// x (B,N,D) fp32 -> the image of every 32-point tile (rows >= N are zero).
// One workgroup per tile; work item = one 16-byte chunk.
__global__ __launch_bounds__(256) void pn_ms3_split_kernel(const float* __restrict__ x, int N,
                                                           int ntiles, u32x4* __restrict__ pimg) {
  const int b = blockIdx.y, tile = blockIdx.x;
  const float* __restrict__ xb = x + (size_t)b * N * MS_D;
  u32x4* __restrict__ P = pimg + ((size_t)b * ntiles + tile) * X3_IMG_U4;
  const int j0 = tile * 32;
  for (int it = threadIdx.x; it < 512; it += 256) {
    const int j = it >> 4, c = it & 15;  // row j, chunk c = channels 8c..8c+7
    float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
    if (j0 + j < N) {
      const float* src = xb + (size_t)(j0 + j) * MS_D + 8 * c;
      v0 = *reinterpret_cast<const float4*>(src);
      v1 = *reinterpret_cast<const float4*>(src + 4);
    }
    u32x4 h, m, l;
    X3_SPLIT_TO(v0.x, v0.y, h, m, l, 0);
    X3_SPLIT_TO(v0.z, v0.w, h, m, l, 1);
    X3_SPLIT_TO(v1.x, v1.y, h, m, l, 2);
    X3_SPLIT_TO(v1.z, v1.w, h, m, l, 3);
    const int slot = j * 16 + (c ^ x3_swz(j));
    P[slot] = h;
    P[X3_PIECE_U4 + slot] = m;
    P[2 * X3_PIECE_U4 + slot] = l;
  }
}

// backward prologue, one wave per row (as pn_ms_prep_bwd_kernel, without the transposed copies):
//   gu = (gy - y (y.gy)) / ||u|| ; c = gu . u (u = y ||u||) ; alpha = 1 / (r b^2)
__global__ __launch_bounds__(256) void pn_ms3_prep_bwd_kernel(
    const float* __restrict__ gy, const float* __restrict__ y, const float* __restrict__ rsum,
    const float* __restrict__ unorm, const float* __restrict__ bsq, int N, float* __restrict__ gu,
    float* __restrict__ cs, float* __restrict__ alpha) {
  const int b = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + wave;
  if (i >= N) return;
  const size_t base = ((size_t)b * N + i) * MS_D;
  const float y0 = y[base + lane], y1 = y[base + lane + 64];
  const float g0 = gy[base + lane], g1 = gy[base + lane + 64];
  const float nn = unorm[(size_t)b * N + i], r = rsum[(size_t)b * N + i];
  const float yg = pn_wave_sum(y0 * g0 + y1 * g1);
  const float u0 = (g0 - y0 * yg) / nn, u1 = (g1 - y1 * yg) / nn;
  const float c = pn_wave_sum(u0 * (y0 * nn) + u1 * (y1 * nn));
  gu[base + lane] = u0;
  gu[base + lane + 64] = u1;
  if (lane == 0) {
    cs[(size_t)b * N + i] = c;
    alpha[(size_t)b * N + i] = 1.0f / (r * bsq[b]);
  }
}

// The backward prologue in ONE launch (round 4; three before: pn_ms3_prep_bwd_kernel + two
// pn_ms3_split_kernel): one workgroup per 32-row tile computes gu, c and alpha of its rows (the same
// arithmetic, one wave per row, eight rows per wave), keeps the gu rows in LDS and writes the tile
// images of q and of gu.
__global__ __launch_bounds__(256) void pn_ms3_prologue_bwd_kernel(
    const float* __restrict__ gy, const float* __restrict__ y, const float* __restrict__ q,
    const float* __restrict__ rsum, const float* __restrict__ unorm, const float* __restrict__ bsq, int N, int ntiles,
    float* __restrict__ gu, float* __restrict__ cs, float* __restrict__ alpha, u32x4* __restrict__ img_q,
    u32x4* __restrict__ img_gu) {
  __shared__ __attribute__((aligned(16))) float gt[32][MS_D];
  const int b = blockIdx.y, tile = blockIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j0 = tile * 32;
  for (int r = wave; r < 32; r += 4) {
    const int i = j0 + r;
    float u0 = 0.f, u1 = 0.f;
    if (i < N) {
      const size_t base = ((size_t)b * N + i) * MS_D;
      const float y0 = y[base + lane], y1 = y[base + lane + 64];
      const float g0 = gy[base + lane], g1 = gy[base + lane + 64];
      const float nn = unorm[(size_t)b * N + i], rr = rsum[(size_t)b * N + i];
      const float yg = pn_wave_sum(y0 * g0 + y1 * g1);
      u0 = (g0 - y0 * yg) / nn;
      u1 = (g1 - y1 * yg) / nn;
      const float c = pn_wave_sum(u0 * (y0 * nn) + u1 * (y1 * nn));
      gu[base + lane] = u0;
      gu[base + lane + 64] = u1;
      if (lane == 0) {
        cs[(size_t)b * N + i] = c;
        alpha[(size_t)b * N + i] = 1.0f / (rr * bsq[b]);
      }
    }
    gt[r][lane] = u0;            // rows >= N: zero image rows
    gt[r][lane + 64] = u1;
  }
  __syncthreads();
  const float* __restrict__ qb = q + (size_t)b * N * MS_D;
  u32x4* __restrict__ Pq = img_q + ((size_t)b * ntiles + tile) * X3_IMG_U4;
  u32x4* __restrict__ Pg = img_gu + ((size_t)b * ntiles + tile) * X3_IMG_U4;
  for (int it = threadIdx.x; it < 512; it += 256) {
    const int j = it >> 4, c = it & 15;  // row j, chunk c = channels 8c..8c+7
    float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
    if (j0 + j < N) {
      const float* src = qb + (size_t)(j0 + j) * MS_D + 8 * c;
      v0 = *reinterpret_cast<const float4*>(src);
      v1 = *reinterpret_cast<const float4*>(src + 4);
    }
    const float4 w0 = *reinterpret_cast<const float4*>(&gt[j][8 * c]);
    const float4 w1 = *reinterpret_cast<const float4*>(&gt[j][8 * c + 4]);
    const int slot = j * 16 + (c ^ x3_swz(j));
    {
      u32x4 h, m, l;
      X3_SPLIT_TO(v0.x, v0.y, h, m, l, 0);
      X3_SPLIT_TO(v0.z, v0.w, h, m, l, 1);
      X3_SPLIT_TO(v1.x, v1.y, h, m, l, 2);
      X3_SPLIT_TO(v1.z, v1.w, h, m, l, 3);
      Pq[slot] = h;
      Pq[X3_PIECE_U4 + slot] = m;
      Pq[2 * X3_PIECE_U4 + slot] = l;
    }
    {
      u32x4 h, m, l;
      X3_SPLIT_TO(w0.x, w0.y, h, m, l, 0);
      X3_SPLIT_TO(w0.z, w0.w, h, m, l, 1);
      X3_SPLIT_TO(w1.x, w1.y, h, m, l, 2);
      X3_SPLIT_TO(w1.z, w1.w, h, m, l, 3);
      Pg[slot] = h;
      Pg[X3_PIECE_U4 + slot] = m;
      Pg[2 * X3_PIECE_U4 + slot] = l;
    }
  }
}

// six-term product of two split operands into two accumulators (large and small terms apart)
#define X3_MFMA(ACC, A, B) ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, ACC, 0, 0, 0)

// PASS 0 forward       resident rows Q;      streamed X:      out[f][i] += X[j][f] K
// PASS 1 backward/rows resident rows Q, GU;  streamed X:      out[f][i] += X[j][f] gs
// PASS 2 backward/cols resident cols X;      streamed Q, GU:  out[f][j] += Q[i][f] gs + GU[i][f] K / r_i
//
// R, R1       (B,N,D) fp32 resident operands (split in registers once per workgroup)
// PA, PB      tile images of the streamed operand(s) (PB: GU, PASS 2 only); both GEMMs read them
// cs, rs      per-row c_i and alpha_i = 1/(r_i b^2): of the resident row (PASS 1) / streamed (PASS 2)
// grid (slices, blocks of 128 resident indices, B), 256 threads: wave w owns 32 w .. 32 w + 31.
// LDS: images double buffered: 48 KiB (PASS 0/1), 96 KiB (PASS 2); one barrier per tile
// (MODE 0; the ping-pong schedule of MODE 1: three buffers, two barriers per tile — below).

// Scalar (SMEM) loads of plan data inside the ping-pong loop: the lists and pair flags are
// read-only for the whole launch, and through the constant address space a uniform address is a
// s_load_dword tracked by lgkmcnt — the loop then has no vector-memory operation besides the
// LDS DMA, so no compiler-placed vmcnt(0) (all it can express once a DMA is in flight) ever waits
// for an image that was only just requested.
typedef const __attribute__((address_space(4))) int* x3_cptr_i;
typedef const __attribute__((address_space(4))) unsigned* x3_cptr_u;
__device__ static inline int x3_cint(const int* p) { return ((x3_cptr_i)p)[0]; }
__device__ static inline int x3_cflag(const unsigned char* p) {   // one byte via its aligned dword
  const uintptr_t a = (uintptr_t)p;
  const unsigned raw = ((x3_cptr_u)(a & ~(uintptr_t)3))[0];
  return (int)((raw >> ((unsigned)(a & 3) * 8u)) & 0xffu);
}

// developer switch (-DX3_PACKED_VALU): the elementwise stage of the 8-wave passes with packed fp32
// VALU instructions as before round 3 (A/B of tools/jobs/r3zb.sh, r3zc.sh)
#ifdef X3_PACKED_VALU
#define X3_PACKED true
#else
#define X3_PACKED false
#endif
#define X3_WAVES(PASS) ((PASS) == 1 ? 4 : 8)  // forward / column pass: 8 waves (2 per SIMD) share the LDS images
// PP ("ping-pong", 8-wave passes only): the two waves of a SIMD run half a tile apart.  With one
// barrier per tile both waves of a SIMD enter the elementwise stage (VALU only) of the same tile
// one after the other while the matrix pipe has nothing else to do for the later one (in-kernel
// timers, column pass: one wave of the SIMD waits 4 000 of 18 500 cycles per tile at the barrier
// for its sibling, whose stage nobody overlaps).  Here one wave of every SIMD leads and the other
// trails by one half step (`grp`, decided per SIMD at run time): a tile is two half steps
// (H1: first GEMM + first half of the stage, H2: second GEMM) separated by workgroup barriers, so that in every
// half step one wave of the SIMD is in H1 and the other in H2 and the stage of either is covered
// by MFMAs of the other.  An image lives for 3 half steps: three LDS buffers; each wave issues its
// share of the DMA for tile k + 2 at the END of its own H2(k) — the leading and the trailing
// waves at different times, behind the MFMAs of the other group, instead of all eight in one
// burst behind the barrier (325 / 1 400 cycles per tile in which no wave issued an MFMA).
// Vector-memory traffic of the loop is the DMA alone (plan data through scalar loads), and the
// only vmcnt wait is the explicit one at the end of H1: the DMA has a whole half step to land.
// Executed work, counted by the launches themselves: every wave counts the (resident 32-row tile,
// streamed tile) pairs whose GEMMs it actually runs — all of them in a dense launch, the pairs its
// plan keeps in a planned one — and adds the count to pn_ms3_exec[PASS] when it retires (one integer
// atomic per wave and launch).  One pair = 2 * 32 * 32 * 128 FLOP per GEMM unit (forward 2, row pass
// 3, column pass 4 units) x 6 bf16 piece products: bench.py divides this by the event time of the
// same launches (pn_meanshift_x3_exec_tiles reads and clears the counters), and
// SQ_VALU_MFMA_BUSY_CYCLES / 32 x 32 768 FLOP of a counter run reproduces it.
__device__ unsigned long long pn_ms3_exec[3];

template <int PASS, int MODE>
__global__ __launch_bounds__(64 * X3_WAVES(PASS))
__attribute__((amdgpu_waves_per_eu(PASS == 1 ? 1 : 2, PASS == 1 ? 1 : 2))) void pn_ms3_kernel(
    const float* __restrict__ R, const float* __restrict__ R1, const u32x4* __restrict__ PA,
    const u32x4* __restrict__ PB, const float* __restrict__ cs, const float* __restrict__ rs, const float* __restrict__ bsq_, int N,
    int ntiles, int tiles_per_slice, float* __restrict__ opart, float* __restrict__ rpart,
    const unsigned char* __restrict__ pairs, const int* __restrict__ counts, const int* __restrict__ lists,
    const int* __restrict__ offs, int nblk_total, int blk_off, int nbp, int nbB, int cmin, int sstride) {
  constexpr int NIMG = PASS == 2 ? 2 : 1;
  // MODE 0: the round-2 schedule; 1: ping-pong (8-wave passes) / spread DMA (row pass)
  constexpr bool PP = MODE == 1 && X3_WAVES(PASS) == 8;   // the row pass runs one wave per SIMD: no ping-pong
  // spread DMA: the DMA pieces of the next image go between the k-steps of the first GEMM instead
  // of all behind the barrier (row pass: 780 of 7 860 cycles per tile in which the wave issues no
  // MFMA; the same in the 8-wave forward pass without ping-pong: 8 158 -> 8 145 cycles, nothing)
  constexpr bool SPREAD = PASS == 1 && MODE == 1;
  constexpr int NBUF = PP ? 3 : 2;
  __shared__ __attribute__((aligned(16))) u32x4 ldsP[NBUF][NIMG][X3_IMG_U4];
  __shared__ __attribute__((aligned(16))) float lds_sc[NBUF][64];
  const int tid = threadIdx.x;
  // (ping-pong: the wave number as a scalar — list / flag / image addresses become scalar too)
  const int wave = PP ? __builtin_amdgcn_readfirstlane(tid >> 6) : tid >> 6, lane = tid & 63;
  const int col = lane & 31, h = lane >> 5;
  constexpr int NW = X3_WAVES(PASS);
  // ping-pong group of this wave: 0 = leading, 1 = trailing, one of each per SIMD whatever the
  // placement of the waves (order of arrival at a per-SIMD counter; SIMD id = HW_ID[5:4])
  int grp = 0;
  if (PP) {
    __shared__ int lds_simd[4];
    if (tid < 4) lds_simd[tid] = 0;
    __syncthreads();
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    int g = 0;
    if (lane == 0) g = atomicAdd(&lds_simd[(hwid >> 4) & 3], 1);
    grp = __builtin_amdgcn_readfirstlane(g) & 1;
  }
  // Work of this workgroup.
  //  dense (no plan): grid (slices, resident blocks, B): one block, one slice of the full range.
  //  flat plan (offs): grid (G): the lists of all (batch item, resident block) pairs of the pass
  //    laid end to end (offs = exclusive prefix of their lengths) and cut into G equal ranges, so
  //    every workgroup visits the same number of streamed tiles whatever the lengths of the lists
  //    (no tail, one partial result per list fragment instead of a fixed number per block).
  //    Workgroup ids go round-robin over the 8 XCDs: id -> range (id & 7) * G / 8 + (id >> 3)
  //    gives every XCD one contiguous eighth of the sequence, i.e. neighbouring blocks of the
  //    locality order, whose lists name the same tile images: they stay in that XCD's L2.
  const bool flat = offs != nullptr;
  int S = gridDim.x, chunk = 0, e_lo = 0, e_hi = 0, fblk = 0;
  if (flat) {
    const int G = gridDim.x;
    const int g = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    const int total = offs[nbB];
    chunk = max((total + G - 1) / G, cmin);
    e_lo = g * chunk;
    e_hi = min(total, e_lo + chunk);
    if (e_lo >= e_hi) return;
    int lo = 0, hi = nbB;  // offs[lo] <= e_lo < offs[hi]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (offs[mid] <= e_lo) lo = mid; else hi = mid;
    }
    fblk = lo;
    S = sstride;
  }
  int nexec = 0;   // tile pairs this wave executes (wave-uniform)
  for (bool first_seg = true;; first_seg = false) {   // list fragments (exactly one without a flat plan)
  int b = blockIdx.z, rblk = blockIdx.y, slice = blockIdx.x;
  // block-sparse plan (pn_meanshift_x3_plan_f32): the streamed tiles this workgroup's resident
  // block interacts with at all (everything else is below rel_eps of the smallest row sum)
  const int* __restrict__ lst = nullptr;
  int t_begin, t_end;
  if (flat) {
    b = fblk / nbp;
    rblk = fblk - b * nbp;
    const int o0 = offs[fblk], o1 = offs[fblk + 1];
    t_begin = e_lo - o0;
    t_end = min(e_hi, o1) - o0;
    slice = e_lo / chunk - o0 / chunk;
    lst = lists + ((size_t)b * nblk_total + blk_off + rblk) * ntiles;
    e_lo = min(e_hi, o1);
    if (!first_seg) __syncthreads();  // every wave is done with the images of the previous fragment
  } else {
    t_begin = slice * tiles_per_slice;
    t_end = min(ntiles, t_begin + tiles_per_slice);
  }
  const int i0 = (rblk * NW + wave) * 32;
  const bool wave_on = i0 < N;
  // this wave's own tile against the streamed one: pairs[tQ][tX]
  const unsigned char* __restrict__ prow = nullptr;
  int pstride = 0;
  if (pairs) {
    const int wt = min(rblk * X3_WAVES(PASS) + wave, ntiles - 1);
    prow = pairs + (size_t)b * ntiles * ntiles + (PASS == 2 ? (size_t)wt : (size_t)wt * ntiles);
    pstride = PASS == 2 ? ntiles : 1;
  }
  const float bsqv = bsq_[b];
  const float hl = (0.5f / bsqv) * MS_LOG2E;
  const size_t bN = (size_t)b * N;
  const size_t boff = (size_t)b * ntiles * X3_IMG_U4;
  const u32x4* __restrict__ PAb = PA + boff;
  const u32x4* __restrict__ PBb = PASS == 2 ? PB + boff : nullptr;

  // a 24 KiB image = 24 chunks of 1 KiB (64 lanes x 16 B), dealt evenly to the NW waves
  // (ping-pong: scalar base of the wave's chunks + one 32-bit lane offset for every DMA)
  const unsigned lane16 = (unsigned)lane * 16u;
#define X3_STAGE(SRC, DST)                                                        \
  {                                                                               \
    if (PP) {                                                                     \
      const char* s_ = reinterpret_cast<const char*>(SRC) + (size_t)(wave * (24 / NW)) * 1024; \
      _Pragma("unroll") for (int u = 0; u < 24 / NW; ++u)                         \
        X3_GLDS16(s_ + (u * 1024 + lane16), &(DST)[(wave * (24 / NW) + u) * 64]); \
    } else {                                                                      \
      _Pragma("unroll") for (int u = 0; u < 24 / NW; ++u) {                       \
        const int q = wave * (24 / NW) + u;                                       \
        X3_GLDS16((SRC) + q * 64 + lane, &(DST)[q * 64]);                         \
      }                                                                           \
    }                                                                             \
  }
#define X3_STAGE_P(MT, BUF)                                                       \
  {                                                                               \
    X3_STAGE(PAb + (size_t)(MT) * X3_IMG_U4, ldsP[BUF][0]);                       \
    if (PASS == 2) {                                                              \
      X3_STAGE(PBb + (size_t)(MT) * X3_IMG_U4, ldsP[BUF][NIMG - 1]);              \
      if (wave == 0) { /* c_i | alpha_i of the 32 streamed rows */                \
        const int jc = min((MT) * 32 + (lane & 31), N - 1);                       \
        if (PP) { /* two half-wave DMAs from scalar bases (lane l writes word l of lds_sc) */ \
          const unsigned o_ = (unsigned)jc * 4u;                                  \
          if (lane < 32)                                                          \
            __builtin_amdgcn_global_load_lds((x3_gptr)(reinterpret_cast<const char*>(cs + bN) + o_), \
                                             (x3_lptr)&lds_sc[BUF][0], 4, 0, 0);  \
          else                                                                    \
            __builtin_amdgcn_global_load_lds((x3_gptr)(reinterpret_cast<const char*>(rs + bN) + o_), \
                                             (x3_lptr)&lds_sc[BUF][0], 4, 0, 0);  \
        } else {                                                                  \
          __builtin_amdgcn_global_load_lds((x3_gptr)((lane < 32 ? cs : rs) + bN + jc), \
                                           (x3_lptr)&lds_sc[BUF][0], 4, 0, 0);    \
        }                                                                         \
      }                                                                           \
    }                                                                             \
  }
  int cur = 0;
#define X3_TILE(E) (lst ? (PP ? x3_cint(lst + (E)) : lst[E]) : (E))
#define X3_FLAG(MT) (PP ? x3_cflag(prow + (size_t)(MT) * pstride) : (int)prow[(size_t)(MT) * pstride])
  // the first image is on its way while the resident rows are fetched and split
  // tile numbers two entries ahead and this wave's pair flag one entry ahead: neither load is
  // waited for between the barrier and the first MFMA (the row pass runs one wave per SIMD)
  int mt_cur = t_begin < t_end ? X3_TILE(t_begin) : 0;
  int mt_nxt = t_begin + 1 < t_end ? X3_TILE(t_begin + 1) : 0;
  int on_cur = prow && t_begin < t_end ? X3_FLAG(mt_cur) : 1;
  if (t_begin < t_end) X3_STAGE_P(mt_cur, 0);
  if (PP && t_begin + 1 < t_end) X3_STAGE_P(mt_nxt, 1);
  // resident operand(s) as B operands of the first GEMM: k-step s = channels 16 s + 8 h + e
  const int ires = min(i0 + col, N - 1);
  bf16x8 qh[8], qm[8], ql[8];
  bf16x8 uh[PASS == 1 ? 8 : 1], um[PASS == 1 ? 8 : 1], ul[PASS == 1 ? 8 : 1];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    {
      const float* src = R + (bN + ires) * MS_D + 16 * s + 8 * h;
      const float4 a = *reinterpret_cast<const float4*>(src);
      const float4 c = *reinterpret_cast<const float4*>(src + 4);
      u32x4 vh, vm, vl;
      X3_SPLIT_TO(a.x, a.y, vh, vm, vl, 0);
      X3_SPLIT_TO(a.z, a.w, vh, vm, vl, 1);
      X3_SPLIT_TO(c.x, c.y, vh, vm, vl, 2);
      X3_SPLIT_TO(c.z, c.w, vh, vm, vl, 3);
      qh[s] = x3_as_bf16(vh);
      qm[s] = x3_as_bf16(vm);
      ql[s] = x3_as_bf16(vl);
    }
    if (PASS == 1) {
      const float* src = R1 + (bN + ires) * MS_D + 16 * s + 8 * h;
      const float4 a = *reinterpret_cast<const float4*>(src);
      const float4 c = *reinterpret_cast<const float4*>(src + 4);
      u32x4 vh, vm, vl;
      X3_SPLIT_TO(a.x, a.y, vh, vm, vl, 0);
      X3_SPLIT_TO(a.z, a.w, vh, vm, vl, 1);
      X3_SPLIT_TO(c.x, c.y, vh, vm, vl, 2);
      X3_SPLIT_TO(c.z, c.w, vh, vm, vl, 3);
      uh[PASS == 1 ? s : 0] = x3_as_bf16(vh);
      um[PASS == 1 ? s : 0] = x3_as_bf16(vm);
      ul[PASS == 1 ? s : 0] = x3_as_bf16(vl);
    }
  }
  float c_res = 0.f, a_res = 0.f;
  if (PASS == 1) {
    c_res = cs[bN + ires];
    a_res = rs[bN + ires];
  }
  f32x16 acc_o[4];
#pragma unroll
  for (int fb = 0; fb < 4; ++fb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_o[fb][r] = 0.f;
  float rsum = 0.f;

#ifdef MS_TIMING
  unsigned long long tb0 = 0, tdma = 0, tg1 = 0, tew = 0, tb1 = 0, tg2 = 0, tall = __builtin_amdgcn_s_memtime();
#endif
  // raw barrier / DMA wait of the ping-pong schedule: no release fence (its vmcnt(0) would make
  // every wave wait for the DMA it has just issued); the images are only ever written by the DMA
#define X3_BAR() asm volatile("s_barrier" ::: "memory")
// s_waitcnt vmcnt(0) (gfx9 encoding: expcnt 7, lgkmcnt 15 = no wait) as the builtin: the
// compiler's counter tracking sees it and adds no vmcnt(0) of its own before the transpose reads
// of H2 (which it cannot tell apart from the buffer the DMA writes)
#define X3_WAIT_VM0() __builtin_amdgcn_s_waitcnt(0x0F70)
  if (PP) {
    X3_WAIT_VM0();                      // this wave's share of the first two images
    if (grp) X3_BAR();                  // the trailing waves sit out the first half step
  }
  for (int e_ = t_begin; e_ < t_end; ++e_) {
    const int mt = mt_cur;
    const int j0 = mt * 32;
    MS_T(U0);
    if (PP) X3_BAR();   // H1(k): the image(s) of tile k landed (every wave waited for its share)
    else __syncthreads();  // image(s) of tile mt landed; every wave is done with tile mt - 1
    MS_T(U1);
    int on_nxt = 1;
    int mt_nn = 0;
    if (e_ + 1 < t_end) {
      if (!PP && !SPREAD) X3_STAGE_P(mt_nxt, cur ^ 1);
      if (prow) on_nxt = X3_FLAG(mt_nxt);
      if (e_ + 2 < t_end) mt_nn = X3_TILE(e_ + 2);
    }
    // wave-level skip: this wave's 32 resident indices do not interact with the streamed tile
    const bool pair_on = on_cur != 0;
    const bool spread_now = SPREAD && e_ + 1 < t_end;
    const int mt_st = e_ + 1 < t_end ? mt_nxt : mt;
    MS_T(U2);
    u32x4 wh[2], wm[2], wl[2];                                   // weights of the second GEMM
    u32x4 vh[PASS == 2 ? 2 : 1], vm[PASS == 2 ? 2 : 1], vl[PASS == 2 ? 2 : 1];  // PASS 2: K
    // two partial accumulators (small / large terms) per product where registers allow (row
    // pass, one wave per SIMD); the 8-wave passes accumulate small-to-large into one
    constexpr bool TWO_ACC = PASS == 1;
    f32x16 sa, ta, sb_, tb_;
#define sb (*(TWO_ACC ? &sb_ : &sa))
#define tb (*(TWO_ACC ? &tb_ : &ta))
    const bool tail = j0 + 32 > N;
    float kv[16], gs[PASS == 0 ? 1 : 16];
    // (the forward pass runs two waves per SIMD in 256 registers: the other wave fills the
    // matrix pipe during the elementwise stage, and the pipelining registers would spill)
    constexpr bool PIPE = PASS != 0;      // elementwise stage in two halves around k-step 0
    if (wave_on && pair_on) {
      ++nexec;
      // ---- first GEMM: S[streamed][resident] (and T with the second operand) ----
      // ping-pong issue priorities (s_setprio; the elementwise stages run at 0):
      //   forward:  first GEMM 2 > second GEMM 0 — the wave in H1 takes the matrix pipe, drops back
      //             for its stage, and the stage runs under the sibling's second GEMM (without
      //             priorities the two GEMMs share the pipe, end together, and the stage of H1
      //             runs with the sibling waiting at the barrier);
      //   column pass (stage in two halves around k-step 0 of the second GEMM):
      //             k-step 0 of the second GEMM 3 > first GEMM 2 > k-step 1 0 — the sibling's
      //             first GEMM runs under the second half of the stage and finishes before
      //             k-step 1, whose MFMAs then cover the first half of the sibling's stage.
// issue priorities of the ping-pong schedule (developer overrides -DX3_P_...: tools/jobs/r3zh.sh)
#ifndef X3_P_G1
#define X3_P_G1 2
#endif
#ifndef X3_P_EW
#define X3_P_EW 0
#endif
#ifndef X3_P_G2A
#define X3_P_G2A 3
#endif
#ifndef X3_P_G2B
#define X3_P_G2B 0
#endif
#ifdef X3_NOPRIO   /* developer switch: the ping-pong schedule without issue priorities */
#define X3_PRIO_ON false
#else
#define X3_PRIO_ON true
#endif
#define X3_PRIO(P_)                          \
  if (PP && X3_PRIO_ON) {                    \
    __builtin_amdgcn_sched_barrier(0);       \
    __builtin_amdgcn_s_setprio(P_);          \
    __builtin_amdgcn_sched_barrier(0);       \
  }
      X3_PRIO(X3_P_G1);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        sa[r] = 0.f;
        ta[r] = 0.f;
        if (TWO_ACC) {
          sb_[r] = 0.f;
          tb_[r] = 0.f;
        }
      }
      const u32x4* __restrict__ lp = ldsP[cur][0];
      const u32x4* __restrict__ lp1 = ldsP[cur][NIMG - 1];
      const int rowoff = col * 16, sw = x3_swz(col);
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const int slot = rowoff + ((2 * s + h) ^ sw);
        const bf16x8 ah = x3_as_bf16(lp[slot]);
        const bf16x8 am = x3_as_bf16(lp[X3_PIECE_U4 + slot]);
        const bf16x8 al = x3_as_bf16(lp[2 * X3_PIECE_U4 + slot]);
        X3_MFMA(sb, al, qh[s]);
        X3_MFMA(sb, ah, ql[s]);
        X3_MFMA(sb, am, qm[s]);
        X3_MFMA(sa, am, qh[s]);
        X3_MFMA(sa, ah, qm[s]);
        X3_MFMA(sa, ah, qh[s]);
        if (PASS == 1) {  // T = X . GU: same streamed operand, second resident one
          const int z = PASS == 1 ? s : 0;
          X3_MFMA(tb, al, uh[z]);
          X3_MFMA(tb, ah, ul[z]);
          X3_MFMA(tb, am, um[z]);
          X3_MFMA(ta, am, uh[z]);
          X3_MFMA(ta, ah, um[z]);
          X3_MFMA(ta, ah, uh[z]);
        }
        if (PASS == 2) {  // T = GU . X: second streamed operand, same resident one
          const bf16x8 gh = x3_as_bf16(lp1[slot]);
          const bf16x8 gm = x3_as_bf16(lp1[X3_PIECE_U4 + slot]);
          const bf16x8 gl = x3_as_bf16(lp1[2 * X3_PIECE_U4 + slot]);
          X3_MFMA(tb, gl, qh[s]);
          X3_MFMA(tb, gh, ql[s]);
          X3_MFMA(tb, gm, qm[s]);
          X3_MFMA(ta, gm, qh[s]);
          X3_MFMA(ta, gh, qm[s]);
          X3_MFMA(ta, gh, qh[s]);
        }
        if (SPREAD && s < 24 / NW) {
          // unconditional (a branch here would cut the k-steps into separate scheduling regions):
          // behind the last tile of a fragment the piece re-reads the current image into the free
          // buffer, which nobody reads
          const int q_ = wave * (24 / NW) + s;
          X3_GLDS16(PAb + (size_t)mt_st * X3_IMG_U4 + q_ * 64 + lane, &ldsP[cur ^ 1][0][q_ * 64]);
        }
      }
      X3_PRIO(X3_P_EW);
      // large + small partial sums: one accumulator stays live across the barrier
      if (TWO_ACC) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          sa[r] += sb_[r];
          ta[r] += tb_[r];
        }
      }
#undef sb
#undef tb
      MS_T(U3);
#ifdef MS_TIMING
      tg1 += U3 - U2;
#endif
      MS_T(U5);
#ifdef MS_TIMING
      tb0 += U1 - U0;
      tdma += U2 - U1;
#endif
      // ---- elementwise stage on D[streamed = (r&3)+8(r>>2)+4h][resident = col], software
      // in two halves in the backward passes: k-step 0 of the second GEMM only needs D registers
      // 0..7, so the values 8..15 are processed after it (fewer live registers) ----
#define X3_EW_(R, MASKED)                                                          \
  {                                                                                \
    const int row = ((R) & 3) + 8 * ((R) >> 2) + 4 * h;                            \
    const float sv = sa[R];                                                        \
    const float dist = __builtin_fmaf(-2.0f, sv, 2.0f);                            \
    float a2 = -dist * hl;                                                         \
    if (PASS == 2 && !X3_PACKED) asm("" : "+v"(a2));   /* no v_pk_mul_f32 beside the sibling's MFMAs */ \
    const float a2c = __builtin_amdgcn_fmed3f(a2, -MS_LIM2, MS_LIM2);              \
    float k = __builtin_amdgcn_exp2f(a2c);                                         \
    /* padded points have all-zero image rows: they add nothing in the second GEMM whatever \
       their weight, so only the row sums of the forward pass need the mask */          \
    if (PASS == 0 && (MASKED) && j0 + row >= N) k = 0.f;                           \
    kv[R] = k;                                                                     \
    if (PASS == 0) rsum += k;                                                      \
    if (PASS != 0) {                                                               \
      const float tv = ta[R];                                                      \
      const float cc = PASS == 1 ? c_res : lds_sc[cur][row];                       \
      const float aa = PASS == 1 ? a_res : lds_sc[cur][32 + row];                  \
      if (PASS == 2) {                                                             \
        float ab = aa * bsqv;                                                      \
        if (!X3_PACKED) asm("" : "+v"(ab));                                        \
        float kk = k * ab;                   /* weight of the GU term: K / r_i */  \
        if (!X3_PACKED) asm("" : "+v"(kk));                                        \
        kv[R] = kk;                                                                \
      }                                                                            \
      float d_ = (tv - cc) * aa;                                                   \
      if (PASS == 2 && !X3_PACKED) asm("" : "+v"(d_));                             \
      float g = k * d_;                                                            \
      asm("" : "+v"(g));          /* keep the select a v_cndmask, not a branch */  \
      gs[PASS == 0 ? 0 : (R)] = a2c == a2 ? g : 0.f;                               \
    }                                                                              \
  }
#define X3_EW(R) X3_EW_(R, false)
// (weights of the second GEMM: scalar residual subtractions in the 8-wave passes, packed ones in the
// row pass — split_common.h, x3_split2_t)
#define X3_SPLIT_TW(A, B, VH, VM, VL, Q)                       \
  {                                                            \
    const X3Pieces _p = x3_split2_t<PASS == 1 || X3_PACKED>(A, B); \
    VH[Q] = _p.h;                                              \
    VM[Q] = _p.m;                                              \
    VL[Q] = _p.l;                                              \
  }
#define X3_SPLIT_W(T, Q)                                                                      \
  {                                                                                           \
    if (PASS == 0) {                                                                          \
      X3_SPLIT_TW(kv[8 * (T) + 2 * (Q)], kv[8 * (T) + 2 * (Q) + 1], wh[T], wm[T], wl[T], Q); \
    } else {                                                                                  \
      const int e = PASS == 0 ? 0 : 8 * (T) + 2 * (Q);                                        \
      X3_SPLIT_TW(gs[e], gs[e + (PASS == 0 ? 0 : 1)], wh[T], wm[T], wl[T], Q);               \
      if (PASS == 2) {                                                                        \
        const int tt = PASS == 2 ? (T) : 0;                                                   \
        X3_SPLIT_TW(kv[8 * (T) + 2 * (Q)], kv[8 * (T) + 2 * (Q) + 1], vh[tt], vm[tt], vl[tt], Q); \
      }                                                                                       \
    }                                                                                         \
  }
      if (PASS == 0 && tail) {  // only the last tile of the forward pass pays for the mask
#pragma unroll
        for (int r = 0; r < 16; ++r) X3_EW_(r, true);
      } else {
#pragma unroll
        for (int r = 0; r < (PIPE ? 8 : 16); ++r) X3_EW(r);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) X3_SPLIT_W(0, q);
      if (!PIPE) {
#pragma unroll
        for (int q = 0; q < 4; ++q) X3_SPLIT_W(1, q);
      }
      MS_T(U6);
#ifdef MS_TIMING
      tew += U6 - U5;
#endif
      if (PP) {         // H1(k) | H2(k)
        MS_T(V0);
        X3_WAIT_VM0();  // this wave's share of the image(s) of tile k + 1 (issued at the end of H2(k - 1))
        X3_BAR();
#ifdef MS_TIMING
        tb1 += __builtin_amdgcn_s_memtime() - V0;
#endif
      }
      MS_T(U7);
      // ---- second GEMM: out[f][resident] += sum_streamed C[f][streamed] w[streamed][resident];
      //      k-step t = D registers 8t..8t+7 of the first GEMM.  Operands of the next (t, fb)
      //      are fetched from LDS before the MFMAs of the current one. ----
      // A operand of k-step t, feature block fb, piece p: two transpose reads (4 points each) of
      // rows 16t + 4h + (0..3) and + 8, columns fb*32 + (lane & 31); lane i of a 16-lane group
      // points at row (i >> 2), columns 4 (i & 3).. of its block
      const char* lbase = reinterpret_cast<const char*>(ldsP[cur][0]);
      const char* lbase1 = reinterpret_cast<const char*>(ldsP[cur][NIMG - 1]);
      const int li = lane & 15, rb = 4 * h + (li >> 2), cb = 16 * ((lane >> 4) & 1) + 4 * (li & 3);
      const int sz0 = (((li >> 2) & 3) << 2) | (h & 3), sz1 = (((li >> 2) & 3) << 2) | ((h + 2) & 3);
      u32x4 xc[3], oc[PASS == 2 ? 3 : 1];
#define X3_TR(BASE, P_, T, W, FB)                                                               \
  __builtin_amdgcn_ds_read_tr16_b64_v4i16((x3_lds_s16x4)(                                        \
      (BASE) + (P_) * (X3_PIECE_U4 * 16) + (16 * (T) + 8 * (W) + rb) * 256 +                    \
      ((((FB) * 4 + (cb >> 3)) ^ ((W) ? sz1 : sz0)) << 4) + ((cb & 7) << 1)))
#define X3_LOAD_C(DX, DO, T, FB)                                                                \
  {                                                                                             \
    _Pragma("unroll") for (int p_ = 0; p_ < 3; ++p_) {                                          \
      const s16x4 lo_ = X3_TR(lbase, p_, T, 0, FB), hi_ = X3_TR(lbase, p_, T, 1, FB);           \
      const s16x8 v_ = __builtin_shufflevector(lo_, hi_, 0, 1, 2, 3, 4, 5, 6, 7);               \
      DX[p_] = __builtin_bit_cast(u32x4, v_);                                                   \
      if (PASS == 2) {                                                                          \
        const s16x4 lo1_ = X3_TR(lbase1, p_, T, 0, FB), hi1_ = X3_TR(lbase1, p_, T, 1, FB);     \
        const s16x8 w_ = __builtin_shufflevector(lo1_, hi1_, 0, 1, 2, 3, 4, 5, 6, 7);           \
        DO[PASS == 2 ? p_ : 0] = __builtin_bit_cast(u32x4, w_);                                 \
      }                                                                                         \
    }                                                                                           \
  }
      if (PIPE) X3_PRIO(X3_P_G2A);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if (PIPE && t == 1) {
          X3_PRIO(X3_P_G2B);
          // second half of the stage between the two k-steps (finer interleaving with the MFMAs
          // was measured: it costs registers and gains nothing)
#pragma unroll
          for (int r = 8; r < 16; ++r) X3_EW(r);
#pragma unroll
          for (int q = 0; q < 4; ++q) X3_SPLIT_W(1, q);
        }
        const bf16x8 bh = x3_as_bf16(wh[t]), bm = x3_as_bf16(wm[t]), bl = x3_as_bf16(wl[t]);
#pragma unroll
        for (int fb = 0; fb < 4; ++fb) {
          X3_LOAD_C(xc, oc, t, fb);
          const bf16x8 xh = x3_as_bf16(xc[0]), xm = x3_as_bf16(xc[1]), xl = x3_as_bf16(xc[2]);
          X3_MFMA(acc_o[fb], xl, bh);
          X3_MFMA(acc_o[fb], xh, bl);
          X3_MFMA(acc_o[fb], xm, bm);
          X3_MFMA(acc_o[fb], xm, bh);
          X3_MFMA(acc_o[fb], xh, bm);
          X3_MFMA(acc_o[fb], xh, bh);
          if (PASS == 2) {
            const int tt = PASS == 2 ? t : 0;
            const bf16x8 kh = x3_as_bf16(vh[tt]), km = x3_as_bf16(vm[tt]), kl = x3_as_bf16(vl[tt]);
            const bf16x8 oh = x3_as_bf16(oc[0]), om = x3_as_bf16(oc[PASS == 2 ? 1 : 0]),
                         ol = x3_as_bf16(oc[PASS == 2 ? 2 : 0]);
            X3_MFMA(acc_o[fb], ol, kh);
            X3_MFMA(acc_o[fb], oh, kl);
            X3_MFMA(acc_o[fb], om, km);
            X3_MFMA(acc_o[fb], om, kh);
            X3_MFMA(acc_o[fb], oh, km);
            X3_MFMA(acc_o[fb], oh, kh);
          }
        }
      }
#undef X3_LOAD_C
#undef X3_TR
#undef X3_PRIO
#undef X3_SPLIT_W
#undef X3_SPLIT_TW
#undef X3_EW
#undef X3_EW_
#ifdef MS_TIMING
      tg2 += __builtin_amdgcn_s_memtime() - U7;
#endif
    } else if (PP) {    // a wave that skips the pair (or has no rows) still keeps the half steps
      X3_WAIT_VM0();
      X3_BAR();
    } else if (SPREAD) {
      if (spread_now) X3_STAGE_P(mt_nxt, cur ^ 1);
    }
    if (PP) {
      // the image(s) of tile k + 2 into the buffer of tile k - 1 (its last reader was H2(k - 1) of
      // the trailing waves, one barrier ago); waited for at the end of this wave's H1(k + 1)
      const int nb = cur == 0 ? 2 : cur - 1;   // (cur + 2) % 3
      MS_T(V1);
      if (e_ + 2 < t_end) X3_STAGE_P(mt_nn, nb);
#ifdef MS_TIMING
      tdma += __builtin_amdgcn_s_memtime() - V1;
#endif
      cur = cur == 2 ? 0 : cur + 1;
    } else {
      cur ^= 1;
    }
    mt_cur = mt_nxt;
    mt_nxt = mt_nn;
    on_cur = on_nxt;
  }
  if (PP && !grp) X3_BAR();   // the leading waves wait out the last half step of the trailing ones
#undef X3_TILE
#undef X3_FLAG
#ifdef MS_TIMING
  if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && tid == 0) {
    ms_dbg[PASS][0] = tg1;
    ms_dbg[PASS][1] = tg2;
    ms_dbg[PASS][2] = tb0;
    ms_dbg[PASS][3] = __builtin_amdgcn_s_memtime() - tall;
    ms_dbg[PASS][4] = t_end - t_begin;
    ms_dbg[PASS][5] = tew;
    ms_dbg[PASS][6] = tdma;
    ms_dbg[PASS][7] = tb1;
  }
#endif
  const int ir = i0 + col;
  if (wave_on && ir < N) {
    float* o = opart + (((size_t)b * S + slice) * N + ir) * MS_D;
#pragma unroll
    for (int fb = 0; fb < 4; ++fb)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(o + fb * 32 + 8 * g + 4 * h) =
            make_float4(acc_o[fb][4 * g], acc_o[fb][4 * g + 1], acc_o[fb][4 * g + 2],
                        acc_o[fb][4 * g + 3]);
  }
  if (PASS == 0) {
    rsum += __shfl_xor(rsum, 32, 64);
    if (wave_on && h == 0 && ir < N) rpart[((size_t)b * S + slice) * N + ir] = rsum;
  }
  if (!flat || e_lo >= e_hi) break;
  ++fblk;
  while (offs[fblk + 1] <= e_lo) ++fblk;  // empty lists
  }
  if (lane == 0 && nexec) atomicAdd(&pn_ms3_exec[PASS], (unsigned long long)nexec);
}

// executed tile pairs of the forward / row / column launches since the last call (read and cleared;
// synchronises with the device)
extern "C" int pn_meanshift_x3_exec_tiles(unsigned long long* out3) {
  PN_CHECK_ARG(out3, "pn_meanshift_x3_exec_tiles: null pointer");
  const unsigned long long zero[3] = {0, 0, 0};
  PN_CHECK_HIP(hipMemcpyFromSymbol(out3, HIP_SYMBOL(pn_ms3_exec), sizeof(zero)));
  PN_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(pn_ms3_exec), zero, sizeof(zero)));
  return PN_OK;
}

// (A row pass with the tile split between the two waves of a SIMD was built and measured in round 4 —
// 1.96 against 1.47 ms per launch of 4 shapes, profiles/r04_rows2_ab.txt — and is not part of the
// library any more; its source is in the history: csrc/meanshift_rows2.h at commit 99c3e53.)

extern "C" size_t pn_meanshift_x3_image_bytes(int B, int N) {
  const int Np = (int)pn_align_up(N, 64);
  return (size_t)B * (Np / 32) * X3_IMG_U4 * 16;
}

// x (B,N,D) -> its tile-image array (pn_meanshift_x3_image_bytes(B,N) bytes)
extern "C" int pn_meanshift_x3_split_f32(const float* x, int B, int N, int D, void* img, void* stream) {
  PN_CHECK_ARG(x && img && B > 0 && N > 0, "pn_meanshift_x3_split_f32: bad arguments");
  PN_CHECK_ARG(D == MS_D, "pn_meanshift: embedding size %d unsupported (built for %d)", D, MS_D);
  const int ntiles = (int)pn_align_up(N, 64) / 32;
  hipLaunchKernelGGL(pn_ms3_split_kernel, dim3(ntiles, B), dim3(256), 0, (hipStream_t)stream, x, N, ntiles,
                     (u32x4*)img);
  PN_CHECK_LAUNCH();
  return PN_OK;
}


// ---- block-sparse plan ------------------------------------------------------------------
// K_ij = exp((q_i . x_j - 1) / b^2) decays fast on a clustered embedding: most (row block, tile)
// pairs contribute less than rel_eps (1e-6 forward-only, 1e-9 under a dense backward: mean_shift.py) of the
// SMALLEST row sum of the block and can be skipped without touching the fp32 result.  The test is rigorous, from bounding caps on the unit sphere (two per
// tile, see pn_ms3_tileinfo_kernel; centre c = normalised mean of the cap's rows, angular radius
// rho = max angle to it; below "tile" reads "cap", and a tile pair is kept when any of its 2 x 2 cap
// pairs is):
//   any pair (q in tile A, x in tile B):  cos(min(pi, th + rA + rB)) <= q.x <= cos(max(0, th - rA - rB)),
//   th = angle(c_A, c_B).
// For a row tile A of the resident side, L_A = max_B cos(th + rA + rB) bounds every row's BEST
// dot product from below; every data cap B contributes at least n_B exp((Lo_AB - 1) / b^2) to every
// row sum of A (n_B rows, Lo_AB = cos(th + rA + rB) <= every dot product of the pair), hence
//   r_i >= exp((L_A - 1) / b^2) R_A,   R_A = sum_B n_B exp((Lo_AB - L_A) / b^2)  (>= 1: the cap attaining L_A).
// What may be dropped for A is any set D of caps whose terms together stay below rel_eps of that:
//   sum_{B in D} n_B exp((U_AB - L_A) / b^2) <= rel_eps R_A.
// Round 4: D = the caps with U_AB < t_A for the LARGEST threshold t_A that satisfies this (found per
// cap A by bisection over the row of U values, pn_ms3_thr_kernel) — the mass actually dropped, not
// "all N points at the bound of the nearest dropped cap" against a row sum of ONE term (rounds 2-3:
// U_AB >= L_A - b^2 log(N / rel_eps)).  On the benchmark's embedding the plans keep 0.51 instead of
// 0.70 of the tile pairs at the same rel_eps (a row of a cluster has hundreds of terms near its
// best one: R_A ~ 200; tools/plan_mass_probe.py, profiles/r04_plan_eps_probe.txt).  The cap that
// attains L_A always passes, so no row sum can vanish.  pairs[tQ][tX] holds the
// predicate; the lists hold, per resident block of every pass, the streamed tiles with at least
// one pair set (pass 0 / 1: blocks of 8 / 4 q tiles against x tiles; pass 2: blocks of 8 x tiles
// against q tiles).
#define X3_PLAN_SLACK 1e-3f
// The cap geometry comes from fp32 dot products of 128 terms: |computed - exact| <= X3_DOT_ERR for
// unit rows ((D + 2) * 2^-23, worst case; the normalisation error of a centre is inside it).  acos
// amplifies that error without bound near angle 0 (d -> sqrt(2 d)), so the error is applied to the
// ARGUMENT, on the side that keeps the bound: a radius is acos(dot - err) >= the true angle, an
// upper bound of a centre angle acos(dot - err), a lower bound acos(dot + err) (0 when >= 1).
#define X3_DOT_ERR 1.6e-5f
// rounding of the cosines of (centre angle +- radii) formed from the dot product, the sines and cosines of the
// radii and one square root (pn_ms3_pairs_kernel): a dozen operations on values <= 1
#define X3_TRIG_ERR 2e-6f
// Backward passes reuse the forward plan of their iteration.  The terms they drop are the same
// kernel values times (q.x - 1) / b^2 factors, so the dropped share of a gradient row is bounded by
// rel_eps / b^2, not rel_eps.  The callers that run these dense backward passes therefore plan with
// rel_eps = 1e-9 (mean_shift.PLAN_REL_EPS_DENSE_BWD): 1e-7 at b = 0.1 (fp32 rounding), 1e-4 at the 0.003
// floor of the bandwidth clamp (src/mean_shift.py:34) — a bandwidth at which a row sees only itself.  The
// forward-only plans of the training path (1e-6) are never seen by a backward pass: that path's gradient
// runs through the centre rows alone, dense and exact (meanshift_rows.hip).

// Two bounding caps per 32-row tile of z (B,N,D); lane = channels (lane, lane + 64).
// A tile of the locality order often straddles two regions of the sphere (the end of one cell and
// the start of the next): one cap around all 32 rows would then be wide enough to meet every
// other tile.  The rows are dealt to two seeds (s1 = the row farthest from the mean direction,
// s2 = the row farthest from s1; a row goes with the seed it has the larger dot product with) and
// each group gets its own cap.  cen (B,ntiles,2,D), rho (B,ntiles,2); rho < 0: empty group;
// cnt (B,ntiles,2): the number of rows of each group (the plan weighs the caps of the data with them).
// (Measured on the cfg5 embedding: active tile pairs 0.29 -> 0.23, visited list entries of the
// row pass 0.40 -> 0.28 of all.)
// (four waves per tile, eight rows each: the dot products of a round are wave-wide sums, and one
// wave per tile left the SIMDs with a single wave of serial reductions: 53 us per call)
// The caps of ONE tile from its rows in registers: this wave's rows 8 wave .. 8 wave + 7, channels (lane,
// lane + 64), rows >= cnt zero; `rows`: the tile's rows again, MS_D floats apart, for the two seed rows (global
// memory, or LDS when the caller produced the rows itself).  All 256 threads of the workgroup call it.
__device__ static inline void x3_tile_caps(const float (&z0)[8], const float (&z1)[8], int cnt,
                                           const float* __restrict__ rows, float* __restrict__ co,
                                           float* __restrict__ ro, float* __restrict__ no) {
  __shared__ float part[4][2][MS_D];
  __shared__ float sd[3][32];
  __shared__ float smin[4][2];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  // dot products of the rows with a vector (v0, v1) -> sd[slot][0..31]
#define X3_TI_DOTS(SLOT, V0, V1)                                                    \
  {                                                                                 \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                 \
      const float d_ = pn_wave_sum(z0[i] * (V0) + z1[i] * (V1));                    \
      if (lane == 0) sd[SLOT][8 * wave + i] = d_;                                   \
    }                                                                               \
    __syncthreads();                                                                \
  }
  float p0 = 0.f, p1 = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    p0 += z0[i];
    p1 += z1[i];
  }
  part[wave][0][lane] = p0;
  part[wave][0][lane + 64] = p1;
  __syncthreads();
  const float m0 = (part[0][0][lane] + part[1][0][lane]) + (part[2][0][lane] + part[3][0][lane]);
  const float m1 = (part[0][0][lane + 64] + part[1][0][lane + 64]) + (part[2][0][lane + 64] + part[3][0][lane + 64]);
  // seeds: s1 = the row farthest from the mean direction, s2 = the row farthest from s1 (ties -> first row)
  X3_TI_DOTS(0, m0, m1);
  int ia = 0, ib = 0;
  float best = 3.4e38f;
  for (int j = 0; j < cnt; ++j)
    if (sd[0][j] < best) {
      best = sd[0][j];
      ia = j;
    }
  const float a0 = rows[(size_t)ia * MS_D + lane], a1 = rows[(size_t)ia * MS_D + lane + 64];
  X3_TI_DOTS(1, a0, a1);
  best = 3.4e38f;
  for (int j = 0; j < cnt; ++j)
    if (sd[1][j] < best) {
      best = sd[1][j];
      ib = j;
    }
  const float b0 = rows[(size_t)ib * MS_D + lane], b1 = rows[(size_t)ib * MS_D + lane + 64];
  X3_TI_DOTS(2, b0, b1);
  unsigned second = 0;   // bit j: row j goes with s2 (larger dot product)
  for (int j = 0; j < cnt; ++j) second |= sd[2][j] > sd[1][j] ? 1u << j : 0u;
  float s[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int g = (int)((second >> (8 * wave + i)) & 1u);   // rows >= cnt hold zeros
    s[0][0] += g ? 0.f : z0[i];
    s[0][1] += g ? 0.f : z1[i];
    s[1][0] += g ? z0[i] : 0.f;
    s[1][1] += g ? z1[i] : 0.f;
  }
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    part[wave][g][lane] = s[g][0];
    part[wave][g][lane + 64] = s[g][1];
  }
  __syncthreads();
  float r[2], c[2][2];
  bool ok[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const float t0 = (part[0][g][lane] + part[1][g][lane]) + (part[2][g][lane] + part[3][g][lane]);
    const float t1 = (part[0][g][lane + 64] + part[1][g][lane + 64]) + (part[2][g][lane + 64] + part[3][g][lane + 64]);
    const float nn = sqrtf(pn_wave_sum(t0 * t0 + t1 * t1));
    const int members = g ? __popc(second) : cnt - __popc(second);
    ok[g] = nn > 1e-6f;
    c[g][0] = ok[g] ? t0 / nn : 0.f;
    c[g][1] = ok[g] ? t1 / nn : 0.f;
    r[g] = members == 0 ? -1.f : 3.2f;   // degenerate group (rows cancel): interacts with everything
  }
  float mn[2] = {1.f, 1.f};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int j = 8 * wave + i;
    const int g = (int)((second >> j) & 1u);
    const float d = pn_wave_sum(z0[i] * (g ? c[1][0] : c[0][0]) + z1[i] * (g ? c[1][1] : c[0][1]));
    if (j < cnt) {
      if (g) mn[1] = fminf(mn[1], d); else mn[0] = fminf(mn[0], d);
    }
  }
  if (lane == 0) {
    smin[wave][0] = mn[0];
    smin[wave][1] = mn[1];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const float mg = fminf(fminf(smin[0][g], smin[1][g]), fminf(smin[2][g], smin[3][g]));
      if (r[g] > 0.f && ok[g]) r[g] = acosf(fminf(fmaxf(mg - X3_DOT_ERR, -1.f), 1.f)) + X3_PLAN_SLACK;
      co[g * MS_D + lane] = c[g][0];
      co[g * MS_D + lane + 64] = c[g][1];
    }
    if (lane == 0) {
      ro[0] = r[0];
      ro[1] = r[1];
      if (no) {
        no[1] = (float)__popc(second);
        no[0] = (float)(cnt - __popc(second));
      }
    }
  }
#undef X3_TI_DOTS
}

// a padding tile (no rows): interacts with everything (its image rows are zero)
__device__ static inline void x3_tile_caps_empty(float* __restrict__ co, float* __restrict__ ro,
                                                 float* __restrict__ no) {
  const int tid = threadIdx.x;
  if (tid < 2 * MS_D) co[tid] = 0.f;
  if (tid == 0) {
    ro[0] = 3.2f;
    ro[1] = -1.f;
    if (no) no[0] = no[1] = 0.f;
  }
}

__global__ __launch_bounds__(256) void pn_ms3_tileinfo_kernel(const float* __restrict__ z, int N, int ntiles,
                                                              float* __restrict__ cen, float* __restrict__ rho,
                                                              float* __restrict__ cnt_out) {
  const int b = blockIdx.y, t = blockIdx.x, tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const float* zb = z + (size_t)b * N * MS_D;
  const int j0 = t * 32, cnt = min(32, N - j0);
  float* co = cen + ((size_t)b * ntiles + t) * 2 * MS_D;
  float* ro = rho + ((size_t)b * ntiles + t) * 2;
  float* no = cnt_out ? cnt_out + ((size_t)b * ntiles + t) * 2 : nullptr;   // rows of the two caps
  if (cnt <= 0) {
    x3_tile_caps_empty(co, ro, no);
    return;
  }
  float z0[8], z1[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int j = 8 * wave + i;
    z0[i] = j < cnt ? zb[(size_t)(j0 + j) * MS_D + lane] : 0.f;
    z1[i] = j < cnt ? zb[(size_t)(j0 + j) * MS_D + lane + 64] : 0.f;
  }
  x3_tile_caps(z0, z1, cnt, zb + (size_t)j0 * MS_D, co, ro, no);
}

// Bounds of the dot products of every cap pair, single-wave workgroups, one per (32 q caps) x (32 x caps)
// block: the 32 x 32 dot products of the cap centres on the fp32 matrix cores (64 v_mfma_f32_32x32x2_f32;
// the angles need 1e-3, bf16 would not do).  Lane (col, h) feeds k-step m with channel 64 h + m of
// row col (the order of the k-steps is free), i.e. reads one contiguous half row.
//   utab[b][q cap][x cap]  = U,  the upper bound  cos(max(th - rho_q - rho_x, 0))  of the pair's dot products
//   lotab[b][q cap][x cap] = Lo, the lower bound  cos(min(th + rho_q + rho_x, pi)) of ALL of them
//   pm[b][q cap][x block]  = max of Lo over the block's x caps (L_q = its maximum over the blocks)
// (-2: a cap without rows).  pn_ms3_thr_kernel turns a cap's rows of the tables into its drop threshold and
// the predicate of its pairs.
__global__ __launch_bounds__(64) void pn_ms3_pairs_kernel(const float* __restrict__ cenQ,
                                                          const float* __restrict__ rhoQ,
                                                          const float* __restrict__ cenX,
                                                          const float* __restrict__ rhoX, int ntiles,
                                                          float* __restrict__ pm, float* __restrict__ utab,
                                                          float* __restrict__ lotab) {
  const int b = blockIdx.z, qb = blockIdx.y, xb = blockIdx.x, lane = threadIdx.x;
  const int col = lane & 31, h = lane >> 5;
  const int ncap = 2 * ntiles, nxb = gridDim.x;
  const int uq = min(qb * 32 + col, ncap - 1), ux = xb * 32 + col, uxc = min(ux, ncap - 1);
  const float4* aq = reinterpret_cast<const float4*>(cenQ + ((size_t)b * ncap + uq) * MS_D + 64 * h);
  const float4* bx = reinterpret_cast<const float4*>(cenX + ((size_t)b * ncap + uxc) * MS_D + 64 * h);
  float4 av[16], bv[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    av[e] = aq[e];
    bv[e] = bx[e];
  }
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e].x, bv[e].x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e].y, bv[e].y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e].z, bv[e].z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e].w, bv[e].w, acc, 0, 0, 0);
  }
  // D[i = q cap][j = x cap]: lane holds column j = col, rows i = (r & 3) + 8 (r >> 2) + 4 h
  const float rx = ux < ncap ? rhoX[(size_t)b * ncap + uxc] : -1.f;
  // lane col keeps the radius of q cap qb * 32 + col; the rows of a lane fetch it by shuffle
  const int qmine = qb * 32 + col;
  const float rq_mine = qmine < ncap ? rhoQ[(size_t)b * ncap + qmine] : -1.f;
  // cos(th +- (rq + rx)) from the dot product d = cos(th) itself:
  //   cos(th + r) = d cos r - sqrt(1 - d^2) sin r,   cos(th - r) = d cos r + sqrt(1 - d^2) sin r,
  //   cos r = cos rq cos rx - sin rq sin rx,  sin r = sin rq cos rx + cos rq sin rx
  // — two sine / cosine pairs per LANE instead of two arc cosines and two cosines per cap pair (what the sweep
  // spent its 35 us on).  A dozen roundings of values <= 1: the bounds move outwards by X3_TRIG_ERR; the radii
  // already carry X3_PLAN_SLACK.
  float cq_mine, sq_mine, cx, sx;
  sincosf(fmaxf(rq_mine, 0.f), &sq_mine, &cq_mine);
  sincosf(fmaxf(rx, 0.f), &sx, &cx);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int qi = qb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
    const float rq = __shfl(rq_mine, qi - qb * 32, 64);
    const float cq = __shfl(cq_mine, qi - qb * 32, 64), sq = __shfl(sq_mine, qi - qb * 32, 64);
    const float rr = rq + rx;                                  // < pi: both cosines below are monotone in th
    const float cr = cq * cx - sq * sx, sr = sq * cx + cq * sx;
    const bool both = rq >= 0.f && rx >= 0.f;
    // the centre angle from above: d - err (for Lo), from below: d + err (for U)
    const float d_lo = fmaxf(acc[r] - X3_DOT_ERR, -1.f), d_hi = fminf(acc[r] + X3_DOT_ERR, 1.f);
    // Lo = cos(min(th_hi + r, pi)): th_hi + r >= pi  <=>  d_lo <= cos(pi - r) = -cos r
    const float lo_cos = (rr >= 3.14159f || d_lo <= -cr)
                             ? -1.f
                             : fmaxf(d_lo * cr - sqrtf(fmaxf(1.f - d_lo * d_lo, 0.f)) * sr - X3_TRIG_ERR, -1.f);
    // U = cos(max(th_lo - r, 0)): th_lo <= r  <=>  d_hi >= cos r
    const float up_cos = (rr >= 3.14159f || d_hi >= cr)
                             ? 1.f
                             : fminf(d_hi * cr + sqrtf(fmaxf(1.f - d_hi * d_hi, 0.f)) * sr + X3_TRIG_ERR, 1.f);
    float m = rx >= 0.f ? lo_cos : -2.f;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (col == 0 && qi < ncap) pm[((size_t)b * ncap + qi) * nxb + xb] = m;
    if (qi < ncap && ux < ncap) {
      utab[((size_t)b * ncap + qi) * ncap + ux] = both ? up_cos : -2.f;
      lotab[((size_t)b * ncap + qi) * ncap + ux] = both ? lo_cos : -2.f;
    }
  }
}

// Drop threshold of every q cap (one wave per cap): the largest t with
//   sum_{x caps with U < t} n_x exp((U - L) / b^2) <= 0.9 rel_eps R,   R = sum_x n_x exp((Lo - L) / b^2)
// (0.9: the fp32 summation of <= 2 ntiles positive terms on either side), by bisection over the cap's
// rows of the U and Lo tables; L = the cap's lower bound of the best dot product (sweep 0), n_x the
// rows of data cap x.  Caps with U >= t are kept — and since the cap's row of U values is in the wave's
// registers, the wave forms the predicate of its cap pairs itself: one workgroup per q TILE, its two waves the
// tile's two caps; pairs[tQ][tX] = OR over the 2 x 2 cap pairs.  (Round 4 recomputed the dot products of all
// cap pairs in a second sweep of pn_ms3_pairs_kernel for this: 20 us per plan.)
template <int MAXV>   // 64 MAXV >= the number of caps (the cap's rows of bounds live in registers)
__global__ __launch_bounds__(128) void pn_ms3_thr_kernel(const float* __restrict__ utab, const float* __restrict__ lotab,
                                                         const float* __restrict__ pm, const float* __restrict__ rhoQ,
                                                         const float* __restrict__ cntX, const float* __restrict__ bsq,
                                                         int ncap, int nxb, float rel_eps, float* __restrict__ thr,
                                                         unsigned char* __restrict__ pairs) {
  __shared__ unsigned char s_on[MAXV][64];
  const int b = blockIdx.y, tq = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int a = 2 * tq + wave, ntiles = ncap >> 1;
  const size_t row = (size_t)b * ncap + a;
  const int nv = (ncap + 63) / 64;
  float u[MAXV];
#pragma unroll
  for (int v = 0; v < MAXV; ++v) {
    const int i = lane + 64 * v;
    u[v] = (v < nv && i < ncap) ? utab[row * ncap + i] : -2.f;
  }
  float t = 2.f;                  // a cap without rows: its pairs are never set
  if (rhoQ[row] >= 0.f) {
    float L = -2.f;
    for (int e = lane; e < nxb; e += 64) L = fmaxf(L, pm[row * nxb + e]);
    L = pn_wave_max(L);
    if (rel_eps < 0.f) {
      // EXACT pruning (pn_meanshift_x3_nearest_f32: the arg-max of a dot product): every row of the cap has
      // a candidate with dot >= L, so a candidate cap whose upper bound lies below L cannot hold a row's
      // maximum — minus the rounding of the fp32 chains the two values are compared in (each within
      // X3_DOT_ERR of the real dot product of unit rows)
      t = L - 4.f * X3_DOT_ERR;
    } else {
      const float ib = 1.0f / bsq[b];
      float m[MAXV];
      float R = 0.f;
#pragma unroll
      for (int v = 0; v < MAXV; ++v) {
        const int i = lane + 64 * v;
        const bool in = v < nv && i < ncap;
        const float uu = u[v];
        const float ll = in ? lotab[row * ncap + i] : -2.f;
        const float n = in ? (cntX ? cntX[(size_t)b * ncap + i] : 32.f) : 0.f;
        m[v] = uu > -1.5f ? n * __expf(fminf((uu - L) * ib, 80.f)) : 0.f;
        // without counts every non-empty cap is known to hold one row
        R += ll > -1.5f ? (cntX ? n : 1.f) * __expf(fminf((ll - L) * ib, 0.f)) : 0.f;
      }
      R = fmaxf(pn_wave_sum(R), 1.f);
      float lo = -1.5f, hi = L;   // f(lo) = 0 <= budget; the cap attaining L has mass >= its share of R > budget
      const float budget = 0.9f * rel_eps * R;
      // (18 halvings of an interval of at most 2.5: `lo`, the side that is always within the budget, ends within
      // 1e-5 of the largest admissible threshold)
      for (int it = 0; it < 18; ++it) {
        const float tm = 0.5f * (lo + hi);
        float f = 0.f;
#pragma unroll
        for (int v = 0; v < MAXV; ++v) f += u[v] < tm ? m[v] : 0.f;
        f = pn_wave_sum(f);
        if (f <= budget) lo = tm; else hi = tm;
      }
      t = lo;
    }
  }
  if (lane == 0) thr[row] = t;
  // the predicate of the cap's pairs (a cap without rows carries U = -2), OR over the x caps 2j, 2j + 1 of a
  // tile (neighbouring lanes), then over the tile's two q caps (the two waves)
  bool on[MAXV];
#pragma unroll
  for (int v = 0; v < MAXV; ++v) {
    const bool o = u[v] >= t;
    const bool other = __shfl_xor((int)o, 1, 64) != 0;     // (every lane takes part: not behind the ||)
    on[v] = o || other;
    if (wave == 1) s_on[v][lane] = on[v] ? 1 : 0;
  }
  __syncthreads();
  if (wave == 0 && (lane & 1) == 0 && tq < ntiles) {
#pragma unroll
    for (int v = 0; v < MAXV; ++v) {
      const int ux = lane + 64 * v;
      if (v < nv && ux < ncap)
        pairs[((size_t)b * ntiles + tq) * ntiles + (ux >> 1)] = (on[v] || s_on[v][lane] != 0) ? 1 : 0;
    }
  }
}

// compact lists; one wave per resident block: [0,nb0) pass 0, [nb0,nb0+nb1) pass 1, then pass 2
__global__ __launch_bounds__(64) void pn_ms3_lists_kernel(const unsigned char* __restrict__ pairs, int ntiles,
                                                          int nb0, int nb1, int nb2, int* __restrict__ counts,
                                                          int* __restrict__ lists) {
  const int b = blockIdx.y, blk = blockIdx.x, lane = threadIdx.x;
  const int nblk = nb0 + nb1 + nb2;
  int first, width;
  bool by_row;   // resident tiles index the rows (q) of pairs
  if (blk < nb0) { first = blk * 8; width = 8; by_row = true; }
  else if (blk < nb0 + nb1) { first = (blk - nb0) * 4; width = 4; by_row = true; }
  else { first = (blk - nb0 - nb1) * 8; width = 8; by_row = false; }
  const unsigned char* P = pairs + (size_t)b * ntiles * ntiles;
  int* out = lists + ((size_t)b * nblk + blk) * ntiles;
  int n = 0;
  for (int t0 = 0; t0 < ntiles; t0 += 64) {
    const int t = t0 + lane;
    bool on = false;
    if (t < ntiles)
      for (int w = 0; w < width && first + w < ntiles; ++w)
        on |= (by_row ? P[(size_t)(first + w) * ntiles + t] : P[(size_t)t * ntiles + first + w]) != 0;
    const unsigned long long m = __ballot(on);
    if (on) out[n + pn_mbcnt(m)] = t;
    n += __popcll(m);
  }
  if (lane == 0) counts[(size_t)b * nblk + blk] = n;
}

// Exclusive prefix of the list lengths of each pass, lists taken in (batch item, block) order: the
// flat schedule of pn_ms3_kernel cuts this sequence into equal ranges.  One workgroup per pass;
// pass p writes B * nb_p + 1 entries at offs + B * (blocks of the earlier passes) + p.
__global__ __launch_bounds__(256) void pn_ms3_offsets_kernel(const int* __restrict__ counts, int B, int nb0, int nb1,
                                                             int nb2, int* __restrict__ offs) {
  __shared__ int part[256];
  const int pass = blockIdx.x, t = threadIdx.x;
  const int nblk = nb0 + nb1 + nb2;
  const int off = pass == 0 ? 0 : (pass == 1 ? nb0 : nb0 + nb1);
  const int nb = pass == 0 ? nb0 : (pass == 1 ? nb1 : nb2);
  const int n = nb * B;
  int* out = offs + (size_t)off * B + pass;
  const int per = (n + 255) / 256;
  const int e0 = min(n, t * per), e1 = min(n, e0 + per);
  int sum = 0;
  for (int e = e0; e < e1; ++e) {
    const int be = e / nb, re = e - be * nb;
    sum += counts[(size_t)be * nblk + off + re];
  }
  part[t] = sum;
  __syncthreads();
  int before = 0;
  for (int u = 0; u < t; ++u) before += part[u];
  for (int e = e0; e < e1; ++e) {
    const int be = e / nb, re = e - be * nb;
    out[e] = before;
    before += counts[(size_t)be * nblk + off + re];
  }
  if (t == 255) out[n] = before;
}

// the partial results of a flat launch: the fragments of block `blk` went to slices 0 .. count - 1
__device__ inline int x3_flat_slices(const int* __restrict__ offs, int blk, int nbB, int G, int cmin) {
  const int total = offs[nbB];
  const int chunk = max((total + G - 1) / G, cmin);
  const int o0 = offs[blk], o1 = offs[blk + 1];
  return o1 > o0 ? (o1 - 1) / chunk - o0 / chunk + 1 : 0;
}

// pn_ms_combine_fwd_kernel / pn_ms_combine_bwd_kernel for the partial results of flat launches
__global__ __launch_bounds__(256) void pn_ms3_combine_fwd_kernel(
    const float* __restrict__ opart, const float* __restrict__ rpart, const float* __restrict__ q, int N, int S,
    const int* __restrict__ offs, int nbp, int nbB, int G, int cmin, float* __restrict__ y,
    float* __restrict__ rsum, float* __restrict__ unorm) {
  const int b = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + wave;
  if (i >= N) return;
  const int ns = x3_flat_slices(offs, b * nbp + i / (32 * X3_WAVES(0)), nbB, G, cmin);
  float o0 = 0.f, o1 = 0.f, r = 0.f;
  for (int s = 0; s < ns; ++s) {
    const float* op = opart + (((size_t)b * S + s) * N + i) * MS_D;
    o0 += op[lane];
    o1 += op[lane + 64];
    r += rpart[((size_t)b * S + s) * N + i];
  }
  const float D = 1.0f / r;
  const size_t base = ((size_t)b * N + i) * MS_D;
  const float q0 = q[base + lane], q1 = q[base + lane + 64];
  const float n0 = q0 + (o0 * D - q0), n1 = q1 + (o1 * D - q1);
  const float nn = sqrtf(pn_wave_sum(n0 * n0 + n1 * n1));
  y[base + lane] = n0 / nn;
  y[base + lane + 64] = n1 / nn;
  if (lane == 0) {
    rsum[(size_t)b * N + i] = r;
    unorm[(size_t)b * N + i] = nn;
  }
}

// pn_ms3_combine_fwd_kernel and pn_ms3_tileinfo_kernel of its result in one launch: one workgroup per tile of
// 32 rows, wave w combines rows 8 w .. 8 w + 7 (the same arithmetic per row: y, rsum, unorm are bit-identical),
// keeps them in registers and in LDS, and the workgroup forms the tile's caps from there — the plan of the next
// iteration needs them, and a kernel of its own read the 20 MB iterate back for it (20 + 22 us per iteration).
__global__ __launch_bounds__(256) void pn_ms3_combine_fwd_info_kernel(
    const float* __restrict__ opart, const float* __restrict__ rpart, const float* __restrict__ q, int N, int S,
    const int* __restrict__ offs, int nbp, int nbB, int G, int cmin, float* __restrict__ y,
    float* __restrict__ rsum, float* __restrict__ unorm, int ntiles, float* __restrict__ cen,
    float* __restrict__ rho, float* __restrict__ cnt_out) {
  __shared__ float rows[32][MS_D];
  const int b = blockIdx.y, t = blockIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j0 = t * 32, cnt = min(32, N - j0);
  float* co = cen + ((size_t)b * ntiles + t) * 2 * MS_D;
  float* ro = rho + ((size_t)b * ntiles + t) * 2;
  float* no = cnt_out ? cnt_out + ((size_t)b * ntiles + t) * 2 : nullptr;
  if (cnt <= 0) {
    x3_tile_caps_empty(co, ro, no);
    return;
  }
  const int ns = x3_flat_slices(offs, b * nbp + j0 / (32 * X3_WAVES(0)), nbB, G, cmin);
  float z0[8], z1[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int j = 8 * wave + u, i = j0 + j;
    z0[u] = z1[u] = 0.f;
    if (j < cnt) {
      float o0 = 0.f, o1 = 0.f, r = 0.f;
      for (int s = 0; s < ns; ++s) {
        const float* op = opart + (((size_t)b * S + s) * N + i) * MS_D;
        o0 += op[lane];
        o1 += op[lane + 64];
        r += rpart[((size_t)b * S + s) * N + i];
      }
      const float D = 1.0f / r;
      const size_t base = ((size_t)b * N + i) * MS_D;
      const float q0 = q[base + lane], q1 = q[base + lane + 64];
      const float n0 = q0 + (o0 * D - q0), n1 = q1 + (o1 * D - q1);
      const float nn = sqrtf(pn_wave_sum(n0 * n0 + n1 * n1));
      z0[u] = n0 / nn;
      z1[u] = n1 / nn;
      y[base + lane] = z0[u];
      y[base + lane + 64] = z1[u];
      if (lane == 0) {
        rsum[(size_t)b * N + i] = r;
        unorm[(size_t)b * N + i] = nn;
      }
    }
    rows[j][lane] = z0[u];
    rows[j][lane + 64] = z1[u];
  }
  __syncthreads();
  x3_tile_caps(z0, z1, cnt, &rows[0][0], co, ro, no);
}

__global__ __launch_bounds__(256) void pn_ms3_combine_bwd_kernel(
    const float* __restrict__ opart_q, const float* __restrict__ opart_x, int N, int S,
    const int* __restrict__ offs_q, int nbp_q, const int* __restrict__ offs_x, int nbp_x, int B, int G, int cmin,
    float* __restrict__ gq, float* __restrict__ gx) {
  const int b = blockIdx.y;
  const long long ND4 = (long long)N * MS_D / 4;
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= ND4) return;
  const int i = (int)(e / (MS_D / 4));
  const int nq = x3_flat_slices(offs_q, b * nbp_q + i / (32 * X3_WAVES(1)), B * nbp_q, G, cmin);
  const int nx = x3_flat_slices(offs_x, b * nbp_x + i / (32 * X3_WAVES(2)), B * nbp_x, G, cmin);
  const float4* pq = reinterpret_cast<const float4*>(opart_q) + (size_t)b * S * ND4 + e;
  const float4* px = reinterpret_cast<const float4*>(opart_x) + (size_t)b * S * ND4 + e;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f), c = a;
  for (int s = 0; s < nq; ++s) {
    const float4 u = pq[(size_t)s * ND4];
    a.x += u.x, a.y += u.y, a.z += u.z, a.w += u.w;
  }
  for (int s = 0; s < nx; ++s) {
    const float4 v = px[(size_t)s * ND4];
    c.x += v.x, c.y += v.y, c.z += v.z, c.w += v.w;
  }
  reinterpret_cast<float4*>(gq)[(size_t)b * ND4 + e] = a;
  float4* g = reinterpret_cast<float4*>(gx) + (size_t)b * ND4 + e;
  float4 o = *g;
  o.x += c.x, o.y += c.y, o.z += c.z, o.w += c.w;
  *g = o;
}

// scratch of the plan kernels behind the lists: [pm: 2 nt x nt / 16 floats | thresholds: 2 nt | U and Lo tables: 2 x (2 nt)^2] per item
static size_t x3_plan_scratch_pm(int B, int ntiles) {
  return pn_align_up((size_t)B * 2 * ntiles * pn_cdiv(2 * ntiles, 32) * sizeof(float), 256);
}
static size_t x3_plan_scratch(int B, int ntiles) {
  return x3_plan_scratch_pm(B, ntiles) + pn_align_up((size_t)B * 2 * ntiles * sizeof(float), 256) +
         2 * pn_align_up((size_t)B * 2 * ntiles * 2 * ntiles * sizeof(float), 256);
}
static void x3_plan_layout(int B, int N, int* ntiles, int* nb0, int* nb1, int* nb2, size_t* off_counts,
                           size_t* off_lists, size_t* total) {
  *ntiles = (int)pn_align_up(N, 64) / 32;
  *nb0 = pn_cdiv(N, 256);
  *nb1 = pn_cdiv(N, 128);
  *nb2 = pn_cdiv(N, 256);
  const size_t nblk = (size_t)*nb0 + *nb1 + *nb2;
  *off_counts = pn_align_up((size_t)B * *ntiles * *ntiles, 256);
  // [pairs | counts | offsets (B * nblk + 3) | lists]
  *off_lists = *off_counts + pn_align_up((size_t)B * nblk * 4, 256) + pn_align_up(((size_t)B * nblk + 3) * 4, 256);
  *total = pn_align_up(*off_lists + (size_t)B * nblk * *ntiles * 4, 256) + x3_plan_scratch(B, *ntiles);
}

extern "C" size_t pn_meanshift_x3_plan_bytes(int B, int N) {
  int nt, a, b_, c;
  size_t oc, ol, tot;
  x3_plan_layout(B, N, &nt, &a, &b_, &c, &oc, &ol, &tot);
  return tot;
}
// The leading part of a plan that its consumers (iterations, nearest, statistics) read: pairs, counts,
// offsets, lists.  The rest — two (2 ntiles)^2 fp32 tables and the sweep buffers — is scratch of the plan
// call itself, dead once it returns: a caller that keeps the plans of T iterations places them
// pn_meanshift_x3_plan_core_bytes apart in ONE buffer of T * core + (plan_bytes - core) bytes, so that the
// scratch of plan t overlaps the not-yet-written plans t + 1 ... and ONE scratch region is kept, not T
// (round-4 advisor finding: ~12.6 MB of dead scratch per saved plan at B = 4, N = 10 000).
extern "C" size_t pn_meanshift_x3_plan_core_bytes(int B, int N) {
  int nt, a, b_, c;
  size_t oc, ol, tot;
  x3_plan_layout(B, N, &nt, &a, &b_, &c, &oc, &ol, &tot);
  return tot - x3_plan_scratch(B, nt);
}

// cen (B,ntiles,2,D), rho (B,ntiles,2) with ntiles = align_up(N,64)/32
extern "C" int pn_meanshift_x3_tileinfo_f32(const float* z, int B, int N, int D, float* cen, float* rho, float* cnt,
                                            void* stream) {
  PN_CHECK_ARG(z && cen && rho && B > 0 && N > 0, "pn_meanshift_x3_tileinfo_f32: bad arguments");
  PN_CHECK_ARG(D == MS_D, "pn_meanshift: embedding size %d unsupported (built for %d)", D, MS_D);
  const int ntiles = (int)pn_align_up(N, 64) / 32;
  hipLaunchKernelGGL(pn_ms3_tileinfo_kernel, dim3(ntiles, B), dim3(256), 0, (hipStream_t)stream, z, N, ntiles, cen,
                     rho, cnt);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

// Greedy nearest-neighbour chain over P = 128 cell centres (the locality order of the caller puts
// the cells of the sphere in this sequence: neighbouring cells of the sequence are neighbours on
// the sphere, so the 4- and 8-tile resident blocks that straddle two cells still have similar
// lists; against sorting the cells by a coarse clustering the visited list entries drop by
// 14 % / 19 % / 20 % in the three passes on the cfg5 embedding).
// sim (B,P,P) = dot products of the centres; rank[b][cell] = position in the chain that starts at
// cell 0 and always moves to the most similar unvisited cell (ties -> smaller index).
// One workgroup of P threads per batch item; the matrix sits in LDS (64 KiB).
#define X3_CHAIN_P 128
__global__ __launch_bounds__(X3_CHAIN_P) void pn_ms3_chain_kernel(const float* __restrict__ sim,
                                                                  int* __restrict__ rank) {
  __shared__ float s_sim[X3_CHAIN_P * X3_CHAIN_P];
  __shared__ float s_v[2];
  __shared__ int s_i[2];
  const int b = blockIdx.x, j = threadIdx.x, wave = j >> 6;
  const float* sb = sim + (size_t)b * X3_CHAIN_P * X3_CHAIN_P;
  for (int e = j; e < X3_CHAIN_P * X3_CHAIN_P; e += X3_CHAIN_P) s_sim[e] = sb[e];
  bool used = j == 0;
  if (j == 0) rank[(size_t)b * X3_CHAIN_P] = 0;
  int cur = 0;
  __syncthreads();
  for (int step = 1; step < X3_CHAIN_P; ++step) {
    float v = used ? -__builtin_inff() : s_sim[cur * X3_CHAIN_P + j];
    if (!(v == v)) v = -3.0e38f;   // NaN similarity: last
    int idx = j;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(v, o, 64);
      const int oi = __shfl_xor(idx, o, 64);
      if (ov > v || (ov == v && oi < idx)) {
        v = ov;
        idx = oi;
      }
    }
    if ((j & 63) == 0) {
      s_v[wave] = v;
      s_i[wave] = idx;
    }
    __syncthreads();
    cur = (s_v[1] > s_v[0] || (s_v[1] == s_v[0] && s_i[1] < s_i[0])) ? s_i[1] : s_i[0];
    if (j == cur) {
      used = true;
      rank[(size_t)b * X3_CHAIN_P + j] = step;
    }
    __syncthreads();
  }
}

extern "C" int pn_meanshift_chain_order_f32(const float* sim, int B, int P, int* rank, void* stream) {
  PN_CHECK_ARG(sim && rank && B > 0, "pn_meanshift_chain_order_f32: bad arguments");
  PN_CHECK_ARG(P == X3_CHAIN_P, "pn_meanshift_chain_order_f32: built for %d cells, got %d", X3_CHAIN_P, P);
  hipLaunchKernelGGL(pn_ms3_chain_kernel, dim3(B), dim3(X3_CHAIN_P), 0, (hipStream_t)stream, sim, rank);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

// plan of one iteration from the tile caps of the iterate (Q) and of the data (X); rel_eps: the
// skipped mass relative to the smallest row sum (the caller's choice: 1e-6 / 1e-9, mean_shift.py)
extern "C" int pn_meanshift_x3_plan_f32(const float* cenQ, const float* rhoQ, const float* cenX,
                                        const float* rhoX, const float* cntX, const float* bsq, int B, int N,
                                        float rel_eps, void* plan, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PN_CHECK_ARG(cenQ && rhoQ && cenX && rhoX && bsq && plan && B > 0 && N > 0 && rel_eps != 0.f,
               "pn_meanshift_x3_plan_f32: bad arguments");
  int nt, nb0, nb1, nb2;
  size_t oc, ol, tot;
  x3_plan_layout(B, N, &nt, &nb0, &nb1, &nb2, &oc, &ol, &tot);
  unsigned char* pairs = (unsigned char*)plan;
  int* counts = (int*)((char*)plan + oc);
  int* lists = (int*)((char*)plan + ol);
  PN_CHECK_ARG(2 * nt <= 2048, "pn_meanshift_x3_plan_f32: N=%d (the threshold search holds <= 2048 caps: N <= 32768)", N);
  // (the scratch of the sweeps sits behind the lists)
  char* scratch = (char*)plan + tot - x3_plan_scratch(B, nt);
  float* pm = (float*)scratch;
  float* thr = (float*)(scratch + x3_plan_scratch_pm(B, nt));
  float* utab = (float*)(scratch + x3_plan_scratch_pm(B, nt) + pn_align_up((size_t)B * 2 * nt * sizeof(float), 256));
  float* lotab = utab + pn_align_up((size_t)B * 2 * nt * 2 * nt * sizeof(float), 256) / sizeof(float);
  const dim3 pgrid(pn_cdiv(2 * nt, 32), pn_cdiv(2 * nt, 32), B);
  hipLaunchKernelGGL(pn_ms3_pairs_kernel, pgrid, dim3(64), 0, stream, cenQ, rhoQ, cenX, rhoX, nt, pm, utab, lotab);
  PN_CHECK_LAUNCH();
  // cntX (rows of every data cap, pn_meanshift_x3_tileinfo_f32) may be NULL: one row per non-empty cap in the
  // row-sum bound, 32 in the dropped mass — still rigorous, keeps more pairs
#define X3_THR(MV)                                                                                              \
  hipLaunchKernelGGL(pn_ms3_thr_kernel<MV>, dim3(nt, B), dim3(128), 0, stream, (const float*)utab,              \
                     (const float*)lotab, (const float*)pm, rhoQ, cntX, bsq, 2 * nt, (int)pgrid.x, rel_eps, thr, pairs)
  if (2 * nt <= 512)
    X3_THR(8);
  else if (2 * nt <= 768)       // N <= 12 288: the benchmark's 626 caps
    X3_THR(12);
  else if (2 * nt <= 1024)
    X3_THR(16);
  else
    X3_THR(32);
#undef X3_THR
  PN_CHECK_LAUNCH();
  hipLaunchKernelGGL(pn_ms3_lists_kernel, dim3(nb0 + nb1 + nb2, B), dim3(64), 0, stream, pairs, nt, nb0, nb1, nb2,
                     counts, lists);
  PN_CHECK_LAUNCH();
  int* offs = counts + pn_align_up((size_t)B * (nb0 + nb1 + nb2) * 4, 256) / 4;
  hipLaunchKernelGGL(pn_ms3_offsets_kernel, dim3(3), dim3(256), 0, stream, counts, B, nb0, nb1, nb2, offs);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

// ---- nearest candidate of every query by EXACT pruning (round 4) ---------------------------------
// MeanShift.nms starts from the nearest shifted point of every point: arg-max_j x_i . c_j over all N
// candidates (src/mean_shift.py:146-149), an INDEX, held bit-exact against the oracle: fp32 fma chain
// over the channels in order, ties to the smaller index.  The engine of knn_mfma.hip evaluates all
// N^2 chains on the fp32 matrix cores (1.1 ms per launch of 4 shapes x 10 000 points).  With the points
// in the locality order of the iterations, the caps of the query tiles and of the candidate tiles say
// which tile pairs can hold a maximum at all (pn_ms3_thr_kernel, exact mode: no tolerance involved), and
// the chains of those pairs alone are evaluated here, in the same arithmetic.
// One workgroup per query tile, four waves; wave w takes the kept candidate tiles w, w + 4, ... of the
// tile's list and evaluates each 32 x 32 block of chains on the fp32 matrix cores exactly like the engine
// (v_mfma_f32_32x32x2_f32: k-step m = channels 2m, 2m + 1 — an fma chain over the channels in order): the
// queries are the resident operand (64 registers), the candidate tile goes through the wave's own LDS
// region (rows padded to 129 floats: the column read of a k-step is conflict-free), the next tile's
// rows are in flight while the current one is evaluated.  Every lane keeps the best candidate of ITS
// query among the rows it sees; lanes, halves and waves are combined once at the end.  perm (position ->
// original index, or NULL): results are written at the query's ORIGINAL index and name the candidate's
// ORIGINAL index; ties go to the smaller original index (decided explicitly: the visiting order is free).
#define NR_PAD 129
__global__ __launch_bounds__(256) void pn_ms3_nearest_kernel(const float* __restrict__ xq, const float* __restrict__ xc,
                                                             const unsigned char* __restrict__ pairs,
                                                             const long long* __restrict__ perm, int N, int ntiles,
                                                             long long* __restrict__ nearest) {
  __shared__ float ctw[4][32 * NR_PAD];
  __shared__ float bvw[4][32];
  __shared__ int biw[4][32];
  __shared__ unsigned short tl[2048];          // the kept candidate tiles of this query tile, ascending
  __shared__ int tl_n;
  __shared__ int wcnt[4];
  const int b = blockIdx.y, tq = blockIdx.x, tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int col = lane & 31, h = lane >> 5;
  if (tq * 32 >= N) return;
  // compact the pair row into a list (one coalesced pass, ballot scan per wave, waves in order)
  {
    const unsigned char* __restrict__ prow = pairs + ((size_t)b * ntiles + tq) * ntiles;
    const int nreal = (N + 31) / 32;            // tiles that hold rows
    if (tid == 0) tl_n = 0;
    __syncthreads();
    for (int base = 0; base < nreal; base += 256) {
      const int t = base + tid;
      const bool on = t < nreal && prow[t] != 0;
      const unsigned long long m = __ballot(on);
      if (lane == 0) wcnt[wave] = __popcll(m);
      __syncthreads();
      int off = tl_n;
      for (int w = 0; w < wave; ++w) off += wcnt[w];
      if (on) tl[off + pn_mbcnt(m)] = (unsigned short)t;
      __syncthreads();
      if (tid == 0) tl_n += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
      __syncthreads();
    }
  }
  const int ntk = tl_n;
  const long long* __restrict__ pb = perm ? perm + (size_t)b * N : nullptr;
  const float* __restrict__ xcb = xc + (size_t)b * N * MS_D;
  // resident queries: B[k = h][j = col] of k-step m = channel 2m + h of query tq * 32 + col
  const int qi = tq * 32 + col;
  float bq[MS_D / 2];
  {
    const float* __restrict__ src = xq + ((size_t)b * N + min(qi, N - 1)) * MS_D + h;
#pragma unroll
    for (int m = 0; m < MS_D / 2; ++m) bq[m] = src[2 * m];
  }
  float bestv = -__builtin_inff();
  int bestj = -1;                                // position in the common order
  float* __restrict__ cw = ctw[wave];
  // staging: lane l takes 16-byte chunks l, l + 64, ... of the 32 x 128 tile (chunk c = row c / 32, columns 4 (c % 32)..)
  float4 stage[16];
#define NR_FETCH(T_)                                                                               \
  {                                                                                                \
    _Pragma("unroll") for (int u = 0; u < 16; ++u) {                                               \
      const int c_ = u * 64 + lane;                                                                \
      const int j_ = min((T_) * 32 + (c_ >> 5), N - 1);                                            \
      stage[u] = reinterpret_cast<const float4*>(xcb + (size_t)j_ * MS_D)[c_ & 31];                \
    }                                                                                              \
  }
  if (wave < ntk) NR_FETCH((int)tl[wave]);
  for (int k = wave; k < ntk; k += 4) {
    const int j0 = (int)tl[k] * 32;
    // (wave-local: LDS traffic of a wave is ordered; the fences keep the compiler from reordering)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int c_ = u * 64 + lane;
      float* d_ = cw + (c_ >> 5) * NR_PAD + 4 * (c_ & 31);
      d_[0] = stage[u].x, d_[1] = stage[u].y, d_[2] = stage[u].z, d_[3] = stage[u].w;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (k + 4 < ntk) NR_FETCH((int)tl[k + 4]);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // A[i = col][k = h]: candidate j0 + col, channel 2m + h
    const float* __restrict__ ar = cw + col * NR_PAD + h;
#pragma unroll
    for (int m = 0; m < MS_D / 2; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[2 * m], bq[m], acc, 0, 0, 0);
    // D[i = candidate (r & 3) + 8 (r >> 2) + 4 h][j = query col]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = j0 + (r & 3) + 8 * (r >> 2) + 4 * h;
      const float v = acc[r];
      if (j < N && v >= bestv) {
        bool take = v > bestv || bestj < 0;
        if (!take) take = pb ? pb[j] < pb[bestj] : j < bestj;     // equal values: the smaller ORIGINAL index
        if (take) {
          bestv = v;
          bestj = j;
        }
      }
    }
  }
#undef NR_FETCH
  // the other half of the rows (lane col + 32), then the four waves
  {
    const float ov = __shfl_xor(bestv, 32, 64);
    const int oj = __shfl_xor(bestj, 32, 64);
    bool take = oj >= 0 && (bestj < 0 || ov > bestv);
    if (!take && oj >= 0 && bestj >= 0 && ov == bestv) take = pb ? pb[oj] < pb[bestj] : oj < bestj;
    if (take) {
      bestv = ov;
      bestj = oj;
    }
  }
  if (h == 0) {
    bvw[wave][col] = bestv;
    biw[wave][col] = bestj;
  }
  __syncthreads();
  if (wave == 0 && h == 0 && qi < N) {
    float v = bvw[0][col];
    int ix = biw[0][col];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float ov = bvw[w][col];
      const int oj = biw[w][col];
      bool take = oj >= 0 && (ix < 0 || ov > v);
      if (!take && oj >= 0 && ix >= 0 && ov == v) take = pb ? pb[oj] < pb[ix] : oj < ix;
      if (take) {
        v = ov;
        ix = oj;
      }
    }
    nearest[(size_t)b * N + (pb ? (int)pb[qi] : qi)] = pb ? pb[ix] : (long long)ix;
  }
}

// queries xq and candidates xc (B,N,D) unit rows in ONE common (locality) order with their tile caps
// (pn_meanshift_x3_tileinfo_f32); workspace: pn_meanshift_x3_plan_bytes(B, N).
extern "C" int pn_meanshift_x3_plan_f32(const float* cenQ, const float* rhoQ, const float* cenX,
                                        const float* rhoX, const float* cntX, const float* bsq, int B, int N,
                                        float rel_eps, void* plan, void* stream_);
extern "C" int pn_meanshift_x3_nearest_f32(const float* xq, const float* xc, const float* cenQ, const float* rhoQ,
                                           const float* cenC, const float* rhoC, const int64_t* perm, int B, int N,
                                           int D, int64_t* nearest, void* workspace, size_t workspace_bytes,
                                           void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PN_CHECK_ARG(xq && xc && cenQ && rhoQ && cenC && rhoC && nearest && workspace, "pn_meanshift_x3_nearest_f32: null pointer");
  PN_CHECK_ARG(D == MS_D, "pn_meanshift: embedding size %d unsupported (built for %d)", D, MS_D);
  PN_CHECK_ARG(workspace_bytes >= pn_meanshift_x3_plan_bytes(B, N), "pn_meanshift_x3_nearest_f32: workspace too small");
  // the pair predicate of the exact mode (bsq is not read: any valid pointer)
  const int rc = pn_meanshift_x3_plan_f32(cenQ, rhoQ, cenC, rhoC, nullptr, rhoQ, B, N, -1.0f, workspace, stream_);
  if (rc != PN_OK) return rc;
  const int ntiles = (int)pn_align_up(N, 64) / 32;
  PN_PROF("sel_pruned_argmax", stream);
  hipLaunchKernelGGL(pn_ms3_nearest_kernel, dim3(pn_cdiv(N, 32), B), dim3(256), 0, stream, xq, xc,
                     (const unsigned char*)workspace, (const long long*)perm, N, ntiles, (long long*)nearest);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

static int x3_slices(int B, int N, int ntiles, int blocks_per_cu, int* tps) {
  // blocks_per_cu == 2 selects the 8-wave workgroups (forward, column pass): 256 rows each
  // fill whole rounds of the 256 CUs (one workgroup per CU: the forward runs 8 waves of <= 256
  // registers on one 72 KiB LDS image, the backward passes 4 waves of 512 registers)
  const int rows_per_block = blocks_per_cu == 2 ? 256 : 128;
  const long long rowblocks = (long long)B * pn_cdiv(N, rows_per_block);
  if (ntiles < 16) {
    *tps = ntiles;
    return 1;
  }
  if (const char* e = getenv("PN_MS_SLICES")) {  // developer override: "<fwd>,<bwd>"
    int sf = 0, sb = 0;
    if (sscanf(e, "%d,%d", &sf, &sb) == 2) {
      const int want = blocks_per_cu == 2 ? sf : sb;
      if (want >= 1 && want <= 32 && want <= ntiles) {
        *tps = pn_cdiv(ntiles, want);
        return want;
      }
    }
  }
  int best = 1;
  double best_score = -1.0;
  const int smax = ntiles / 8 < 32 ? ntiles / 8 : 32;
  for (int S = 1; S <= smax; ++S) {
    const int t = pn_cdiv(ntiles, S);
    if (pn_cdiv(ntiles, t) != S) continue;
    const double rounds = (double)(rowblocks * S) / 256.0;  // one workgroup per CU either way
    const double eff = rounds / (double)(long long)(rounds + 0.999999);
    // every slice costs N x 512 B of partial sums written and read back by the combine kernel:
    // at B = 4 three slices run as fast as eight (42.2 vs 43.4 ms per 10 iterations fwd + bwd)
    // with less than half the HBM traffic
    const double score = eff * (double)t / ((double)t + 1.5) - 0.01 * S;
    if (score > best_score) {
      best_score = score;
      best = S;
    }
  }
  *tps = pn_cdiv(ntiles, best);
  return best;
}

// Round-3 schedules (pn_ms3_kernel<PASS, 1>), PN_MS_PINGPONG = 0 off / 1 (default) ping-pong in the
// column pass + DMA pieces between the k-steps in the row pass / 2 also ping-pong in the forward pass.
// Measured on cfg5 (B = 4 x 10 000, `profiles/r03_pingpong_ab.txt`): column pass 18 455 -> 16 442
// cycles per tile of one workgroup, 1.35 -> 1.28 ms per launch; row pass 7 864 -> 7 702 cycles,
// 1.115 -> 1.07 ms; forward pass 8 172 -> 7 888 cycles but no shorter launches (0.73 ms either
// way: the chip clocks to its power budget and hands half of a cycle saving back), hence not the
// default there.  Results are bit-identical in all modes (tests/test_meanshift_gpu.py).  Also
// measured and not kept: two accumulators in the first GEMM of the forward pass and the second
// GEMM of the forward / row pass over pairs of feature blocks (MFMAs alternating between two
// accumulators): second GEMM -6 % in cycles, row-pass launches +4 % longer.
static int x3_pingpong() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("PN_MS_PINGPONG");
    v = e ? atoi(e) : 1;
  }
  return v;
}
#define X3_LAUNCH_PP(PASS, GRID, BLOCK, STREAM, ...)                                                  \
  {                                                                                                   \
    if (x3_pingpong() >= ((PASS) == 0 ? 2 : 1))                                                       \
      hipLaunchKernelGGL((pn_ms3_kernel<PASS, 1>), GRID, BLOCK, 0, STREAM, __VA_ARGS__);              \
    else                                                                                              \
      hipLaunchKernelGGL((pn_ms3_kernel<PASS, 0>), GRID, BLOCK, 0, STREAM, __VA_ARGS__);              \
  }

// Flat launches (block-sparse plan): one workgroup per CU, a multiple of the 8 XCDs.
static int x3_flat_grid() {
  static int g = 0;
  if (!g) {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess)
      (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (const char* e = getenv("PN_MS_FLAT_G")) cus = atoi(e);   // developer override
    g = cus >= 8 ? cus & ~7 : 8;
  }
  return g;
}

// the plan's views + the flat schedule's parameters; flat needs >= 3 partial-result slices in the
// scratch (a list of <= ntiles entries cut at multiples of cmin >= ntiles / (smax - 2) has at most
// smax - 1 fragments)
struct X3Plan {
  const unsigned char* pairs;
  const int *counts, *lists, *offs;
  int nb0, nb1, nb2, nblk, G, cmin, smax;
  bool flat;
};
static X3Plan x3_plan_view(const void* plan, int B, int N, int ntiles) {
  X3Plan v;
  int pnt;
  size_t oc, ol, ptot;
  x3_plan_layout(B, N, &pnt, &v.nb0, &v.nb1, &v.nb2, &oc, &ol, &ptot);
  v.nblk = v.nb0 + v.nb1 + v.nb2;
  v.smax = pn_meanshift_slices(B, N);
  v.flat = plan != nullptr && v.smax >= 3;
  v.pairs = v.flat ? (const unsigned char*)plan : nullptr;
  v.counts = v.flat ? (const int*)((const char*)plan + oc) : nullptr;
  v.lists = v.flat ? (const int*)((const char*)plan + ol) : nullptr;
  v.offs = v.flat ? v.counts + pn_align_up((size_t)B * v.nblk * 4, 256) / 4 : nullptr;
  v.G = x3_flat_grid();
  v.cmin = v.flat ? pn_cdiv(ntiles, v.smax - 2) : 0;
  return v;
}

// One forward iteration on the bf16 x 3 path: same contract as pn_meanshift_iter_fwd_f32 with the
// tile images of x (pn_meanshift_x3_split_f32) in place of x / xt.
extern "C" int pn_meanshift_x3_iter_fwd_plan_f32(const float* q, const void* img_x, const float* bsq, int B,
                                                 int N, int D, float* opart, float* rpart, float* y,
                                                 float* rsum, float* unorm, const void* plan, void* stream_);
extern "C" int pn_meanshift_x3_iter_fwd_f32(const float* q, const void* img_x, const float* bsq, int B,
                                            int N, int D, float* opart, float* rpart, float* y,
                                            float* rsum, float* unorm, void* stream_) {
  return pn_meanshift_x3_iter_fwd_plan_f32(q, img_x, bsq, B, N, D, opart, rpart, y, rsum, unorm, nullptr, stream_);
}

// The same with a block-sparse plan (pn_meanshift_x3_plan_f32 of THIS q against x; NULL = dense).
extern "C" int pn_meanshift_x3_iter_fwd_info_f32(const float* q, const void* img_x, const float* bsq, int B,
                                                 int N, int D, float* opart, float* rpart, float* y,
                                                 float* rsum, float* unorm, const void* plan, float* cen,
                                                 float* rho, float* cnt, void* stream_);
extern "C" int pn_meanshift_x3_iter_fwd_plan_f32(const float* q, const void* img_x, const float* bsq, int B,
                                                 int N, int D, float* opart, float* rpart, float* y,
                                                 float* rsum, float* unorm, const void* plan, void* stream_) {
  return pn_meanshift_x3_iter_fwd_info_f32(q, img_x, bsq, B, N, D, opart, rpart, y, rsum, unorm, plan, nullptr,
                                           nullptr, nullptr, stream_);
}

// ... and, when cen / rho are given, the bounding caps of the result's tiles (pn_meanshift_x3_tileinfo_f32 of y:
// cen (B,ntiles,2,D), rho (B,ntiles,2), cnt (B,ntiles,2) or NULL) — what the plan of the NEXT iteration takes as
// its q caps; with a plan they come out of the launch that combines the partial results.
extern "C" int pn_meanshift_x3_iter_fwd_info_f32(const float* q, const void* img_x, const float* bsq, int B,
                                                 int N, int D, float* opart, float* rpart, float* y,
                                                 float* rsum, float* unorm, const void* plan, float* cen,
                                                 float* rho, float* cnt, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PN_CHECK_ARG((cen == nullptr) == (rho == nullptr), "pn_meanshift_x3_iter_fwd_info_f32: cen and rho go together");
  PN_CHECK_ARG(q && img_x && bsq && opart && rpart && y && rsum && unorm,
               "pn_meanshift_x3_iter_fwd_f32: null pointer");
  PN_CHECK_ARG(D == MS_D, "pn_meanshift: embedding size %d unsupported (built for %d)", D, MS_D);
  PN_CHECK_ARG(B > 0 && N > 0, "pn_meanshift_x3_iter_fwd_f32: empty input");
  const int ntiles = (int)pn_align_up(N, 64) / 32;
  const X3Plan pv = x3_plan_view(plan, B, N, ntiles);
  if (pv.flat) {
    {
      PN_PROF("meanshift_fwd", stream);
      X3_LAUNCH_PP(0, dim3(pv.G), dim3(512), stream, q, nullptr, (const u32x4*)img_x,
                         nullptr, nullptr, nullptr, bsq, N, ntiles, 0, opart, rpart, pv.pairs, pv.counts, pv.lists,
                         pv.offs, pv.nblk, 0, pv.nb0, B * pv.nb0, pv.cmin, pv.smax);
    }
    PN_CHECK_LAUNCH();
    if (cen)
      hipLaunchKernelGGL(pn_ms3_combine_fwd_info_kernel, dim3(ntiles, B), dim3(256), 0, stream, opart, rpart, q, N,
                         pv.smax, pv.offs, pv.nb0, B * pv.nb0, pv.G, pv.cmin, y, rsum, unorm, ntiles, cen, rho, cnt);
    else
      hipLaunchKernelGGL(pn_ms3_combine_fwd_kernel, dim3(pn_cdiv(N, 4), B), dim3(256), 0, stream, opart, rpart, q, N,
                         pv.smax, pv.offs, pv.nb0, B * pv.nb0, pv.G, pv.cmin, y, rsum, unorm);
    PN_CHECK_LAUNCH();
    return PN_OK;
  }
  int tps;
  int S = x3_slices(B, N, ntiles, 2, &tps);
  if (S > pv.smax) {  // the scratch is sized for this many slices
    S = pv.smax;
    tps = pn_cdiv(ntiles, S);
  }
  dim3 grid(S, pn_cdiv(N, 256), B);
  {
    PN_PROF("meanshift_fwd", stream);
    X3_LAUNCH_PP(0, grid, dim3(512), stream, q, nullptr, (const u32x4*)img_x,
                       nullptr, nullptr, nullptr, bsq, N, ntiles, tps, opart, rpart, nullptr, nullptr, nullptr,
                       nullptr, 0, 0, 0, 0, 0, 0);
  }
  PN_CHECK_LAUNCH();
  hipLaunchKernelGGL(pn_ms_combine_fwd_kernel, dim3(pn_cdiv(N, 4), B), dim3(256), 0, stream, opart,
                     rpart, q, N, S, y, rsum, unorm);
  PN_CHECK_LAUNCH();
  if (cen) {
    hipLaunchKernelGGL(pn_ms3_tileinfo_kernel, dim3(ntiles, B), dim3(256), 0, stream, y, N, ntiles, cen, rho, cnt);
    PN_CHECK_LAUNCH();
  }
  return PN_OK;
}

// Backward of one iteration on the bf16 x 3 path: same contract as pn_meanshift_iter_bwd_f32 with
// the tile images of x in place of xt and two scratch image arrays (images of q and gu, each
// pn_meanshift_x3_image_bytes(B,N) bytes) in place of (qt, gut).  x itself is still read as the
// resident operand of the column pass; go is not needed (the column pass folds 1/r into K).
extern "C" int pn_meanshift_x3_iter_bwd_plan_f32(const float* gy, const float* y, const float* q,
                                                 const float* x, const void* img_x, const float* rsum,
                                                 const float* unorm, const float* bsq, int B, int N, int D,
                                                 float* gu, float* cs, void* img_q, void* img_gu,
                                                 float* opart_q, float* opart_x, float* gq, float* gx,
                                                 const void* plan, void* stream_);
extern "C" int pn_meanshift_x3_iter_bwd_f32(const float* gy, const float* y, const float* q,
                                            const float* x, const void* img_x, const float* rsum,
                                            const float* unorm, const float* bsq, int B, int N, int D,
                                            float* gu, float* cs, void* img_q, void* img_gu,
                                            float* opart_q, float* opart_x, float* gq, float* gx,
                                            void* stream_) {
  return pn_meanshift_x3_iter_bwd_plan_f32(gy, y, q, x, img_x, rsum, unorm, bsq, B, N, D, gu, cs, img_q, img_gu,
                                           opart_q, opart_x, gq, gx, nullptr, stream_);
}

// The same with the plan the forward call of this iteration used (NULL = dense).
extern "C" int pn_meanshift_x3_iter_bwd_plan_f32(const float* gy, const float* y, const float* q,
                                                 const float* x, const void* img_x, const float* rsum,
                                                 const float* unorm, const float* bsq, int B, int N, int D,
                                                 float* gu, float* cs, void* img_q, void* img_gu,
                                                 float* opart_q, float* opart_x, float* gq, float* gx,
                                                 const void* plan, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PN_CHECK_ARG(gy && y && q && x && img_x && rsum && unorm && bsq && gu && cs && img_q && img_gu &&
                   opart_q && opart_x && gq && gx,
               "pn_meanshift_x3_iter_bwd_f32: null pointer");
  PN_CHECK_ARG(D == MS_D, "pn_meanshift: embedding size %d unsupported (built for %d)", D, MS_D);
  const int ntiles = (int)pn_align_up(N, 64) / 32;
  const X3Plan pv = x3_plan_view(plan, B, N, ntiles);
  float* alpha = cs + (size_t)B * N;
  hipLaunchKernelGGL(pn_ms3_prologue_bwd_kernel, dim3(ntiles, B), dim3(256), 0, stream, gy, y, q, rsum, unorm, bsq, N,
                     ntiles, gu, cs, alpha, (u32x4*)img_q, (u32x4*)img_gu);
  PN_CHECK_LAUNCH();
  const long long ND4 = (long long)N * MS_D / 4;
  if (pv.flat) {
    const int* offs_q = pv.offs + (size_t)B * pv.nb0 + 1;
    const int* offs_x = pv.offs + (size_t)B * (pv.nb0 + pv.nb1) + 2;
    {
      PN_PROF("meanshift_bwd_rows", stream);
      X3_LAUNCH_PP(1, dim3(pv.G), dim3(64 * X3_WAVES(1)), stream, q, (const float*)gu,
                         (const u32x4*)img_x, nullptr, (const float*)cs, (const float*)alpha, bsq, N, ntiles, 0,
                         opart_q, nullptr, pv.pairs, pv.counts, pv.lists, offs_q, pv.nblk, pv.nb0, pv.nb1,
                         B * pv.nb1, pv.cmin, pv.smax);
    }
    PN_CHECK_LAUNCH();
    {
      PN_PROF("meanshift_bwd_cols", stream);
      X3_LAUNCH_PP(2, dim3(pv.G), dim3(64 * X3_WAVES(2)), stream, x, nullptr,
                         (const u32x4*)img_q, (const u32x4*)img_gu, (const float*)cs, (const float*)alpha, bsq, N,
                         ntiles, 0, opart_x, nullptr, pv.pairs, pv.counts, pv.lists, offs_x, pv.nblk,
                         pv.nb0 + pv.nb1, pv.nb2, B * pv.nb2, pv.cmin, pv.smax);
    }
    PN_CHECK_LAUNCH();
    hipLaunchKernelGGL(pn_ms3_combine_bwd_kernel, dim3(pn_cdiv(ND4, 256), B), dim3(256), 0, stream, opart_q,
                       opart_x, N, pv.smax, offs_q, pv.nb1, offs_x, pv.nb2, B, pv.G, pv.cmin, gq, gx);
    PN_CHECK_LAUNCH();
    return PN_OK;
  }
  int tps, tps2;
  const int smax = pv.smax;
  int S = x3_slices(B, N, ntiles, 1, &tps);    // row pass: 4-wave workgroups
  if (S > smax) {
    S = smax;
    tps = pn_cdiv(ntiles, S);
  }
  int S2 = x3_slices(B, N, ntiles, 2, &tps2);  // column pass: 8-wave workgroups
  if (S2 > smax) {
    S2 = smax;
    tps2 = pn_cdiv(ntiles, S2);
  }
  {
    PN_PROF("meanshift_bwd_rows", stream);
    dim3 grid(S, pn_cdiv(N, 32 * X3_WAVES(1)), B);
    X3_LAUNCH_PP(1, grid, dim3(64 * X3_WAVES(1)), stream, q, (const float*)gu,
                       (const u32x4*)img_x, nullptr, (const float*)cs, (const float*)alpha, bsq, N,
                       ntiles, tps, opart_q, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, 0);
  }
  PN_CHECK_LAUNCH();
  {
    PN_PROF("meanshift_bwd_cols", stream);
    dim3 grid2(S2, pn_cdiv(N, 32 * X3_WAVES(2)), B);
    X3_LAUNCH_PP(2, grid2, dim3(64 * X3_WAVES(2)), stream, x, nullptr,
                       (const u32x4*)img_q, (const u32x4*)img_gu, (const float*)cs,
                       (const float*)alpha, bsq, N, ntiles, tps2, opart_x, nullptr, nullptr, nullptr, nullptr,
                       nullptr, 0, 0, 0, 0, 0, 0);
  }
  PN_CHECK_LAUNCH();
  hipLaunchKernelGGL(pn_ms_combine_bwd_kernel, dim3(pn_cdiv(ND4, 256), B), dim3(256), 0, stream,
                     opart_q, opart_x, ND4, S, S2, gq, gx);
  PN_CHECK_LAUNCH();
  return PN_OK;
}
