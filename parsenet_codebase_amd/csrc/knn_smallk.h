// Small-k kNN graphs in ONE distance pass (k <= 16): the graphs of the SplineNets' edge-conv layers
// (src/model.py:9-22 with k = 10 on a few thousand points, feature widths 3 ... 256).
//
// The two-pass selection of knn_mfma.hip (tile maxima -> threshold -> collect -> sort) exists to avoid
// sorting, per query, more than ~k of N values when k is large (k = 80 of 10 000).  For k <= 16 the k
// best candidates of a lane fit into its registers, every distance is formed ONCE, and the seven
// launches of a layer (prep, image, pass 1, tau, pass 2, final, fallback gate) become two or three
// (prep, this kernel, and a merge when the candidates were sliced over workgroups).
//
// Arithmetic: the engine's — v_mfma_f32_32x32x2_f32 evaluates the oracle's fma chain over the channels
// in order; v = (-xx[j] - (-2 dot)) - xx[i] with every operation rounded once (model.py:14-16).
// Exact values, so the selection needs no margins, no repairs and no fallback.
//
// Selection.  In the accumulator layout a lane owns ONE query (column) and 16 candidates of every
// 32 x 32 tile; lanes l and l + 32 share a query.  Each lane keeps the KK best (value, index) pairs it
// has seen, sorted, in registers.  A candidate that beats the lane's KK-th value is not inserted at
// once — with 64 lanes per wave some lane would insert at almost every candidate and all lanes would
// pay for the insertion network — but APPENDED (a predicated LDS store) to a buffer of the lane; when
// some lane's buffer could overflow in the next tile the wave flushes: every lane inserts its buffered
// candidates in arrival order (strict >, so that equal values keep the smaller index first: a lane
// meets its candidates in increasing index order) and tightens its threshold.  A wave flushes every
// few tiles at the start of its range and almost never later.  At the end the two lanes of a query
// merge their lists (full (value, smaller index) order); one slice writes the indices directly, several
// slices write sorted key lists that pn_knn_smallk_merge_kernel merges.
#pragma once

// buffer entries per lane: a flush is due when a lane holds more than CAP - 16 (a tile appends at most 16);
// how many entries the flushes process in all does not depend on the trigger (measured on a
// host model: 170 +- 10 entries per 157 tiles for triggers of 4 ... 16), so the capacity is chosen for the
// LDS budget: 8 bytes per entry, one more slot per lane as the target of candidates that do not pass.
template <int KSX, int KK, int NW_>
struct KskCfg {
  // tiles per stage: the narrow instances (3 ... 7 channels) form a tile in ~300 cycles, far less than the
  // latency of the DMA that stages the next one — they stage four tiles at a time
  static constexpr int TPS = KSX <= 4 ? 4 : 1;
  // waves per workgroup (32 queries each).  The 256-channel instance holds its 129 query operands in registers
  // and its two tile buffers are 66 KiB: four waves = ONE wave per SIMD, and a single chain of dependent
  // v_mfma_f32_32x32x2_f32 issues every ~82 cycles instead of every 64 (phase timers, tools/probes/ksk_timers.py:
  // 10 500 cycles per tile of 129 k-steps).  Eight waves share the tiles — two per SIMD, 256 registers each —
  // when the segments are long enough to fill the chip with 256-query workgroups (ksk_plan).
  static constexpr int NW = NW_;
  // 65 k-steps: 33.3 + 46 KiB, two workgroups per CU;  129 k-steps, eight waves: 64.5 + 84 KiB, one
  static constexpr int CAP = NW == 8 ? 20 : (KSX == 65 ? 22 : 24);
};

template <int KK>
struct KskList {
  float v[KK];
  int j[KK];
};

// insert (v, j) behind every entry that is >= v (strict >): c_i = v > L.v[i] is monotone in i
template <int KK>
__device__ static inline void ksk_insert(KskList<KK>& L, float v, int j) {
#pragma unroll
  for (int i = KK - 1; i >= 0; --i) {
    const bool ci = v > L.v[i];
    const bool cp = i > 0 ? (v > L.v[i - 1]) : false;
    const float iv = cp ? L.v[i - 1] : v;
    const int ij = cp ? L.j[i - 1] : j;
    L.v[i] = ci ? iv : L.v[i];
    L.j[i] = ci ? ij : L.j[i];
  }
}
// full order: larger value first, then the smaller index (merging lists of different lanes / slices)
template <int KK>
__device__ static inline void ksk_insert_key(KskList<KK>& L, float v, int j) {
#pragma unroll
  for (int i = KK - 1; i >= 0; --i) {
    const bool ci = v > L.v[i] || (v == L.v[i] && j < L.j[i]);
    const bool cp = i > 0 ? (v > L.v[i - 1] || (v == L.v[i - 1] && j < L.j[i - 1])) : false;
    const float iv = cp ? L.v[i - 1] : v;
    const int ij = cp ? L.j[i - 1] : j;
    L.v[i] = ci ? iv : L.v[i];
    L.j[i] = ci ? ij : L.j[i];
  }
}

// The KK-th best value of a QUERY = of the union of the sorted lists of its two lanes (l, l + 32):
//   max( b[KK-1], a[KK-1], max_{i = 1 .. KK-1} min(a[i-1], b[KK-1-i]) )     (a, b descending)
// — the threshold both lanes may use: KK candidates of the query are already at least this good and precede
// everything still to come in index order, so a later candidate has to BEAT it.  A lane's own KK-th value
// is the query's ~2 KK-th: with it the lanes buffered about twice as many candidates in the steady state.
template <int KK>
__device__ static inline float ksk_union_kth(const KskList<KK>& L) {
  float b[KK];
#pragma unroll
  for (int i = 0; i < KK; ++i) b[i] = __shfl_xor(L.v[i], 32, 64);
  float kth = __builtin_fmaxf(L.v[KK - 1], b[KK - 1]);
#pragma unroll
  for (int i = 1; i < KK; ++i) kth = __builtin_fmaxf(kth, __builtin_fminf(L.v[i - 1], b[KK - 1 - i]));
  return kth;
}

// K0 of this path: xp (B, 2 KSX, Np), channel-first in ORIGINAL order: rows 0 .. C-1 the channels, zero rows,
// and as LAST row -|x_j|^2 / 2 (squared norm = the fma chain over the channels, halving is exact): the matrix
// core adds it as the last term of the chain, acc' = fl(dot - |x_j|^2 / 2), and 2 acc' = fl(2 dot - |x_j|^2) is
// the reference's first subtraction (model.py:14-16) exactly — scaling by two commutes with rounding.  PADDED
// columns get -inf there: their acc' is -inf and passes no threshold, the tile loop needs no validity test.
__global__ void pn_knn_smallk_prep_kernel(const float* __restrict__ x, int C, int N, int CPX, int Np,
                                          float* __restrict__ xp) {
  const int b = blockIdx.y;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= Np) return;
  const float* xb = x + (size_t)b * C * N;
  float* xpb = xp + (size_t)b * CPX * Np;
  float acc = 0.f;
  if (j < N) {
    // eight channels at a time: the loads of a group are independent of one another and of the norm's fma
    // chain (one load -> store -> fma per trip left the wave waiting for every load in turn: 67 us for
    // 6 x 256 x 5 000 values)
    int c = 0;
    for (; c + 8 <= C; c += 8) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = xb[(size_t)(c + e) * N + j];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        xpb[(size_t)(c + e) * Np + j] = v[e];
        acc = __builtin_fmaf(v[e], v[e], acc);
      }
    }
    for (; c < C; ++c) {
      const float v = xb[(size_t)c * N + j];
      xpb[(size_t)c * Np + j] = v;
      acc = __builtin_fmaf(v, v, acc);
    }
    for (int c = C; c < CPX - 1; ++c) xpb[(size_t)c * Np + j] = 0.f;
    acc = -0.5f * acc;
  } else {
    for (int c = 0; c < CPX - 1; ++c) xpb[(size_t)c * Np + j] = 0.f;
    acc = -__builtin_inff();
  }
  xpb[(size_t)(CPX - 1) * Np + j] = acc;
}

// grid (slices, blocks of 32 NW queries, B); NW = 4 (or 8: KskCfg) waves, each 32 queries; candidate tiles staged once per
// workgroup with the LDS DMA like pn_knn_mfma_kernel.  S == 1: out (B, N, k) indices (int64, or int32 when
// out32); otherwise lists (B, Np, S, KK) keys.
//
// The tile loop is software pipelined inside the wave: while the matrix core works through the k-steps of
// tile t (a chain of dependent MFMAs, 64 cycles each), the wave's vector instructions select among the 16
// values per lane of tile t - 1.  The selection is branch free — a candidate that does not pass is stored
// to the lane's spare slot — so that a tile is ONE basic block and the rows sit between the MFMAs; only the
// flush is a branch, taken every few tiles.  A row costs six instructions: the test runs on the accumulator
// itself, against a bound a_lo with  fl(2 a - |q|^2) > thr  =>  a > (thr + |q|^2) / 2 >= a_lo  (a value rounds
// above the representable thr only if it IS above it); the buffer keeps the accumulator and the flush forms
// v = fl(2 acc' - |q|^2), the reference's value, for what it inserts (strict >: what the bound let through
// without being above the threshold is dropped there).
#ifdef KSK_TIMERS
// -DKSK_TIMERS (tools/jobs only): shader cycles of wave 0 of every workgroup by phase, summed over the launch
__device__ unsigned long long ksk_timers[8];
#define KT_NOW() ((long long)__builtin_readcyclecounter())
#define KT_ADD(I, T0) kt[I] += KT_NOW() - (T0)
extern "C" int pn_knn_smallk_timers(unsigned long long* host8, int reset) {
  if (hipMemcpyFromSymbol(host8, HIP_SYMBOL(ksk_timers), sizeof(ksk_timers)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(ksk_timers), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#else
#define KT_NOW() 0ll
#define KT_ADD(I, T0) (void)(T0)
#endif

template <int KSX, int KK, int NW>
__global__ __launch_bounds__(64 * NW) void pn_knn_smallk_kernel(const float* __restrict__ xp, int N, int Np,
                                                            int stages_per_slice, int k, u64* __restrict__ lists,
                                                            void* __restrict__ out, int out32) {
  constexpr int CPX = 2 * KSX;
  constexpr int TPS = KskCfg<KSX, KK, NW>::TPS;
  constexpr int TILE = CPX * 32;            // floats of one tile: [CPX rows][32 candidates]
  constexpr int STAGE = TILE * TPS;
  constexpr int CAP = KskCfg<KSX, KK, NW>::CAP;
  constexpr int SLOTS = CAP + 1;
  // dynamic LDS: [2][STAGE] floats of candidate tiles, then per wave [SLOTS][64] accumulators and
  // [SLOTS][64] indices (the two stores of an entry are SLOTS * 256 bytes apart: one ds_write2st64_b32)
  extern __shared__ __attribute__((aligned(16))) float ksk_smem[];
  float (*lds)[STAGE] = reinterpret_cast<float (*)[STAGE]>(ksk_smem);
  float* buf = ksk_smem + 2 * STAGE;
  const int b = blockIdx.z;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int col = lane & 31, h = lane >> 5;
  const int q = (blockIdx.y * NW + wave) * 32 + col;
  const int qcl = q < Np ? q : Np - 1;
  const bool wave_on = (blockIdx.y * NW + wave) * 32 < N;
  const float* __restrict__ xb = xp + (size_t)b * CPX * Np;
  const int ntiles = (N + 31) / 32;
  const int nstages = (ntiles + TPS - 1) / TPS;
  const int S = gridDim.x, slice = blockIdx.x;
  const int s_begin = slice * stages_per_slice;
  const int s_end = min(nstages, s_begin + stages_per_slice);
#ifdef KSK_TIMERS
  long long kt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  const long long kt_start = KT_NOW();

  // resident query operand; the query meets the candidates' norm row with 1 (k = 1 of the last step)
  float bq[KSX];
#pragma unroll
  for (int m = 0; m < KSX; ++m) bq[m] = xb[(size_t)(2 * m + h) * Np + qcl];
  const float xxq = -2.0f * xb[(size_t)(CPX - 1) * Np + qcl];
  if (h == 1) bq[KSX - 1] = 1.0f;
  KskList<KK> L;
#pragma unroll
  for (int i = 0; i < KK; ++i) {
    L.v[i] = -__builtin_inff();
    L.j[i] = 0;
  }
  float a_lo = -__builtin_inff();
  typedef __attribute__((address_space(3))) float ksk_lds_float;
  typedef __attribute__((address_space(3))) int ksk_lds_int;
  ksk_lds_float* const bbase =                                  // accumulator of slot e at bbase[e * 64],
      (ksk_lds_float*)(buf + wave * (2 * SLOTS * 64) + lane);   // its index SLOTS * 64 dwords further
  ksk_lds_float* const dump = bbase + CAP * 64;
  ksk_lds_float* wp = bbase;
  // flush together: a wave that has to flush says so in flag[tile mod 3]; behind the barrier of the tile all
  // waves flush.  (A flush of one wave holds the others at the next barrier, so four flushes at four
  // different tiles cost the workgroup four times what one common flush costs.)
  ksk_lds_int* const flag = (ksk_lds_int*)(buf + NW * (2 * SLOTS * 64));
  if (tid < 3) flag[tid] = 0;   // (three flags: the one cleared during tile t was last read behind barrier t - 2)

  typedef const __attribute__((address_space(1))) void* ks_gptr;
  typedef __attribute__((address_space(3))) void* ks_lptr;
  constexpr int NCHUNK = (TILE + 255) / 256;
  // one wave instruction copies 1 KiB: 8 rows x 32 candidates of one tile
#define KS_STAGE(ST, BUF)                                                                                   \
  {                                                                                                         \
    _Pragma("unroll") for (int tt = 0; tt < TPS; ++tt) {                                                    \
      const int j0s = min(((ST) * TPS + tt) * 32, Np - 32);                                                 \
      _Pragma("unroll") for (int c = 0; c < NCHUNK; ++c) {                                                  \
        if (((c + tt * NCHUNK) & (NW - 1)) == wave) {                                                              \
          const int row = c * 8 + (lane >> 3);                                                              \
          if (row < CPX)                                                                                    \
            __builtin_amdgcn_global_load_lds((ks_gptr)(xb + (size_t)row * Np + j0s + ((lane & 7) << 2)),    \
                                             (ks_lptr)(&lds[BUF][tt * TILE + c * 256]), 16, 0, 0);          \
        }                                                                                                   \
      }                                                                                                     \
    }                                                                                                       \
  }
  // insert what the lanes have buffered, in arrival order (the LDS reads of entry e + 1 are issued before
  // entry e is inserted)
#define KS_FLUSH()                                                                \
  {                                                                               \
    const int cnt_ = (int)(wp - bbase) >> 6;                                      \
    float fa_ = bbase[0];                                                         \
    int fj_ = reinterpret_cast<ksk_lds_int*>(bbase)[SLOTS * 64];                  \
    for (int e = 0; __ballot(e < cnt_) != 0ull; ++e) {                            \
      const float na_ = bbase[(e + 1) * 64];                                      \
      const int nj_ = reinterpret_cast<ksk_lds_int*>(bbase)[(SLOTS + e + 1) * 64]; \
      const float fv_ = e < cnt_ ? __builtin_fmaf(2.0f, fa_, -xxq) : -__builtin_inff(); \
      ksk_insert<KK>(L, fv_, fj_);                                                \
      fa_ = na_;                                                                  \
      fj_ = nj_;                                                                  \
    }                                                                             \
    wp = bbase;                                                                   \
    const float s_ = ksk_union_kth<KK>(L) + xxq;                                  \
    a_lo = __builtin_fmaf(-__builtin_fabsf(s_), 0x1p-22f, 0.5f * s_);             \
  }
  // one row of the previous tile: threshold test on the accumulator, branch-free append
#define KS_ROW(R)                                                                 \
  {                                                                               \
    const bool pass_ = accp[R] > a_lo;                                            \
    ksk_lds_float* w_ = pass_ ? wp : dump;                                        \
    w_[0] = accp[R];                                                              \
    reinterpret_cast<ksk_lds_int*>(w_)[SLOTS * 64] = jp | ((R & 3) + 8 * (R >> 2)); \
    wp = pass_ ? wp + 64 : wp;                                                    \
  }
  int cur = 0;
  if (s_begin < s_end) KS_STAGE(s_begin, 0);
  __syncthreads();
  f32x16 accp;       // accumulators of the previous tile and its first index (+ 4 h)
  int jp = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) accp[r] = -__builtin_inff();    // (nothing passes before the first tile)
  KT_ADD(1, kt_start);
  for (int st = s_begin; st < s_end; ++st) {
    const long long kt_a = KT_NOW();
    if (st + 1 < s_end) KS_STAGE(st + 1, cur ^ 1);
    KT_ADD(2, kt_a);
    const long long kt_b = KT_NOW();
    if (wave_on) {
#pragma unroll
      for (int tt = 0; tt < TPS; ++tt) {
        if (TPS > 1 && st * TPS + tt >= ntiles) break;     // (the last stage of a narrow instance)
        const float* __restrict__ lx = lds[cur] + tt * TILE;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        // A operands one group of k-steps ahead of their MFMAs (an LDS read issued right before its use
        // leaves the matrix core idle for the read's latency after every group)
        constexpr int PER = KSX >= 16 ? KSX / 16 : KSX;
        constexpr int NG = 16 <= KSX ? 16 : 1;
        constexpr int REST = KSX - PER * NG;     // the norm step (and nothing else) of the wide instances
        static_assert(REST <= PER, "the norm step rides in the operand registers of a group");
        // (two waves per SIMD, 256 registers each: no second operand group — the other wave's MFMAs cover the
        // reads)
        constexpr bool AHEAD = NW == 4;
        float a_cur[PER], a_nxt[AHEAD ? PER : 1];
#pragma unroll
        for (int p = 0; p < (AHEAD ? PER : 1); ++p) a_nxt[p] = 0.f;
#pragma unroll
        for (int p = 0; p < PER; ++p) a_cur[p] = lx[(2 * p + h) * 32 + col];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          if (AHEAD) {
            if (g + 1 < NG) {
#pragma unroll
              for (int p = 0; p < PER; ++p) a_nxt[AHEAD ? p : 0] = lx[(2 * ((g + 1) * PER + p) + h) * 32 + col];
            } else {
#pragma unroll
              for (int p = 0; p < REST; ++p) a_nxt[AHEAD ? p : 0] = lx[(2 * (NG * PER + p) + h) * 32 + col];
            }
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int p = 0; p < PER; ++p)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[p], bq[g * PER + p], acc, 0, 0, 0);
          if (KSX >= 16 && AHEAD) KS_ROW(g);
          __builtin_amdgcn_sched_barrier(0);
          if (AHEAD) {
#pragma unroll
            for (int p = 0; p < PER; ++p) a_cur[p] = a_nxt[AHEAD ? p : 0];
          } else if (g + 1 < NG) {
#pragma unroll
            for (int p = 0; p < PER; ++p) a_cur[p] = lx[(2 * ((g + 1) * PER + p) + h) * 32 + col];
          } else {
#pragma unroll
            for (int p = 0; p < REST; ++p) a_cur[p] = lx[(2 * (NG * PER + p) + h) * 32 + col];
          }
        }
#pragma unroll
        for (int p = 0; p < REST; ++p)
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[p], bq[NG * PER + p], acc, 0, 0, 0);
        if (KSX < 16) {
#pragma unroll
          for (int r = 0; r < 16; ++r) KS_ROW(r);
        }
        accp = acc;
        jp = (st * TPS + tt) * 32 + 4 * h;
        if (!AHEAD) {
          // two waves per SIMD: the rows of THIS tile right behind its MFMAs (the other wave's chain runs
          // meanwhile) — no second set of accumulators
#pragma unroll
          for (int r = 0; r < 16; ++r) KS_ROW(r);
#pragma unroll
          for (int r = 0; r < 16; ++r) accp[r] = -__builtin_inff();
        }
        if (TPS > 1) {
          if (__ballot(wp > bbase + (CAP - 16) * 64) != 0ull) KS_FLUSH();
        }
      }
      if (TPS == 1 && __ballot(wp > bbase + (CAP - 16) * 64) != 0ull && lane == 0) flag[st % 3] = 1;
    }
    KT_ADD(3, kt_b);
    const long long kt_c = KT_NOW();
    if (TPS == 1 && tid == 0) flag[(st + 1) % 3] = 0;
    __syncthreads();
    KT_ADD(4, kt_c);
    const long long kt_d = KT_NOW();
    if (TPS == 1 && flag[st % 3] != 0 && wave_on) KS_FLUSH();
    KT_ADD(5, kt_d);
    cur ^= 1;
  }
  const long long kt_e = KT_NOW();
#undef KS_STAGE
  if (!wave_on) return;
  // the last tile's rows
#pragma unroll
  for (int r = 0; r < 16; ++r) KS_ROW(r);
#undef KS_ROW
  KS_FLUSH();
#undef KS_FLUSH
  // the two lanes of a query: lane col takes the list of lane col + 32
  // (all shuffles of the ORIGINAL lists first, the insertions after)
  float pv[KK];
  int pj[KK];
#pragma unroll
  for (int i = 0; i < KK; ++i) {
    pv[i] = __shfl_xor(L.v[i], 32, 64);
    pj[i] = __shfl_xor(L.j[i], 32, 64);
  }
  if (h == 0) {
#pragma unroll
    for (int i = 0; i < KK; ++i) ksk_insert_key<KK>(L, pv[i], pj[i]);
  }
  KT_ADD(6, kt_e);
  KT_ADD(0, kt_start);
#ifdef KSK_TIMERS
  if (tid == 0) {
    for (int i = 0; i < 7; ++i) atomicAdd(&ksk_timers[i], (unsigned long long)kt[i]);
    atomicAdd(&ksk_timers[7], 1ull);
  }
#endif
  if (h != 0 || q >= N) return;
  if (S == 1) {
    if (out32) {
      int* o = (int*)out + ((size_t)b * N + q) * k;
#pragma unroll
      for (int i = 0; i < KK; ++i)
        if (i < k) o[i] = L.j[i];
    } else {
      int64_t* o = (int64_t*)out + ((size_t)b * N + q) * k;
#pragma unroll
      for (int i = 0; i < KK; ++i)
        if (i < k) o[i] = (int64_t)L.j[i];
    }
  } else {
    u64* o = lists + (((size_t)b * Np + q) * S + slice) * KK;
#pragma unroll
    for (int i = 0; i < KK; ++i) o[i] = knn_key(L.v[i], L.j[i]);
  }
}

// merge of the S sorted key lists of a query: one thread per query
template <int KK>
__global__ __launch_bounds__(256) void pn_knn_smallk_merge_kernel(const u64* __restrict__ lists, int N, int Np,
                                                                  int S, int k, void* __restrict__ out,
                                                                  int out32) {
  const int b = blockIdx.y;
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= N) return;
  const u64* __restrict__ l = lists + ((size_t)b * Np + q) * S * KK;
  u64 best[KK];
#pragma unroll
  for (int i = 0; i < KK; ++i) best[i] = l[i];
  for (int s = 1; s < S; ++s) {
    // (the slice's keys first, all loads in flight, then the insertions)
    u64 in_[KK];
#pragma unroll
    for (int e = 0; e < KK; ++e) in_[e] = l[s * KK + e];
#pragma unroll
    for (int e = 0; e < KK; ++e) {
      const u64 key = in_[e];
#pragma unroll
      for (int i = KK - 1; i >= 0; --i) {
        const bool ci = key > best[i];
        const bool cp = i > 0 ? key > best[i - 1] : false;
        const u64 in = cp ? best[i - 1] : key;
        best[i] = ci ? in : best[i];
      }
    }
  }
  if (out32) {
    int* o = (int*)out + ((size_t)b * N + q) * k;
#pragma unroll
    for (int i = 0; i < KK; ++i)
      if (i < k) o[i] = (int)knn_key_index(best[i]);
  } else {
    int64_t* o = (int64_t*)out + ((size_t)b * N + q) * k;
#pragma unroll
    for (int i = 0; i < KK; ++i)
      if (i < k) o[i] = knn_key_index(best[i]);
  }
}

template <int KSX, int KK, int NW>
static size_t ksk_smem_bytes_t() {
  typedef KskCfg<KSX, KK, NW> Cfg;
  return (size_t)2 * (2 * KSX * 32) * Cfg::TPS * 4 + (size_t)NW * 2 * (Cfg::CAP + 1) * 64 * 4 + 16;
}
template <int KK>
static size_t ksk_smem_bytes_k(int ksx, int nw) {
  return ksx == 2 ? ksk_smem_bytes_t<2, KK, 4>() : ksx == 4 ? ksk_smem_bytes_t<4, KK, 4>()
         : ksx == 33 ? ksk_smem_bytes_t<33, KK, 4>() : ksx == 65 ? ksk_smem_bytes_t<65, KK, 4>()
         : (nw == 8 ? ksk_smem_bytes_t<129, KK, 8>() : ksk_smem_bytes_t<129, KK, 4>());
}
static size_t ksk_smem_bytes(int ksx, int kk, int nw) {
  return kk == 10 ? ksk_smem_bytes_k<10>(ksx, nw) : ksk_smem_bytes_k<16>(ksx, nw);
}

struct KskPlan {
  bool ok;
  int ksx, Np, S, stages_per_slice, KK, nw;
  size_t xp, lists, total;
};

static int ksk_level() {
  const char* e = getenv("PN_KNN_SMALLK");
  return e ? atoi(e) : 1;
}

// feature metric, k <= 16
static KskPlan ksk_plan(int mode, int B, int C, int N, int k) {
  KskPlan p;
  memset(&p, 0, sizeof(p));
  if (mode != 0 || k > 16 || N < 32 || C > 256 || !ksk_level()) return p;
  // k-steps including the norm row (the last row of the copy): a spare channel of the last step, or one more
  p.ksx = C <= 3 ? 2 : (C <= 7 ? 4 : (C <= 64 ? 33 : (C <= 128 ? 65 : 129)));
  p.Np = (int)pn_align_up(N, 64);
  p.KK = k <= 10 ? 10 : 16;
  const int tps = p.ksx <= 4 ? 4 : 1;
  const int nstages = pn_cdiv(pn_cdiv(N, 32), tps);
  // waves per workgroup: eight for the 256-channel instance on segments of >= 2 048 points (B = 32 patches of
  // 700 points measured 0.26 against 0.23 ms with eight: 96 workgroups of 256 queries do not fill the chip)
  p.nw = (p.ksx == 129 && p.KK == 10 && N >= 2048) ? 8 : 4;
  const long long wgs = (long long)B * pn_cdiv(N, 32 * p.nw);
  // Slices of the candidate range: whole rounds over the CU slots (two workgroups per CU while the LDS allows,
  // one for the 256-channel instance), against the cold start every slice pays — its first ~50 candidates per
  // lane all pass: ~11 600 cycles of insertions, in units of a stage's duration.
  const long long slots = 256 * (ksk_smem_bytes(p.ksx, p.KK, p.nw) <= 80 * 1024 ? 2 : 1);
  const double stage_cycles = (64.0 * p.ksx + 200.0 < 500.0 ? 500.0 : 64.0 * p.ksx + 200.0) * tps;
  const double cold = 11600.0 / stage_cycles;
  int S = 1;
  double best = 1e30;
  for (int s = 1; s <= 8; ++s) {
    const int sps = pn_cdiv(nstages, s);
    if (s > 1 && sps * tps < 12) break;
    const double cost = (double)pn_cdiv(wgs * pn_cdiv(nstages, sps), slots) * ((double)sps + cold);
    if (cost < best * 0.97) {
      best = cost;
      S = s;
    }
  }
  p.stages_per_slice = pn_cdiv(nstages, S);
  p.S = pn_cdiv(nstages, p.stages_per_slice);
  size_t o = 0;
  auto take = [&](size_t bytes) {
    size_t at = o;
    o += pn_align_up(bytes, 256);
    return at;
  };
  p.xp = take((size_t)B * 2 * p.ksx * p.Np * 4);
  p.lists = take(p.S > 1 ? (size_t)B * p.Np * p.S * p.KK * 8 : 0);
  p.total = o;
  p.ok = true;
  return p;
}

static int ksk_run(const KskPlan& p, const float* x, int B, int C, int N, int k, void* out, int out32, char* base,
                   hipStream_t stream) {
  float* xp = (float*)(base + p.xp);
  u64* lists = (u64*)(base + p.lists);
  {
    PN_PROF("knn_prep", stream);
    // one thread walks ALL channels of its column (the norm is one fma chain): 64-thread workgroups, so that 6 x 5 000
    // columns are 474 workgroups on the 256 CUs instead of 120 (the launch is latency-bound: 0.57 ms of a cfg5 step
    // in twelve such launches)
    hipLaunchKernelGGL(pn_knn_smallk_prep_kernel, dim3(pn_cdiv(p.Np, 64), B), dim3(64), 0, stream, x, C, N,
                       2 * p.ksx, p.Np, xp);
  }
  PN_CHECK_LAUNCH();
  dim3 grid(p.S, pn_cdiv(N, 32 * p.nw), B);
  {
    PN_PROF(p.ksx <= 4 ? "knn_smallk_c4" : (p.ksx == 33 ? "knn_smallk_c64" : "knn_smallk_wide"), stream);
#define KSK_GO(KS, KK_, NW_)                                                                                    \
  {                                                                                                             \
    const size_t smem = ksk_smem_bytes_t<KS, KK_, NW_>();                                                       \
    static unsigned attr_devs = 0;  /* per device: pn_first_on_device */                                        \
    if (pn_first_on_device(&attr_devs)) {                                                                       \
      PN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(pn_knn_smallk_kernel<KS, KK_, NW_>),       \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));                 \
    }                                                                                                           \
    hipLaunchKernelGGL((pn_knn_smallk_kernel<KS, KK_, NW_>), grid, dim3(64 * NW_), smem, stream,                \
                       (const float*)xp, N, p.Np, p.stages_per_slice, k, lists, out, out32);                    \
  }
#define KSK_GO_K(KS, NW_) \
  if (p.KK == 10)         \
    KSK_GO(KS, 10, NW_)   \
  else                    \
    KSK_GO(KS, 16, NW_)
    if (p.ksx == 2) {
      KSK_GO_K(2, 4);
    } else if (p.ksx == 4) {
      KSK_GO_K(4, 4);
    } else if (p.ksx == 33) {
      KSK_GO_K(33, 4);
    } else if (p.ksx == 65) {
      KSK_GO_K(65, 4);
    } else if (p.nw == 8) {
      KSK_GO(129, 10, 8);
    } else {
      KSK_GO_K(129, 4);
    }
#undef KSK_GO_K
#undef KSK_GO
  }
  PN_CHECK_LAUNCH();
  if (p.S > 1) {
    PN_PROF("knn_smallk_merge", stream);
    dim3 g(pn_cdiv(N, 256), B);
    if (p.KK == 10)
      hipLaunchKernelGGL(pn_knn_smallk_merge_kernel<10>, g, dim3(256), 0, stream, (const u64*)lists, N, p.Np, p.S, k,
                         out, out32);
    else
      hipLaunchKernelGGL(pn_knn_smallk_merge_kernel<16>, g, dim3(256), 0, stream, (const u64*)lists, N, p.Np, p.S, k,
                         out, out32);
    PN_CHECK_LAUNCH();
  }
  return PN_OK;
}
