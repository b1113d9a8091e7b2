// Differentiable mean-shift iterations on the unit hypersphere, fused "flash style" on the
// fp32 matrix cores of gfx950 — replaces src/mean_shift.py:45-79 (mean_shift_):
//
//   dist = 2 - 2 * new_X @ X^T ; K = exp(clamp(-dist / b^2 / 2, +-75)) ; D = 1 / sum_j K
//   new_X = new_X + ((K @ X) * D - new_X) ; new_X /= ||new_X||
//
// The reference materialises three N x N fp32 tensors per iteration and keeps them for
// autograd (~12 GB for 10 iterations at N = 10 000).  Here no N x N tensor ever exists:
// a wave owns 32 rows (their 128-d operand resident in VGPRs), streams tiles of 32 columns,
// forms S = rows . cols with v_mfma_f32_32x32x2_f32 (exact fp32 fma chains), applies the
// kernel elementwise on the 16 accumulator values it holds, and immediately contracts the
// result with the 128-d column vectors on the matrix cores again.  The D-layout of the first
// product (lane = one row-operand column, 16 streamed indices per lane) IS the B-operand
// layout of the second one once the contraction index is enumerated as
// (m&3) + 8*(m>>2) + 4*(lane>>5), so no shuffle / LDS round trip sits between the two GEMMs.
// The streamed range is split into slices (blockIdx.y) whose partial sums are added
// afterwards — exp arguments are <= 0 (+ rounding), so no running-max rescaling is needed.
//
// The backward recomputes S instead of storing it (saved per iteration: the iterate, the row
// sums and the pre-normalisation norms):
//   gu = (gy - y (y.gy)) / ||u|| ; c = gu.u ; go = gu / r
//   gs_ij = K_ij * (gu_i.x_j - c_i) / (r_i b^2)          (zero where the clamp is active)
//   gq_i  = sum_j gs_ij x_j                               (PASS 1, rows resident)
//   gX_j += sum_i gs_ij q_i + sum_i K_ij go_i             (PASS 2, columns resident)
#include "common.h"
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MS_D 128
#define MS_KS (MS_D / 2)  // k-steps of the 128-d dot products

// Kernel value, reference: dist = 2 - 2 s ; arg = -dist / b^2 / 2 ; clamp(+-75) ; exp.
// Evaluated as exp2(clamp(-dist * hl)) with hl = (0.5 / b^2) * log2(e) and the hardware exp2
// (v_exp_f32, ~1e-7 relative): one rounding of the argument instead of the reference's two
// divisions (<= 1-2 ulp of arg, i.e. <= |arg| * 1.2e-7 relative on K — far below the 1e-5
// bar); the division/exp library sequences were a third of the kernel's time.
#define MS_LOG2E 1.4426950408889634f
#define MS_LIM2 (75.0f * MS_LOG2E)

// PASS 0: forward          resident rows = Q,  streamed cols = X : out[f][row] += X[col][f] * K
// PASS 1: backward, rows   resident rows = Q, GU; streamed cols = X : out += X[col][f] * gs
// PASS 2: backward, cols   resident cols = X;  streamed rows = Q, GU: out += Q[row][f]*gs + GO[row][f]*K
//
// R  (B,N,D)  point-major resident operand; R1 second resident operand (PASS 1: GU)
// At (B,D,Np) channel-first padded streamed operand for S; At1 for T (PASS 2: GUt; PASS 1 reuses At)
// P0 (B,N,D)  point-major streamed operand of the second GEMM; P1 second one (PASS 2: GO)
// cs, rs      per-row scalars c_i and alpha_i = 1/(r_i b^2) (PASS 1: of the resident row; PASS 2: streamed)
// opart (B,S,N,D), rpart (B,S,N) partial outputs of slice blockIdx.y
#define MS_TILE 4096  // floats per staged array tile (16 KiB): 128 channels x 32 columns

// Cooperative stage of one streamed tile with the LDS DMA (global_load_lds, 16 bytes per lane:
// one wave instruction moves 1 KiB from per-lane global addresses to a lane-linear LDS chunk).
// No VGPRs are spent on staging and the copy of tile t+1 runs under the MFMAs of tile t; the
// __syncthreads() that ends the tile waits for it (vmcnt) before anyone reads the buffer.
typedef const __attribute__((address_space(1))) void* ms_gptr;
typedef __attribute__((address_space(3))) void* ms_lptr;
#define MS_GLDS16(G, L) __builtin_amdgcn_global_load_lds((ms_gptr)(G), (ms_lptr)(L), 16, 0, 0)
#define MS_GLDS4(G, L) __builtin_amdgcn_global_load_lds((ms_gptr)(G), (ms_lptr)(L), 4, 0, 0)
#define MS_STAGE_PIECE(MT, BUF, U)                                                              \
  {                                                                                             \
    const int j0s = (MT) * 32;                                                                  \
    const int q = wave * 4 + (U); /* 1 KiB chunk of the 16 KiB tile */                          \
    /* channel-first tiles [128 ch][32 idx]: chunk q = rows 8q..8q+7 */                         \
    const size_t ga = (size_t)(q * 8 + (lane >> 3)) * Np + j0s + ((lane & 7) << 2);             \
    MS_GLDS16(Atb + ga, &lds[BUF][0][q * 256]);                                                 \
    if (PASS == 2) MS_GLDS16(At1b + ga, &lds[BUF][NARR - 2][q * 256]);                          \
    /* point-major tiles [32 idx][128 feat]: chunk q = rows 2q, 2q+1; rows past N clamp */      \
    const size_t gp = (size_t)min(j0s + q * 2 + (lane >> 5), N - 1) * MS_D + ((lane & 31) << 2); \
    MS_GLDS16(P0b + gp, &lds[BUF][1][q * 256]);                                                 \
    if (PASS == 2) MS_GLDS16(P1b + gp, &lds[BUF][NARR - 1][q * 256]);                           \
  }
#define MS_STAGE_SCALARS(MT, BUF)                                                               \
  if (PASS == 2 && wave == 0) { /* per-row scalars c_i | alpha_i of the tile: 64 floats */      \
    const int jc = min((MT) * 32 + (lane & 31), N - 1);                                         \
    MS_GLDS4((lane < 32 ? cs : rs) + bN + jc, &lds_sc[BUF][0]);                                 \
  }
#define MS_STAGE(MT, BUF)                                                                       \
  {                                                                                             \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) MS_STAGE_PIECE(MT, BUF, u);                   \
    MS_STAGE_SCALARS(MT, BUF);                                                                  \
  }

#ifdef MS_TIMING
// developer build only: per-phase shader-clock totals of wave 0 of workgroup (0,0,0)
__device__ unsigned long long ms_dbg[3][8];
extern "C" int pn_ms_debug_read(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ms_dbg), sizeof(ms_dbg));
}
#define MS_T(V) const unsigned long long V = __builtin_amdgcn_s_memtime()
#else
#define MS_T(V)
#endif

template <int PASS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PASS == 0 ? 2 : 1, PASS == 0 ? 2 : 1))) void pn_ms_kernel(
    const float* __restrict__ R, const float* __restrict__ R1, const float* __restrict__ At,
    const float* __restrict__ At1, const float* __restrict__ P0, const float* __restrict__ P1,
    const float* __restrict__ cs, const float* __restrict__ rs, const float* __restrict__ bsq_,
    int N, int Np, int tiles_per_slice, float* __restrict__ opart, float* __restrict__ rpart) {
  // LDS image of the streamed operands of one 32-index tile, double buffered and shared by the
  // four waves of the block (they own different resident rows but stream the same tiles):
  //   arr 0: At  tile [128 ch][32 idx]      (first GEMM, A operand)
  //   arr 1: P0  tile [32 idx][128 feat]    (second GEMM, A operand)
  //   arr 2/3 (PASS 2): At1 and P1 likewise
  constexpr int NARR = PASS == 2 ? 4 : 2;
  __shared__ __attribute__((aligned(16))) float lds[2][NARR][MS_TILE];
  __shared__ __attribute__((aligned(16))) float lds_sc[2][64];
  const int b = blockIdx.z;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int col = lane & 31, h = lane >> 5;
  // grid.x = slice so that consecutive workgroups (round-robin over the 8 XCDs) differ in the
  // SLICE: with S a multiple of 8 an XCD only ever streams 1/8 of the column range
  const int i0 = (blockIdx.y * 4 + wave) * 32;  // resident block of this wave
  const bool wave_on = i0 < N;                   // idle waves still help staging and barriers
  const int S = gridDim.x, slice = blockIdx.x;
  const int ntiles = Np / 32;
  const int t_begin = slice * tiles_per_slice;
  const int t_end = min(ntiles, t_begin + tiles_per_slice);
  const float bsq = bsq_[b];
  const float hl = (0.5f / bsq) * MS_LOG2E;
  const size_t bN = (size_t)b * N;
  const float* __restrict__ Atb = At + (size_t)b * MS_D * Np;
  const float* __restrict__ At1b = PASS == 2 ? At1 + (size_t)b * MS_D * Np : nullptr;
  const float* __restrict__ P0b = P0 + bN * MS_D;
  const float* __restrict__ P1b = PASS == 2 ? P1 + bN * MS_D : nullptr;

  const int ires = min(i0 + col, N - 1);
  float br[MS_KS], br1[PASS == 1 ? MS_KS : 1];
#pragma unroll
  for (int m = 0; m < MS_KS; ++m) {
    br[m] = R[(bN + ires) * MS_D + 2 * m + h];
    if (PASS == 1) br1[m] = R1[(bN + ires) * MS_D + 2 * m + h];
  }
  float c_res = 0.f, rinv_res = 0.f;
  if (PASS == 1) {
    c_res = cs[bN + ires];
    rinv_res = rs[bN + ires];
  }
  f32x16 acc_o[4];
#pragma unroll
  for (int fb = 0; fb < 4; ++fb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_o[fb][r] = 0.f;
  float rsum = 0.f;

  int cur = 0;
  if (t_begin < t_end) MS_STAGE(t_begin, 0);
  __syncthreads();
#ifdef MS_TIMING
  unsigned long long tg1 = 0, tew = 0, tg2 = 0, tbar = 0, tall = __builtin_amdgcn_s_memtime();
#endif
  for (int mt = t_begin; mt < t_end; ++mt) {
    MS_T(T0);
    const int j0 = mt * 32;
    const bool has_next = mt + 1 < t_end;
    // The DMA of tile t+1 runs under the MFMAs of tile t.  Each global_load_lds costs its wave
    // 60-180 issue cycles, so the pieces are spread over the MFMA groups of the first GEMM (one
    // per 64-cycle MFMA shadow) instead of being issued back to back in front of them.
    if (has_next && !wave_on) MS_STAGE(mt + 1, cur ^ 1);
    if (wave_on) {
      const float* __restrict__ lAt = lds[cur][0];
      const float* __restrict__ lP0 = lds[cur][1];
      const float* __restrict__ lAt1 = PASS == 2 ? lds[cur][2] : nullptr;
      const float* __restrict__ lP1 = PASS == 2 ? lds[cur][3] : nullptr;
      f32x16 s, t;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[r] = 0.f;
        t[r] = 0.f;
      }
      // first GEMM, software pipelined by hand: the LDS reads of group g+1 are issued before the
      // eight MFMAs of group g, so that a dependent MFMA chain never waits on an LDS round trip
      {
        float ac[8], an[8], ac1[PASS == 2 ? 8 : 1], an1[PASS == 2 ? 8 : 1];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          ac[u] = lAt[(2 * u + h) * 32 + col];
          if (PASS == 2) ac1[u] = lAt1[(2 * u + h) * 32 + col];
        }
#pragma unroll
        for (int g = 0; g < MS_KS / 8; ++g) {
          if (g + 1 < MS_KS / 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              an[u] = lAt[(2 * (8 * (g + 1) + u) + h) * 32 + col];
              if (PASS == 2) an1[u] = lAt1[(2 * (8 * (g + 1) + u) + h) * 32 + col];
            }
          }
          if (has_next) {
            if ((g & 1) == 0) MS_STAGE_PIECE(mt + 1, cur ^ 1, g >> 1);
            if (g == 1) MS_STAGE_SCALARS(mt + 1, cur ^ 1);
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int m = 8 * g + u;
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[u], br[m], s, 0, 0, 0);
            if (PASS == 1) t = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[u], br1[m], t, 0, 0, 0);
            if (PASS == 2) t = __builtin_amdgcn_mfma_f32_32x32x2f32(ac1[u], br[m], t, 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            ac[u] = an[u];
            if (PASS == 2) ac1[u] = an1[u];
          }
        }
      }
      MS_T(T1);
#ifdef MS_TIMING
      tg1 += T1 - T0;
#endif
      // elementwise stage on D[streamed = (r&3)+8(r>>2)+4h][resident = col]
      float kv[16], gs[PASS == 0 ? 1 : 16];
      float cst[PASS == 2 ? 16 : 1], rst[PASS == 2 ? 16 : 1];
      if (PASS == 2) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int lr = 8 * g + 4 * h + u;  // streamed row inside the tile
            cst[4 * g + u] = lds_sc[cur][lr];
            rst[4 * g + u] = lds_sc[cur][32 + lr];
          }
        }
      }
      // Elementwise stage.  It is NOT interleaved with the MFMAs of the second GEMM: measured
      // with the phase timers (MS_TIMING), fp32 MFMA and VALU work of a wave do not overlap on
      // gfx950 — interleaving cost 7400 cycles per tile against 2400 + 4200 back to back — so
      // the stage is kept short instead.
      // ~10 VALU instructions per value: exp2 with log2(e) folded into the bandwidth factor,
      // the clamp test as (clamped == unclamped), the padded-column test only in the last tile.
#define MS_EW(R, TAIL)                                                                 \
  {                                                                                    \
    const float dist = __builtin_fmaf(-2.0f, s[R], 2.0f);                              \
    const float a2 = -dist * hl;                                                       \
    const float a2c = __builtin_amdgcn_fmed3f(a2, -MS_LIM2, MS_LIM2);                  \
    float k = __builtin_amdgcn_exp2f(a2c);                                             \
    if (TAIL && j0 + ((R) & 3) + 8 * ((R) >> 2) + 4 * h >= N) k = 0.f;                 \
    kv[R] = k;                                                                         \
    if (PASS == 0) rsum += k;                                                          \
    if (PASS != 0) {                                                                   \
      float g = PASS == 1 ? k * ((t[R] - c_res) * rinv_res)                            \
                          : k * ((t[R] - cst[PASS == 2 ? (R) : 0]) * rst[PASS == 2 ? (R) : 0]); \
      asm("" : "+v"(g));          /* keep the select a v_cndmask, not a branch */        \
      gs[PASS == 0 ? 0 : (R)] = a2c == a2 ? g : 0.f;                                   \
    }                                                                                  \
  }
      if (j0 + 32 > N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) MS_EW(r, true);
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) MS_EW(r, false);
      }
      MS_T(T2);
#ifdef MS_TIMING
      tew += T2 - T1;
#endif
      // second GEMM: out[f][resident] += sum_streamed P[streamed][f] * w[streamed][resident];
      // k-step m pairs the streamed indices row(m) of the two half-waves
      {
        float pc[4], pn[4], pc1[PASS == 2 ? 4 : 1], pn1[PASS == 2 ? 4 : 1];
#pragma unroll
        for (int fb = 0; fb < 4; ++fb) {
          pc[fb] = lP0[(4 * h) * MS_D + fb * 32 + col];
          if (PASS == 2) pc1[fb] = lP1[(4 * h) * MS_D + fb * 32 + col];
        }
#pragma unroll
        for (int m = 0; m < 16; ++m) {
          if (m + 1 < 16) {
            const int lrow = ((m + 1) & 3) + 8 * ((m + 1) >> 2) + 4 * h;
#pragma unroll
            for (int fb = 0; fb < 4; ++fb) {
              pn[fb] = lP0[lrow * MS_D + fb * 32 + col];
              if (PASS == 2) pn1[fb] = lP1[lrow * MS_D + fb * 32 + col];
            }
          }
#pragma unroll
          for (int fb = 0; fb < 4; ++fb) {
            const float w = PASS == 0 ? kv[m] : gs[PASS == 0 ? 0 : m];
            acc_o[fb] = __builtin_amdgcn_mfma_f32_32x32x2f32(pc[fb], w, acc_o[fb], 0, 0, 0);
            if (PASS == 2)
              acc_o[fb] = __builtin_amdgcn_mfma_f32_32x32x2f32(pc1[fb], kv[m], acc_o[fb], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int fb = 0; fb < 4; ++fb) {
            pc[fb] = pn[fb];
            if (PASS == 2) pc1[fb] = pn1[fb];
          }
        }
      }
#undef MS_EW
#ifdef MS_TIMING
      tg2 += __builtin_amdgcn_s_memtime() - T2;
#endif
    }
    MS_T(T3);
    __syncthreads();
#ifdef MS_TIMING
    tbar += __builtin_amdgcn_s_memtime() - T3;
#endif
    cur ^= 1;
  }
#ifdef MS_TIMING
  if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && tid == 0) {
    ms_dbg[PASS][0] = tg1;
    ms_dbg[PASS][1] = tg2;
    ms_dbg[PASS][2] = tbar;
    ms_dbg[PASS][3] = __builtin_amdgcn_s_memtime() - tall;
    ms_dbg[PASS][4] = t_end - t_begin;
    ms_dbg[PASS][5] = tew;
  }
#endif
  if (!wave_on) return;
  // acc_o[fb]: D[f = fb*32 + (r&3)+8(r>>2)+4h][resident = col]
  const int ir = i0 + col;
  if (ir < N) {
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
    if (h == 0 && ir < N) rpart[((size_t)b * S + slice) * N + ir] = rsum;
  }
}

// forward epilogue, one wave per row: o = sum_s opart, r = sum_s rpart;
// new = q + (o * (1/r) - q) ; y = new / ||new||   (reference order, mean_shift.py:70-77)
__global__ __launch_bounds__(256) void pn_ms_combine_fwd_kernel(
    const float* __restrict__ opart, const float* __restrict__ rpart, const float* __restrict__ q,
    int N, int S, float* __restrict__ y, float* __restrict__ rsum, float* __restrict__ unorm) {
  const int b = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + wave;
  if (i >= N) return;
  float o0 = 0.f, o1 = 0.f, r = 0.f;
  for (int s = 0; s < S; ++s) {
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

// backward prologue, one wave per row:
//   gu = (gy - y (y.gy)) / ||u|| ; c = gu . u (u = y ||u||) ; go = gu / r
// also emits the channel-first padded copies Qt, GUt the column pass streams.
__global__ __launch_bounds__(256) void pn_ms_prep_bwd_kernel(
    const float* __restrict__ gy, const float* __restrict__ y, const float* __restrict__ q,
    const float* __restrict__ rsum, const float* __restrict__ unorm, const float* __restrict__ bsq,
    int N, int Np, float* __restrict__ gu, float* __restrict__ go, float* __restrict__ cs,
    float* __restrict__ alpha, float* __restrict__ Qt, float* __restrict__ GUt) {
  const int b = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + wave;
  if (i >= Np) return;
  float* Qtb = Qt + (size_t)b * MS_D * Np;
  float* GUtb = GUt + (size_t)b * MS_D * Np;
  if (i >= N) {  // zero padding of the streamed copies
    Qtb[(size_t)lane * Np + i] = 0.f;
    Qtb[(size_t)(lane + 64) * Np + i] = 0.f;
    GUtb[(size_t)lane * Np + i] = 0.f;
    GUtb[(size_t)(lane + 64) * Np + i] = 0.f;
    return;
  }
  const size_t base = ((size_t)b * N + i) * MS_D;
  const float y0 = y[base + lane], y1 = y[base + lane + 64];
  const float g0 = gy[base + lane], g1 = gy[base + lane + 64];
  const float nn = unorm[(size_t)b * N + i], r = rsum[(size_t)b * N + i];
  const float yg = pn_wave_sum(y0 * g0 + y1 * g1);
  const float u0 = (g0 - y0 * yg) / nn, u1 = (g1 - y1 * yg) / nn;
  const float c = pn_wave_sum(u0 * (y0 * nn) + u1 * (y1 * nn));
  gu[base + lane] = u0;
  gu[base + lane + 64] = u1;
  go[base + lane] = u0 / r;
  go[base + lane + 64] = u1 / r;
  if (lane == 0) {
    cs[(size_t)b * N + i] = c;
    alpha[(size_t)b * N + i] = 1.0f / (r * bsq[b]);
  }
  Qtb[(size_t)lane * Np + i] = q[base + lane];
  Qtb[(size_t)(lane + 64) * Np + i] = q[base + lane + 64];
  GUtb[(size_t)lane * Np + i] = u0;
  GUtb[(size_t)(lane + 64) * Np + i] = u1;
}

// (B,N,D) point-major -> (B,D,Np) channel-first, zero padded
__global__ void pn_ms_pack_kernel(const float* __restrict__ x, int N, int Np,
                                  float* __restrict__ xt) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int n0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int n = n0 + i;
    tile[i][tx] = n < N ? x[((size_t)b * N + n) * MS_D + c0 + tx] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int n = n0 + tx;
    if (n < Np) xt[((size_t)b * MS_D + c0 + i) * Np + n] = tile[tx][i];
  }
}

// backward epilogue: gq = sum_s opart_q (gradient w.r.t. the previous iterate), gx += sum_s opart_x
__global__ __launch_bounds__(256) void pn_ms_combine_bwd_kernel(const float* __restrict__ opart_q,
                                                                const float* __restrict__ opart_x,
                                                                long long ND4, int Sq, int Sx,
                                                                float* __restrict__ gq,
                                                                float* __restrict__ gx) {
  const int b = blockIdx.y;
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= ND4) return;
  const float4* pq = reinterpret_cast<const float4*>(opart_q) + (size_t)b * Sq * ND4 + e;
  const float4* px = reinterpret_cast<const float4*>(opart_x) + (size_t)b * Sx * ND4 + e;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f), c = a;
  for (int s = 0; s < Sq; ++s) {
    const float4 u = pq[(size_t)s * ND4];
    a.x += u.x, a.y += u.y, a.z += u.z, a.w += u.w;
  }
  for (int s = 0; s < Sx; ++s) {
    const float4 v = px[(size_t)s * ND4];
    c.x += v.x, c.y += v.y, c.z += v.z, c.w += v.w;
  }
  reinterpret_cast<float4*>(gq)[(size_t)b * ND4 + e] = a;
  float4* g = reinterpret_cast<float4*>(gx) + (size_t)b * ND4 + e;
  float4 o = *g;
  o.x += c.x, o.y += c.y, o.z += c.z, o.w += c.w;
  *g = o;
}

// Number of slices of the streamed range.  All workgroups of a launch take the same time, so
// the grid is sized to fill an integral number of rounds over the chip's workgroup slots
// (256 CUs x blocks_per_cu): 1264 blocks on 512 slots would idle half the chip in the third
// round.  More slices cost partial-sum traffic (S x N x 512 B), hence the mild penalty.
#define MS_BPC_FWD 2  // pn_ms_kernel<0>: two workgroups per CU (256 VGPRs, 64 KiB LDS)
#define MS_BPC_BWD 1
static int ms_slices(int B, int N, int Np, int blocks_per_cu, int* tps) {
  const int ntiles = Np / 32;
  const long long rowblocks = (long long)B * pn_cdiv(N, 128);
  const long long slots = 256LL * blocks_per_cu;
  if (ntiles < 16) {
    *tps = ntiles;
    return 1;
  }
  // developer override for tuning runs: PN_MS_SLICES="<fwd>,<bwd>"
  if (const char* e = getenv("PN_MS_SLICES")) {
    int sf = 0, sb = 0;
    if (sscanf(e, "%d,%d", &sf, &sb) == 2) {
      const int want = blocks_per_cu == MS_BPC_FWD ? sf : sb;
      if (want >= 1 && want <= 32 && want <= ntiles) {
        *tps = pn_cdiv(ntiles, want);
        return want;
      }
    }
  }
  int best = 1;
  double best_score = -1.0;
  const int smax = ntiles / 8 < 32 ? ntiles / 8 : 32;  // at least 8 tiles per slice
  for (int S = 1; S <= smax; ++S) {
    const int t = pn_cdiv(ntiles, S);
    if (pn_cdiv(ntiles, t) != S) continue;  // not a distinct split
    const double rounds = (double)(rowblocks * S) / (double)slots;
    const double eff = rounds / (double)(long long)(rounds + 0.999999);
    // per-slice fixed cost (prologue/epilogue ~ 1.5 tiles) and partial-sum traffic
    const double score = eff * (double)t / ((double)t + 1.5) - 0.002 * S;
    if (score > best_score) {
      best_score = score;
      best = S;
    }
  }
  *tps = pn_cdiv(ntiles, best);
  return best;
}

extern "C" int pn_meanshift_slices(int B, int N) {
  int tps;
  const int Np = (int)pn_align_up(N, 64);
  const int sf = ms_slices(B, N, Np, MS_BPC_FWD, &tps), sb = ms_slices(B, N, Np, MS_BPC_BWD, &tps);
  const int s = sf > sb ? sf : sb;
  // the flat schedule of the block-sparse bf16 x 3 launches (meanshift_x3.h) writes up to
  // smax - 1 list fragments per block whatever the batch size: keep room for 7
  return N >= 2048 && s < 8 ? 8 : s;
}

extern "C" int pn_meanshift_pack_f32(const float* x, int B, int N, int D, float* xt, void* stream) {
  PN_CHECK_ARG(x && xt && B > 0 && N > 0, "pn_meanshift_pack_f32: bad arguments");
  PN_CHECK_ARG(D == MS_D, "pn_meanshift: embedding size %d unsupported (built for %d)", D, MS_D);
  const int Np = (int)pn_align_up(N, 64);
  dim3 grid(Np / 32, MS_D / 32, B);
  hipLaunchKernelGGL(pn_ms_pack_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, N, Np, xt);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

// One forward iteration.  q, x (B,N,D) point-major; xt = pack(x) (B,D,Np); bsq (B) = b^2.
// opart (B,S,N,D), rpart (B,S,N) scratch with S = pn_meanshift_slices(B,N).
// Outputs: y (B,N,D) next iterate, rsum, unorm (B,N) saved for the backward.
extern "C" int pn_meanshift_iter_fwd_f32(const float* q, const float* x, const float* xt,
                                         const float* bsq, int B, int N, int D, float* opart,
                                         float* rpart, float* y, float* rsum, float* unorm,
                                         void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PN_CHECK_ARG(q && x && xt && bsq && opart && rpart && y && rsum && unorm,
               "pn_meanshift_iter_fwd_f32: null pointer");
  PN_CHECK_ARG(D == MS_D, "pn_meanshift: embedding size %d unsupported (built for %d)", D, MS_D);
  PN_CHECK_ARG(B > 0 && N > 0, "pn_meanshift_iter_fwd_f32: empty input");
  const int Np = (int)pn_align_up(N, 64);
  int tps;
  const int S = ms_slices(B, N, Np, MS_BPC_FWD, &tps);
  dim3 grid(S, pn_cdiv(N, 128), B);
  {
    PN_PROF("meanshift_fwd", stream);
    hipLaunchKernelGGL(pn_ms_kernel<0>, grid, dim3(256), 0, stream, q, nullptr, xt, nullptr, x,
                       nullptr, nullptr, nullptr, bsq, N, Np, tps, opart, rpart);
  }
  PN_CHECK_LAUNCH();
  hipLaunchKernelGGL(pn_ms_combine_fwd_kernel, dim3(pn_cdiv(N, 4), B), dim3(256), 0, stream, opart,
                     rpart, q, N, S, y, rsum, unorm);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

// Backward of one iteration.  gy (B,N,D) gradient w.r.t. the iterate produced by the forward
// call with the same q/x/bsq; y, rsum, unorm its saved outputs.
// Scratch: gu, go (B,N,D), cs (B,2,N), qt, gut (B,D,Np), opart_q, opart_x (B,S,N,D).
// Outputs: gq (B,N,D) = gradient w.r.t. q (overwritten); gx (B,N,D) += gradient w.r.t. x.
extern "C" int pn_meanshift_iter_bwd_f32(const float* gy, const float* y, const float* q,
                                         const float* x, const float* xt, const float* rsum,
                                         const float* unorm, const float* bsq, int B, int N, int D,
                                         float* gu, float* go, float* cs, float* qt, float* gut,
                                         float* opart_q, float* opart_x, float* gq, float* gx,
                                         void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PN_CHECK_ARG(gy && y && q && x && xt && rsum && unorm && bsq && gu && go && cs && qt && gut &&
                   opart_q && opart_x && gq && gx,
               "pn_meanshift_iter_bwd_f32: null pointer");
  PN_CHECK_ARG(D == MS_D, "pn_meanshift: embedding size %d unsupported (built for %d)", D, MS_D);
  const int Np = (int)pn_align_up(N, 64);
  int tps;
  const int S = ms_slices(B, N, Np, MS_BPC_BWD, &tps);
  float* alpha = cs + (size_t)B * N;
  hipLaunchKernelGGL(pn_ms_prep_bwd_kernel, dim3(pn_cdiv(Np, 4), B), dim3(256), 0, stream, gy, y, q,
                     rsum, unorm, bsq, N, Np, gu, go, cs, alpha, qt, gut);
  PN_CHECK_LAUNCH();
  dim3 grid(S, pn_cdiv(N, 128), B);
  {
    PN_PROF("meanshift_bwd_rows", stream);
    hipLaunchKernelGGL(pn_ms_kernel<1>, grid, dim3(256), 0, stream, q, gu, xt, nullptr, x, nullptr,
                       cs, alpha, bsq, N, Np, tps, opart_q, nullptr);
  }
  PN_CHECK_LAUNCH();
  {
    PN_PROF("meanshift_bwd_cols", stream);
    hipLaunchKernelGGL(pn_ms_kernel<2>, grid, dim3(256), 0, stream, x, nullptr, qt, gut, q, go, cs,
                       alpha, bsq, N, Np, tps, opart_x, nullptr);
  }
  PN_CHECK_LAUNCH();
  const long long ND4 = (long long)N * MS_D / 4;
  hipLaunchKernelGGL(pn_ms_combine_bwd_kernel, dim3(pn_cdiv(ND4, 256), B), dim3(256), 0, stream,
                     opart_q, opart_x, ND4, S, S, gq, gx);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

#include "meanshift_x3.h"
#include "meanshift_h2.h"
