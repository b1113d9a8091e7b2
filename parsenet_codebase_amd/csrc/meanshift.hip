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

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MS_D 128
#define MS_KS (MS_D / 2)  // k-steps of the 128-d dot products

__device__ static inline float ms_kernel_value(float s, float hinv, bool* inside) {
  // reference: dist = 2 - 2 s ; arg = -dist / b^2 / 2 ; clamp(+-75) ; exp.
  // Evaluated as arg = -dist * (0.5 / b^2) (one rounding instead of the division's; <= 1 ulp of
  // arg) and with the hardware exp2 (v_exp_f32, ~1e-7 relative): the two division/exp library
  // sequences were a third of this kernel's time and the difference is far below the 1e-5 bar.
  const float dist = __builtin_fmaf(-2.0f, s, 2.0f);
  float arg = -dist * hinv;
  *inside = (arg >= -75.0f) && (arg <= 75.0f);
  arg = fminf(fmaxf(arg, -75.0f), 75.0f);
  return __builtin_amdgcn_exp2f(arg * 1.4426950408889634f);
}

// PASS 0: forward          resident rows = Q,  streamed cols = X : out[f][row] += X[col][f] * K
// PASS 1: backward, rows   resident rows = Q, GU; streamed cols = X : out += X[col][f] * gs
// PASS 2: backward, cols   resident cols = X;  streamed rows = Q, GU: out += Q[row][f]*gs + GO[row][f]*K
//
// R  (B,N,D)  point-major resident operand; R1 second resident operand (PASS 1: GU)
// At (B,D,Np) channel-first padded streamed operand for S; At1 for T (PASS 2: GUt; PASS 1 reuses At)
// P0 (B,N,D)  point-major streamed operand of the second GEMM; P1 second one (PASS 2: GO)
// cs, rs      per-row scalars c_i and r_i (PASS 1: indexed by the resident row; PASS 2: streamed)
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
#define MS_STAGE(MT, BUF)                                                                       \
  {                                                                                             \
    const int j0s = (MT) * 32;                                                                  \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                             \
      const int q = wave * 4 + u; /* 1 KiB chunk of the 16 KiB tile */                          \
      /* channel-first tiles [128 ch][32 idx]: chunk q = rows 8q..8q+7 */                       \
      const size_t ga = (size_t)(q * 8 + (lane >> 3)) * Np + j0s + ((lane & 7) << 2);           \
      MS_GLDS16(Atb + ga, &lds[BUF][0][q * 256]);                                               \
      if (PASS == 2) MS_GLDS16(At1b + ga, &lds[BUF][NARR - 2][q * 256]);                        \
      /* point-major tiles [32 idx][128 feat]: chunk q = rows 2q, 2q+1; rows past N clamp */    \
      const size_t gp = (size_t)min(j0s + q * 2 + (lane >> 5), N - 1) * MS_D + ((lane & 31) << 2); \
      MS_GLDS16(P0b + gp, &lds[BUF][1][q * 256]);                                               \
      if (PASS == 2) MS_GLDS16(P1b + gp, &lds[BUF][NARR - 1][q * 256]);                         \
    }                                                                                           \
    if (PASS == 2 && wave == 0) { /* per-row scalars c_i | r_i of the tile: 64 floats */        \
      const int jc = min(j0s + (lane & 31), N - 1);                                             \
      MS_GLDS4((lane < 32 ? cs : rs) + bN + jc, &lds_sc[BUF][0]);                               \
    }                                                                                           \
  }

template <int PASS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, PASS == 0 ? 2 : 1))) void pn_ms_kernel(
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
  const float hinv = 0.5f / bsq;
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
    rinv_res = 1.0f / (rs[bN + ires] * bsq);
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
  for (int mt = t_begin; mt < t_end; ++mt) {
    const int j0 = mt * 32;
    const bool has_next = mt + 1 < t_end;
    if (has_next) MS_STAGE(mt + 1, cur ^ 1);  // DMA in flight while this tile is computed
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
            rst[4 * g + u] = 1.0f / (lds_sc[cur][32 + lr] * bsq);
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        bool inside;
        float k = ms_kernel_value(s[r], hinv, &inside);
        if (j0 + row >= N) k = 0.f;
        kv[r] = k;
        if (PASS == 0) rsum += k;
        if (PASS == 1) gs[r] = inside ? k * (t[r] - c_res) * rinv_res : 0.f;
        if (PASS == 2) gs[r] = inside ? k * (t[r] - cst[r]) * rst[r] : 0.f;
      }
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
            const float w = PASS == 0 ? kv[m] : gs[m];
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
    }
    __syncthreads();
    cur ^= 1;
  }
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
    const float* __restrict__ rsum, const float* __restrict__ unorm, int N, int Np,
    float* __restrict__ gu, float* __restrict__ go, float* __restrict__ cs,
    float* __restrict__ Qt, float* __restrict__ GUt) {
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
  if (lane == 0) cs[(size_t)b * N + i] = c;
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

static int ms_slices(int B, int N, int Np, int* tps) {
  const long long waves = (long long)B * pn_cdiv(N, 32);
  (void)waves;
  const int ntiles = Np / 32;
  // 8 or 16 slices (a multiple of the 8 XCDs, see pn_ms_kernel); tiny problems take fewer
  int S = ntiles >= 64 ? 16 : (ntiles >= 16 ? 8 : 1);
  *tps = pn_cdiv(ntiles, S);
  // keep S itself (not cdiv(ntiles, tps)): trailing empty slices are harmless and preserve the
  // slice -> XCD mapping
  return S;
}

extern "C" int pn_meanshift_slices(int B, int N) {
  int tps;
  return ms_slices(B, N, (int)pn_align_up(N, 64), &tps);
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
  const int S = ms_slices(B, N, Np, &tps);
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
// Scratch: gu, go (B,N,D), cs (B,N), qt, gut (B,D,Np), opart_q, opart_x (B,S,N,D).
// After the call sum_s opart_q is the gradient w.r.t. q and sum_s opart_x the contribution to
// the gradient w.r.t. x (the caller reduces over s and accumulates across iterations).
extern "C" int pn_meanshift_iter_bwd_f32(const float* gy, const float* y, const float* q,
                                         const float* x, const float* xt, const float* rsum,
                                         const float* unorm, const float* bsq, int B, int N, int D,
                                         float* gu, float* go, float* cs, float* qt, float* gut,
                                         float* opart_q, float* opart_x, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PN_CHECK_ARG(gy && y && q && x && xt && rsum && unorm && bsq && gu && go && cs && qt && gut &&
                   opart_q && opart_x,
               "pn_meanshift_iter_bwd_f32: null pointer");
  PN_CHECK_ARG(D == MS_D, "pn_meanshift: embedding size %d unsupported (built for %d)", D, MS_D);
  const int Np = (int)pn_align_up(N, 64);
  int tps;
  const int S = ms_slices(B, N, Np, &tps);
  hipLaunchKernelGGL(pn_ms_prep_bwd_kernel, dim3(pn_cdiv(Np, 4), B), dim3(256), 0, stream, gy, y, q,
                     rsum, unorm, N, Np, gu, go, cs, qt, gut);
  PN_CHECK_LAUNCH();
  dim3 grid(S, pn_cdiv(N, 128), B);
  {
    PN_PROF("meanshift_bwd_rows", stream);
    hipLaunchKernelGGL(pn_ms_kernel<1>, grid, dim3(256), 0, stream, q, gu, xt, nullptr, x, nullptr,
                       cs, rsum, bsq, N, Np, tps, opart_q, nullptr);
  }
  PN_CHECK_LAUNCH();
  {
    PN_PROF("meanshift_bwd_cols", stream);
    hipLaunchKernelGGL(pn_ms_kernel<2>, grid, dim3(256), 0, stream, x, nullptr, qt, gut, q, go, cs,
                       rsum, bsq, N, Np, tps, opart_x, nullptr);
  }
  PN_CHECK_LAUNCH();
  return PN_OK;
}
