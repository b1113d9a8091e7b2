// Mean-shift iterations with fp32-grade products on the fp16 matrix cores ("fp16 x 2").
//
// Same mathematics, data flow and outputs as meanshift_x3.h; the operands are split into two
// fp16 pieces instead of three bf16 ones:
//     x 2^s = xh + xm (+ e),  xh = fp16(x 2^s), xm = fp16(x 2^s - xh),  |e| <= 2^-22 |x 2^s|
// and a product is evaluated as  xh*yh + xh*ym + xm*yh  with fp32 accumulation on
// v_mfma_f32_32x32x16_f16: three MFMAs per 32x32x16 block instead of six.  The dropped term
// xm*ym and the representation error e are both <= 2^-22 relative per product, random in sign;
// in a dot product they sum to less than the rounding error of an fp32 fma chain of the same
// length (measured against the fp64 oracle in tests/test_meanshift_gpu.py next to the other two
// arithmetics, and in tools/h2_accuracy.py against a plain fp32 GEMM: 1.3e-7 vs 6e-7 on S).
//
// fp16 has a 5-bit exponent, so every operand is brought into range by a power of two (exact):
//   unit rows (X, the iterates)         x 2^12                                (|.| <= 4096)
//   kernel values k in [0,1]            x 2^14   (exp2 of the shifted argument)
//   gu rows (backward)                  x ge_i 2^12, ge_i = 2^-e_i the row's own power of two
//                                       (max_c |gu_ic| ge_i in [1/2,1))
//   backward weights                    (k 2^14) (T' - c') 2^-5 [x rho_i sigma in the column pass],
//                                       T' = gu'_i . x_j, c' = c_i ge_i, |T' - c'| < 2 sqrt(128)
//   rho_i = alpha_i / ge_i              the row's gradient magnitude; the column pass contracts
//                                       over rows, so their scales must share one power of two:
//                                       sigma = 2^-E, rho_max sigma in [1/2,1) (per-tile maxima from
//                                       the prologue).  Rows more than 2^-24 below the largest one
//                                       lose relative — not absolute — precision.
// The scales are undone by exact power-of-two factors at the store (per lane in the row pass).
// Values below 2^-24 after scaling flush to zero: an absolute error below 2^-36 of the
// operand's bound.  Rows of X / the iterates must be unit vectors (|x_c| < 16 is what fits).
//
// Tile images: 16 KiB per 32-point tile, [piece 2][row 32][16 chunks of 8 channels], the
// swizzle, the two read paths (ds_read_b128 / ds_read_b64_tr_b16) and the D-layout -> B-operand
// hand-over are those of meanshift_x3.h.
// (included at the end of meanshift.hip, after meanshift_x3.h)

#include "split_common.h"

// 2^-(e+1) for a positive float with exponent e (value in [2^e, 2^(e+1))): the power of two that
// brings it into [1/2, 1), limited to 2^+-100 (so that products with it stay finite; rows that
// small are zero for every purpose).
__device__ static inline float h2_norm_scale(float v) {
  int f = 253 - (int)(__builtin_bit_cast(uint32_t, v) >> 23 & 0xff);
  f = f < 27 ? 27 : (f > 227 ? 227 : f);
  return __builtin_bit_cast(float, (uint32_t)f << 23);
}

// x (B,N,D) fp32, optional per-row scale (rowscale[b rs_stride + i]) -> the image of every 32-point tile (rows >= N zero).
__global__ __launch_bounds__(256) void pn_msh_split_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ rowscale,
                                                           long long rs_stride, int N, int ntiles,
                                                           u32x4* __restrict__ pimg) {
  const int b = blockIdx.y, tile = blockIdx.x;
  const float* __restrict__ xb = x + (size_t)b * N * MS_D;
  u32x4* __restrict__ P = pimg + ((size_t)b * ntiles + tile) * H2_IMG_U4;
  const int j0 = tile * 32;
  for (int it = threadIdx.x; it < 512; it += 256) {
    const int j = it >> 4, c = it & 15;  // row j, chunk c = channels 8c..8c+7
    float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
    float sc = H2_SX;
    if (j0 + j < N) {
      const float* src = xb + (size_t)(j0 + j) * MS_D + 8 * c;
      v0 = *reinterpret_cast<const float4*>(src);
      v1 = *reinterpret_cast<const float4*>(src + 4);
      if (rowscale) sc *= rowscale[(size_t)b * rs_stride + j0 + j];
    }
    u32x4 h, m;
    H2_SPLIT_TO(v0.x * sc, v0.y * sc, h, m, 0);
    H2_SPLIT_TO(v0.z * sc, v0.w * sc, h, m, 1);
    H2_SPLIT_TO(v1.x * sc, v1.y * sc, h, m, 2);
    H2_SPLIT_TO(v1.z * sc, v1.w * sc, h, m, 3);
    const int slot = j * 16 + (c ^ x3_swz(j));
    P[slot] = h;
    P[H2_PIECE_U4 + slot] = m;
  }
}

// backward prologue, one workgroup per 32-row tile, one wave per row at a time (rows N .. Np-1 of
// the images are written as zeros):
//   gu = (gy - y (y.gy)) / ||u|| ; c = gu . u (u = y ||u||) ; alpha = 1 / (r b^2)
//   ge = power of two normalising the row of gu ; rowsc = [c ge 2^-5 | rho = alpha / ge | ge]
//   rhotile[b][tile] = max over the tile's rows of rho (the column pass reduces these; no atomics:
//   10 000 same-address atomic maxima serialise to ~100 us)
//   img_q, img_gu: the tile images of q and of gu' = gu ge (each lane owns channels 2 lane, 2 lane + 1:
//   one packed pair per piece, 4 bytes into chunk lane / 4)
__global__ __launch_bounds__(256) void pn_msh_prep_bwd_kernel(
    const float* __restrict__ gy, const float* __restrict__ y, const float* __restrict__ q,
    const float* __restrict__ rsum, const float* __restrict__ unorm, const float* __restrict__ bsq, int N,
    int ntiles, float* __restrict__ gu, float* __restrict__ rowsc, float* __restrict__ rhotile,
    uint32_t* __restrict__ img_q, uint32_t* __restrict__ img_gu) {
  __shared__ float wmax[4];
  const int b = blockIdx.y, tile = blockIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float ibsq = 1.0f / bsq[b];
  float rmax = 0.f;
  for (int j = wave; j < 32; j += 4) {
    const int i = tile * 32 + j;
    // word (4 bytes) of this lane inside the row's 256-byte image line, per piece
    const size_t word = ((size_t)b * ntiles + tile) * (H2_IMG_U4 * 4) +
                        (size_t)(j * 16 + ((lane >> 2) ^ x3_swz(j))) * 4 + (lane & 3);
    if (i >= N) {
      img_q[word] = 0u;
      img_q[word + H2_PIECE_U4 * 4] = 0u;
      img_gu[word] = 0u;
      img_gu[word + H2_PIECE_U4 * 4] = 0u;
      continue;
    }
    const size_t base = ((size_t)b * N + i) * MS_D + 2 * lane;
    const float2 yv = *reinterpret_cast<const float2*>(y + base);
    const float2 gv = *reinterpret_cast<const float2*>(gy + base);
    const float2 qv = *reinterpret_cast<const float2*>(q + base);
    const float nn = unorm[(size_t)b * N + i], r = rsum[(size_t)b * N + i];
    const float yg = pn_wave_sum(yv.x * gv.x + yv.y * gv.y);
    const float u0 = (gv.x - yv.x * yg) / nn, u1 = (gv.y - yv.y * yg) / nn;
    const float c = pn_wave_sum(u0 * (yv.x * nn) + u1 * (yv.y * nn));
    float mx = fmaxf(fabsf(u0), fabsf(u1));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    *reinterpret_cast<float2*>(gu + base) = make_float2(u0, u1);
    const float ge = h2_norm_scale(mx);
    const H2Pieces pq = h2_split2(qv.x * H2_SX, qv.y * H2_SX);
    const float gs = ge * H2_SX;
    const H2Pieces pg = h2_split2(u0 * gs, u1 * gs);
    img_q[word] = pq.h;
    img_q[word + H2_PIECE_U4 * 4] = pq.m;
    img_gu[word] = pg.h;
    img_gu[word + H2_PIECE_U4 * 4] = pg.m;
    const float rho = (ibsq / r) / ge;
    if (lane == 0) {
      float* rs = rowsc + (size_t)b * 3 * N;
      rs[i] = c * ge * 0.03125f;
      rs[N + i] = rho;
      rs[2 * N + i] = ge;
    }
    if (rho > 0.f && rho < __builtin_inff()) rmax = fmaxf(rmax, rho);
  }
  if (lane == 0) wmax[wave] = rmax;
  __syncthreads();
  if (threadIdx.x == 0)
    rhotile[(size_t)b * ntiles + tile] = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
}

// PASS 0 forward       resident rows Q;      streamed X:      out[f][i] += X[j][f] K
// PASS 1 backward/rows resident rows Q, GU;  streamed X:      out[f][i] += X[j][f] gs
// PASS 2 backward/cols resident cols X;      streamed Q, GU:  out[f][j] += Q[i][f] gs + GU[i][f] K / r_i
//
// R, R1       (B,N,D) fp32 resident operands (scaled and split in registers once per workgroup)
// PA, PB      tile images of the streamed operand(s) (PB: GU', PASS 2 only); both GEMMs read them
// rowsc       (B,3,N): c_i ge_i 2^-5 | rho_i | ge_i — of the resident row (PASS 1) / streamed (PASS 2)
// grid (slices, blocks of 32 NW resident indices, B), 64 NW threads: wave w owns 32 w .. 32 w + 31.
// LDS: images double buffered: 32 KiB (PASS 0/1), 64 KiB (PASS 2); one barrier per tile.
#define H2_WAVES(PASS) ((PASS) == 1 ? H2_ROW_WAVES : 8)
#ifndef H2_ROW_WAVES
#define H2_ROW_WAVES 8
#endif
template <int PASS>
__global__ __launch_bounds__(64 * H2_WAVES(PASS))
__attribute__((amdgpu_waves_per_eu(H2_WAVES(PASS) / 4, H2_WAVES(PASS) / 4))) void pn_msh_kernel(
    const float* __restrict__ R, const float* __restrict__ R1, const u32x4* __restrict__ PA,
    const u32x4* __restrict__ PB, const float* __restrict__ rowsc, const float* __restrict__ rhotile,
    const float* __restrict__ bsq_, int N, int ntiles, int tiles_per_slice, float* __restrict__ opart,
    float* __restrict__ rpart) {
  constexpr int NIMG = PASS == 2 ? 2 : 1;
  __shared__ __attribute__((aligned(16))) u32x4 ldsP[2][NIMG][H2_IMG_U4];
  __shared__ __attribute__((aligned(16))) float lds_sc[2][64];
  const int b = blockIdx.z;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int col = lane & 31, h = lane >> 5;
  constexpr int NW = H2_WAVES(PASS);
  const int i0 = (blockIdx.y * NW + wave) * 32;
  const bool wave_on = i0 < N;
  const int S = gridDim.x, slice = blockIdx.x;
  const int t_begin = slice * tiles_per_slice;
  const int t_end = min(ntiles, t_begin + tiles_per_slice);
  const float bsqv = bsq_[b];
  const float hl = (0.5f / bsqv) * MS_LOG2E;
  const size_t bN = (size_t)b * N;
  const size_t boff = (size_t)b * ntiles * H2_IMG_U4;
  const u32x4* __restrict__ PAb = PA + boff;
  const u32x4* __restrict__ PBb = PASS == 2 ? PB + boff : nullptr;
  const float* __restrict__ rs_c = PASS == 0 ? nullptr : rowsc + (size_t)b * 3 * N;
  const float* __restrict__ rs_rho = PASS == 0 ? nullptr : rs_c + N;

  // global scale of the column pass from the largest rho of the shape (maximum over the tile
  // maxima of the prologue): sigma = 2^(126 - E), 1 / sigma = 2^(E - 126)
  float sigma = 1.f, isigma = 1.f;
  if (PASS == 2) {
    float m = 0.f;
    for (int t = lane; t < ntiles; t += 64) m = fmaxf(m, rhotile[(size_t)b * ntiles + t]);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    int E = (int)(__builtin_bit_cast(uint32_t, m) >> 23 & 0xff);
    E = E < 1 ? 126 : (E > 252 ? 252 : E);
    sigma = __builtin_bit_cast(float, (uint32_t)(253 - E) << 23);
    isigma = __builtin_bit_cast(float, (uint32_t)(E + 1) << 23);
  }

  // resident operand(s) as B operands of the first GEMM: k-step s = channels 16 s + 8 h + e
  const int ires = min(i0 + col, N - 1);
  float c_res = 0.f, rho_res = 0.f, gsc = 0.f;
  if (PASS == 1) {
    c_res = rs_c[ires];
    rho_res = rs_rho[ires];
    gsc = rs_c[2 * N + ires] * H2_SX;
  }
  f16x8 qh[8], qm[8];
  f16x8 uh[PASS == 1 ? 8 : 1], um[PASS == 1 ? 8 : 1];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    {
      const float* src = R + (bN + ires) * MS_D + 16 * s + 8 * h;
      const float4 a = *reinterpret_cast<const float4*>(src);
      const float4 c = *reinterpret_cast<const float4*>(src + 4);
      u32x4 vh, vm;
      H2_SPLIT_TO(a.x * H2_SX, a.y * H2_SX, vh, vm, 0);
      H2_SPLIT_TO(a.z * H2_SX, a.w * H2_SX, vh, vm, 1);
      H2_SPLIT_TO(c.x * H2_SX, c.y * H2_SX, vh, vm, 2);
      H2_SPLIT_TO(c.z * H2_SX, c.w * H2_SX, vh, vm, 3);
      qh[s] = h2_as_f16(vh);
      qm[s] = h2_as_f16(vm);
    }
    if (PASS == 1) {
      const float* src = R1 + (bN + ires) * MS_D + 16 * s + 8 * h;
      const float4 a = *reinterpret_cast<const float4*>(src);
      const float4 c = *reinterpret_cast<const float4*>(src + 4);
      u32x4 vh, vm;
      H2_SPLIT_TO(a.x * gsc, a.y * gsc, vh, vm, 0);
      H2_SPLIT_TO(a.z * gsc, a.w * gsc, vh, vm, 1);
      H2_SPLIT_TO(c.x * gsc, c.y * gsc, vh, vm, 2);
      H2_SPLIT_TO(c.z * gsc, c.w * gsc, vh, vm, 3);
      uh[PASS == 1 ? s : 0] = h2_as_f16(vh);
      um[PASS == 1 ? s : 0] = h2_as_f16(vm);
    }
  }
  f32x16 acc_o[4];
#pragma unroll
  for (int fb = 0; fb < 4; ++fb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_o[fb][r] = 0.f;
  float rsum = 0.f;

  // a 16 KiB image = 16 chunks of 1 KiB (64 lanes x 16 B), dealt evenly to the NW waves
#define H2_STAGE(SRC, DST)                                                        \
  {                                                                               \
    _Pragma("unroll") for (int u = 0; u < 16 / NW; ++u) {                         \
      const int q = wave * (16 / NW) + u;                                         \
      X3_GLDS16((SRC) + q * 64 + lane, &(DST)[q * 64]);                           \
    }                                                                             \
  }
#define H2_STAGE_P(MT, BUF)                                                       \
  {                                                                               \
    H2_STAGE(PAb + (size_t)(MT) * H2_IMG_U4, ldsP[BUF][0]);                       \
    if (PASS == 2) {                                                              \
      H2_STAGE(PBb + (size_t)(MT) * H2_IMG_U4, ldsP[BUF][NIMG - 1]);              \
      if (wave == 0) { /* c' | rho of the 32 streamed rows */                     \
        const int jc = min((MT) * 32 + (lane & 31), N - 1);                       \
        __builtin_amdgcn_global_load_lds((x3_gptr)((lane < 32 ? rs_c : rs_rho) + jc), \
                                         (x3_lptr)&lds_sc[BUF][0], 4, 0, 0);      \
      }                                                                           \
    }                                                                             \
  }
  int cur = 0;
  if (t_begin < t_end) H2_STAGE_P(t_begin, 0);
#ifdef MS_TIMING
  unsigned long long tb0 = 0, tdma = 0, tg1 = 0, tew = 0, tb1 = 0, tg2 = 0, tall = __builtin_amdgcn_s_memtime();
#endif
  for (int mt = t_begin; mt < t_end; ++mt) {
    const int j0 = mt * 32;
    MS_T(U0);
#ifdef MS_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tb1 += __builtin_amdgcn_s_memtime() - U0;
#endif
    __syncthreads();  // image(s) of tile mt landed; every wave is done with tile mt - 1
    MS_T(U1);
    if (mt + 1 < t_end) H2_STAGE_P(mt + 1, cur ^ 1);
    MS_T(U2);
    u32x4 wh[2], wm[2];                                // weights of the second GEMM
    u32x4 vh[PASS == 2 ? 2 : 1], vm[PASS == 2 ? 2 : 1];  // PASS 2: K / r
    f32x16 sa, ta;
    if (wave_on) {
      // ---- first GEMM: S[streamed][resident] (and T with the second operand) ----
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        sa[r] = 0.f;
        ta[r] = 0.f;
      }
      const u32x4* __restrict__ lp = ldsP[cur][0];
      const u32x4* __restrict__ lp1 = ldsP[cur][NIMG - 1];
      const int rowoff = col * 16, sw = x3_swz(col);
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const int slot = rowoff + ((2 * s + h) ^ sw);
        const f16x8 ah = h2_as_f16(lp[slot]);
        const f16x8 am = h2_as_f16(lp[H2_PIECE_U4 + slot]);
        H2_MFMA(sa, am, qh[s]);
        H2_MFMA(sa, ah, qm[s]);
        H2_MFMA(sa, ah, qh[s]);
        if (PASS == 1) {  // T = X . GU': same streamed operand, second resident one
          const int z = PASS == 1 ? s : 0;
          H2_MFMA(ta, am, uh[z]);
          H2_MFMA(ta, ah, um[z]);
          H2_MFMA(ta, ah, uh[z]);
        }
        if (PASS == 2) {  // T = GU' . X: second streamed operand, same resident one
          const f16x8 gh = h2_as_f16(lp1[slot]);
          const f16x8 gm = h2_as_f16(lp1[H2_PIECE_U4 + slot]);
          H2_MFMA(ta, gm, qh[s]);
          H2_MFMA(ta, gh, qm[s]);
          H2_MFMA(ta, gh, qh[s]);
        }
      }
      MS_T(U3);
#ifdef MS_TIMING
      tg1 += U3 - U2;
#endif
      MS_T(U5);
#ifdef MS_TIMING
      tb0 += U1 - U0;
      tdma += U2 - U1;
#endif
      // ---- elementwise stage on D[streamed = (r&3)+8(r>>2)+4h][resident = col] ----
      //   S = sa 2^-24;  k 2^14 = exp2(a2c + 14);  backward weight = k 2^14 (T' - c') 2^-5
      const bool tail = j0 + 32 > N;
      float kv[16], gs[PASS == 0 ? 1 : 16];
#define H2_EW_(R, MASKED)                                                          \
  {                                                                                \
    const int row = ((R) & 3) + 8 * ((R) >> 2) + 4 * h;                            \
    const float sv = sa[R];                                                        \
    const float dist = __builtin_fmaf(-2.0f * H2_ISX2, sv, 2.0f);                  \
    const float a2 = -dist * hl;                                                   \
    const float a2c = __builtin_amdgcn_fmed3f(a2, -MS_LIM2, MS_LIM2);              \
    float k = __builtin_amdgcn_exp2f(a2c + 14.0f);                                 \
    if (PASS == 0 && (MASKED) && j0 + row >= N) k = 0.f;                           \
    kv[R] = k;                                                                     \
    if (PASS == 0) rsum += k;                                                      \
    if (PASS != 0) {                                                               \
      const float cc = PASS == 1 ? c_res : lds_sc[cur][row];                       \
      float g = k * __builtin_fmaf(ta[R], 0x1p-29f, -cc);                          \
      if (PASS == 2) {                                                             \
        const float aa = lds_sc[cur][32 + row] * sigma;                            \
        g *= aa;                                                                   \
        kv[R] = k * (aa * (bsqv * 0x1p-5f)); /* weight of the GU' term */          \
      }                                                                            \
      asm("" : "+v"(g));          /* keep the select a v_cndmask, not a branch */  \
      gs[PASS == 0 ? 0 : (R)] = a2c == a2 ? g : 0.f;                               \
    }                                                                              \
  }
#define H2_EW(R) H2_EW_(R, false)
#define H2_SPLIT_W(T, Q)                                                                      \
  {                                                                                           \
    if (PASS == 0) {                                                                          \
      H2_SPLIT_TO(kv[8 * (T) + 2 * (Q)], kv[8 * (T) + 2 * (Q) + 1], wh[T], wm[T], Q);        \
    } else {                                                                                  \
      const int e = PASS == 0 ? 0 : 8 * (T) + 2 * (Q);                                        \
      H2_SPLIT_TO(gs[e], gs[e + (PASS == 0 ? 0 : 1)], wh[T], wm[T], Q);                      \
      if (PASS == 2) {                                                                        \
        const int tt = PASS == 2 ? (T) : 0;                                                   \
        H2_SPLIT_TO(kv[8 * (T) + 2 * (Q)], kv[8 * (T) + 2 * (Q) + 1], vh[tt], vm[tt], Q);    \
      }                                                                                       \
    }                                                                                         \
  }
      constexpr bool PIPE = PASS != 0;      // elementwise stage in two halves around k-step 0
      // (only the last tile of the forward pass pays for the mask)
#define H2_EW_RANGE(LO, HI)                                                   \
  if (PASS == 0 && tail) {                                                    \
    _Pragma("unroll") for (int r = (LO); r < (HI); ++r) H2_EW_(r, true);      \
  } else {                                                                    \
    _Pragma("unroll") for (int r = (LO); r < (HI); ++r) H2_EW(r);             \
  }
      H2_EW_RANGE(0, PIPE ? 8 : 16);
#pragma unroll
      for (int q = 0; q < 4; ++q) H2_SPLIT_W(0, q);
      if (!PIPE) {
#pragma unroll
        for (int q = 0; q < 4; ++q) H2_SPLIT_W(1, q);
      }
      MS_T(U6);
#ifdef MS_TIMING
      tew += U6 - U5;
#endif
      // ---- second GEMM: out[f][resident] += sum_streamed C[f][streamed] w[streamed][resident];
      //      k-step t = D registers 8t..8t+7 of the first GEMM (see meanshift_x3.h) ----
      const char* lbase = reinterpret_cast<const char*>(ldsP[cur][0]);
      const char* lbase1 = reinterpret_cast<const char*>(ldsP[cur][NIMG - 1]);
      const int li = lane & 15, rb = 4 * h + (li >> 2), cb = 16 * ((lane >> 4) & 1) + 4 * (li & 3);
      const int sz0 = (((li >> 2) & 3) << 2) | (h & 3), sz1 = (((li >> 2) & 3) << 2) | ((h + 2) & 3);
      u32x4 xc[2], oc[PASS == 2 ? 2 : 1];
#define H2_TR(BASE, P_, T, W, FB)                                                               \
  __builtin_amdgcn_ds_read_tr16_b64_v4i16((x3_lds_s16x4)(                                        \
      (BASE) + (P_) * (H2_PIECE_U4 * 16) + (16 * (T) + 8 * (W) + rb) * 256 +                    \
      ((((FB) * 4 + (cb >> 3)) ^ ((W) ? sz1 : sz0)) << 4) + ((cb & 7) << 1)))
#define H2_LOAD_C(DX, DO, T, FB)                                                                \
  {                                                                                             \
    _Pragma("unroll") for (int p_ = 0; p_ < 2; ++p_) {                                          \
      const s16x4 lo_ = H2_TR(lbase, p_, T, 0, FB), hi_ = H2_TR(lbase, p_, T, 1, FB);           \
      const s16x8 v_ = __builtin_shufflevector(lo_, hi_, 0, 1, 2, 3, 4, 5, 6, 7);               \
      DX[p_] = __builtin_bit_cast(u32x4, v_);                                                   \
      if (PASS == 2) {                                                                          \
        const s16x4 lo1_ = H2_TR(lbase1, p_, T, 0, FB), hi1_ = H2_TR(lbase1, p_, T, 1, FB);     \
        const s16x8 w_ = __builtin_shufflevector(lo1_, hi1_, 0, 1, 2, 3, 4, 5, 6, 7);           \
        DO[PASS == 2 ? p_ : 0] = __builtin_bit_cast(u32x4, w_);                                 \
      }                                                                                         \
    }                                                                                           \
  }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if (PIPE && t == 1) {
          H2_EW_RANGE(8, 16);
#pragma unroll
          for (int q = 0; q < 4; ++q) H2_SPLIT_W(1, q);
        }
        const f16x8 bh = h2_as_f16(wh[t]), bm = h2_as_f16(wm[t]);
#pragma unroll
        for (int fb = 0; fb < 4; ++fb) {
          H2_LOAD_C(xc, oc, t, fb);
          const f16x8 xh = h2_as_f16(xc[0]), xm = h2_as_f16(xc[1]);
          H2_MFMA(acc_o[fb], xm, bh);
          H2_MFMA(acc_o[fb], xh, bm);
          H2_MFMA(acc_o[fb], xh, bh);
          if (PASS == 2) {
            const int tt = PASS == 2 ? t : 0;
            const f16x8 kh = h2_as_f16(vh[tt]), km = h2_as_f16(vm[tt]);
            const f16x8 oh = h2_as_f16(oc[0]), om = h2_as_f16(oc[PASS == 2 ? 1 : 0]);
            H2_MFMA(acc_o[fb], om, kh);
            H2_MFMA(acc_o[fb], oh, km);
            H2_MFMA(acc_o[fb], oh, kh);
          }
        }
      }
#undef H2_LOAD_C
#undef H2_TR
#undef H2_SPLIT_W
#undef H2_EW_RANGE
#undef H2_EW
#undef H2_EW_
#ifdef MS_TIMING
      tg2 += __builtin_amdgcn_s_memtime() - U6;
#endif
    }
    cur ^= 1;
  }
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
  if (!wave_on) return;
  const int ir = i0 + col;
  // undo the operand scales: forward 2^-(14+12); rows 2^-(9+12) rho_i; columns 2^-(9+12) / sigma
  const float oscale = PASS == 0 ? 0x1p-26f : (PASS == 1 ? 0x1p-21f * rho_res : 0x1p-21f * isigma);
  if (ir < N) {
    float* o = opart + (((size_t)b * S + slice) * N + ir) * MS_D;
#pragma unroll
    for (int fb = 0; fb < 4; ++fb)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(o + fb * 32 + 8 * g + 4 * h) =
            make_float4(acc_o[fb][4 * g] * oscale, acc_o[fb][4 * g + 1] * oscale,
                        acc_o[fb][4 * g + 2] * oscale, acc_o[fb][4 * g + 3] * oscale);
  }
  if (PASS == 0) {
    rsum += __shfl_xor(rsum, 32, 64);
    if (h == 0 && ir < N) rpart[((size_t)b * S + slice) * N + ir] = rsum * 0x1p-14f;
  }
}

extern "C" size_t pn_meanshift_h2_image_bytes(int B, int N) {
  const int Np = (int)pn_align_up(N, 64);
  return (size_t)B * (Np / 32) * H2_IMG_U4 * 16;
}

// x (B,N,D), rows of unit length -> its tile-image array (pn_meanshift_h2_image_bytes(B,N) bytes)
extern "C" int pn_meanshift_h2_split_f32(const float* x, int B, int N, int D, void* img, void* stream) {
  PN_CHECK_ARG(x && img && B > 0 && N > 0, "pn_meanshift_h2_split_f32: bad arguments");
  PN_CHECK_ARG(D == MS_D, "pn_meanshift: embedding size %d unsupported (built for %d)", D, MS_D);
  const int ntiles = (int)pn_align_up(N, 64) / 32;
  hipLaunchKernelGGL(pn_msh_split_kernel, dim3(ntiles, B), dim3(256), 0, (hipStream_t)stream, x,
                     (const float*)nullptr, 0LL, N, ntiles, (u32x4*)img);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

// One forward iteration on the fp16 x 2 path: same contract as pn_meanshift_x3_iter_fwd_f32 with
// the images of pn_meanshift_h2_split_f32.
extern "C" int pn_meanshift_h2_iter_fwd_f32(const float* q, const void* img_x, const float* bsq, int B,
                                            int N, int D, float* opart, float* rpart, float* y,
                                            float* rsum, float* unorm, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PN_CHECK_ARG(q && img_x && bsq && opart && rpart && y && rsum && unorm,
               "pn_meanshift_h2_iter_fwd_f32: null pointer");
  PN_CHECK_ARG(D == MS_D, "pn_meanshift: embedding size %d unsupported (built for %d)", D, MS_D);
  PN_CHECK_ARG(B > 0 && N > 0, "pn_meanshift_h2_iter_fwd_f32: empty input");
  const int ntiles = (int)pn_align_up(N, 64) / 32;
  int tps;
  int S = x3_slices(B, N, ntiles, 2, &tps);
  const int smax = pn_meanshift_slices(B, N);  // the scratch is sized for this many slices
  if (S > smax) {
    S = smax;
    tps = pn_cdiv(ntiles, S);
  }
  dim3 grid(S, pn_cdiv(N, 256), B);
  {
    PN_PROF("meanshift_fwd", stream);
    hipLaunchKernelGGL(pn_msh_kernel<0>, grid, dim3(512), 0, stream, q, nullptr, (const u32x4*)img_x,
                       nullptr, nullptr, nullptr, bsq, N, ntiles, tps, opart, rpart);
  }
  PN_CHECK_LAUNCH();
  hipLaunchKernelGGL(pn_ms_combine_fwd_kernel, dim3(pn_cdiv(N, 4), B), dim3(256), 0, stream, opart,
                     rpart, q, N, S, y, rsum, unorm);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

// Backward of one iteration on the fp16 x 2 path: same contract as pn_meanshift_x3_iter_bwd_f32;
// rowsc is a scratch of 3 B N + B ntiles floats, ntiles = 2 ceil(N / 64) (row scalars and per-tile maxima).
extern "C" int pn_meanshift_h2_iter_bwd_f32(const float* gy, const float* y, const float* q,
                                            const float* x, const void* img_x, const float* rsum,
                                            const float* unorm, const float* bsq, int B, int N, int D,
                                            float* gu, float* rowsc, void* img_q, void* img_gu,
                                            float* opart_q, float* opart_x, float* gq, float* gx,
                                            void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PN_CHECK_ARG(gy && y && q && x && img_x && rsum && unorm && bsq && gu && rowsc && img_q && img_gu &&
                   opart_q && opart_x && gq && gx,
               "pn_meanshift_h2_iter_bwd_f32: null pointer");
  PN_CHECK_ARG(D == MS_D, "pn_meanshift: embedding size %d unsupported (built for %d)", D, MS_D);
  const int ntiles = (int)pn_align_up(N, 64) / 32;
  int tps, tps2;
  const int smax = pn_meanshift_slices(B, N);
  int S = x3_slices(B, N, ntiles, H2_ROW_WAVES == 8 ? 2 : 1, &tps);    // row pass
  if (S > smax) {
    S = smax;
    tps = pn_cdiv(ntiles, S);
  }
  int S2 = x3_slices(B, N, ntiles, 2, &tps2);  // column pass: 8-wave workgroups
  if (S2 > smax) {
    S2 = smax;
    tps2 = pn_cdiv(ntiles, S2);
  }
  float* rhotile = rowsc + (size_t)3 * B * N;
  hipLaunchKernelGGL(pn_msh_prep_bwd_kernel, dim3(ntiles, B), dim3(256), 0, stream, gy, y, q, rsum, unorm, bsq,
                     N, ntiles, gu, rowsc, rhotile, (uint32_t*)img_q, (uint32_t*)img_gu);
  PN_CHECK_LAUNCH();
  {
    PN_PROF("meanshift_bwd_rows", stream);
    dim3 grid(S, pn_cdiv(N, 32 * H2_WAVES(1)), B);
    hipLaunchKernelGGL(pn_msh_kernel<1>, grid, dim3(64 * H2_WAVES(1)), 0, stream, q, (const float*)gu,
                       (const u32x4*)img_x, nullptr, (const float*)rowsc, (const float*)rhotile, bsq, N,
                       ntiles, tps, opart_q, nullptr);
  }
  PN_CHECK_LAUNCH();
  {
    PN_PROF("meanshift_bwd_cols", stream);
    dim3 grid2(S2, pn_cdiv(N, 32 * H2_WAVES(2)), B);
    hipLaunchKernelGGL(pn_msh_kernel<2>, grid2, dim3(64 * H2_WAVES(2)), 0, stream, x, nullptr,
                       (const u32x4*)img_q, (const u32x4*)img_gu, (const float*)rowsc,
                       (const float*)rhotile, bsq, N, ntiles, tps2, opart_x, nullptr);
  }
  PN_CHECK_LAUNCH();
  const long long ND4 = (long long)N * MS_D / 4;
  hipLaunchKernelGGL(pn_ms_combine_bwd_kernel, dim3(pn_cdiv(ND4, 256), B), dim3(256), 0, stream,
                     opart_q, opart_x, ND4, S, S2, gq, gx);
  PN_CHECK_LAUNCH();
  return PN_OK;
}
