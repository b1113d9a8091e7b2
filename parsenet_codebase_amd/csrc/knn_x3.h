// Pass 1 of the selection engine (tile maxima -> threshold) on the bf16 matrix cores.
// Included by knn_mfma.hip.
//
// The engine evaluates all Nq x Nc values twice: pass 1 for the per-16-candidate maxima that give
// every query a threshold tau with at least k values >= tau, pass 2 to collect the values >= tau.
// Only pass 2 decides the result: its values are the fp32 fma chains the oracle defines.  Pass 1
// merely has to produce a threshold that is certainly not above the k-th largest exact value, so
// it may use any arithmetic with a known error bound:
//   operands split error-free into three bf16 pieces (24 mantissa bits), 6 of the 9 piece products
//   on v_mfma_f32_32x32x16_bf16 (2.5 PFLOP/s dense instead of 157 TFLOP/s fp32), fp32 accumulate:
//     |dot~ - dot| <= A |q| |c|,  A = 16 (C + 4) 2^-24
//   (per fp32 accumulation at most one ulp of a partial sum <= 1.02 |q||c|, 6 C / 16 MFMAs of 16
//   terms; the three dropped products are below 3 * 2^-24 |q||c|; the same bound again covers the
//   rounding of the exact chain itself, and the factor 16 instead of 7 leaves room for truncating
//   instead of rounding accumulators).
//   MODE 0 (v = 2 dot - |c|^2 - |q|^2, norms shared with the exact pass):
//     eps_q = 2 A sqrt(|q|^2 max|c|^2) + 2^-21 (|q|^2 + max|c|^2)
//   MODE 2 (v = dot):  eps_q = A sqrt(|q|^2 max|c|^2)
// tau~ = threshold of the approximate tile maxima: k different candidates have v~ >= tau~, hence
// exact v >= tau~ - eps_q: pass 2 (exact) collects v >= tau~ - eps_q and loses nothing.  The margin
// adds a handful of survivors (eps is ~1e-4 of the value range).
#pragma once

// chunk swizzle of the candidate images: rows of NCH 16-byte chunks (8 channels each)
template <int NCH>
__host__ __device__ static inline int knx_slot(int r, int q) {
  return r * NCH + (q ^ (NCH == 8 ? ((r >> 1) & 7) : x3_swz(r)));
}

// xp (B, 8 NCH, Np) channel-first permuted fp32 (pn_knn_prep_kernel) -> images
// [B][Np/32][piece 3][32 rows x NCH chunks] and the largest squared norm of each batch item
template <int NCH>
__global__ __launch_bounds__(256) void pn_knn_x3_image_kernel(const float* __restrict__ xp,
                                                              const float* __restrict__ xxp, int Np,
                                                              u32x4* __restrict__ img,
                                                              unsigned* __restrict__ xxmax) {
  constexpr int CP = 8 * NCH;
  const int b = blockIdx.y, t = blockIdx.x, tid = threadIdx.x;
  const int r = tid & 31;
  const float* src = xp + (size_t)b * CP * Np + (size_t)t * 32 + r;
  u32x4* dst = img + ((size_t)b * (Np / 32) + t) * (3 * 32 * NCH);
  for (int q = tid >> 5; q < NCH; q += 8) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = src[(size_t)(8 * q + e) * Np];
    u32x4 vh, vm, vl;
    X3_SPLIT_TO(v[0], v[1], vh, vm, vl, 0);
    X3_SPLIT_TO(v[2], v[3], vh, vm, vl, 1);
    X3_SPLIT_TO(v[4], v[5], vh, vm, vl, 2);
    X3_SPLIT_TO(v[6], v[7], vh, vm, vl, 3);
    const int slot = knx_slot<NCH>(r, q);
    dst[slot] = vh;
    dst[32 * NCH + slot] = vm;
    dst[2 * 32 * NCH + slot] = vl;
  }
  if (tid < 64) {
    const float m = pn_wave_max(tid < 32 ? xxp[(size_t)b * Np + (size_t)t * 32 + tid] : 0.f);
    if (tid == 0) atomicMax(&xxmax[b], __float_as_uint(fmaxf(m, 0.f)));   // norms are >= 0: uint order
  }
}

// tile maxima of the approximate values; grid (slices, blocks of 128 QSETS queries, B), 256 threads:
// four waves (one per SIMD) and at most 48 KiB of LDS, so that TWO workgroups share a CU: the
// prologue of one (64 strided query loads, their split, the first DMA) and its launch gap run under
// the MFMAs of the other, and the two waves of a SIMD are not in lock step.  (One 8-wave workgroup
// per CU measured 63 % wave residency and 38 % MFMA utilisation.)
// A step of the loop handles TPS consecutive tiles (one barrier, one batch of LDS DMA).
#define KX_NW 4
// (the 128-channel dot-product variant needs 165 registers: three workgroups per CU)
#define KX_WPE(NCH, MODE) ((NCH) == 16 && (MODE) == 2 ? 3 : 2)
template <int NCH, int QSETS, int MODE, int TPS>
__global__ __launch_bounds__(64 * KX_NW) __attribute__((amdgpu_waves_per_eu(KX_WPE(NCH, MODE), KX_WPE(NCH, MODE)))) void pn_knn_x3_pass1_kernel(
    const float* __restrict__ xq, const float* __restrict__ xxq_, int Nq, int Nqp,
    const u32x4* __restrict__ PC, const float* __restrict__ xxc_, int Nc, int Ncp, int tiles_per_slice,
    float* __restrict__ tilemax) {
  constexpr int CP = 8 * NCH, KS = NCH / 2, PIECE = 32 * NCH, IMG = 3 * PIECE;
  constexpr int CHUNKS = TPS * IMG / 64;   // 1 KiB DMA chunks per step
  static_assert(CHUNKS % KX_NW == 0 && (TPS == 1 || TPS == 2 || TPS == 4), "chunks are dealt evenly to the waves");
  __shared__ __attribute__((aligned(16))) u32x4 ldsP[2][TPS * IMG];
  __shared__ __attribute__((aligned(16))) float lds_xx[2][TPS * 32 < 64 ? 64 : TPS * 32];
  const int b = blockIdx.z;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int col = lane & 31, h = lane >> 5;
  const int q0 = (blockIdx.y * KX_NW + wave) * (32 * QSETS);
  const bool wave_on = q0 < Nq;
  const int ntiles = Ncp / 32;
  const int slice = blockIdx.x;
  const int t_begin = slice * tiles_per_slice;           // a multiple of TPS
  const int t_end = min(ntiles, t_begin + tiles_per_slice);
  const int T16 = Ncp / 16;
  const u32x4* __restrict__ PCb = PC + (size_t)b * ntiles * IMG;
  const float* __restrict__ xqb = xq + (size_t)b * CP * Nqp;
  const float* __restrict__ xxcb = xxc_ + (size_t)b * Ncp;

  // the images of consecutive tiles are contiguous: a step's batch is one linear copy
#define KX_STAGE(M0, BUF)                                                                       \
  {                                                                                             \
    const int nch_ = min(TPS, t_end - (M0)) * (IMG / 64);                                       \
    _Pragma("unroll") for (int u_ = 0; u_ < CHUNKS / KX_NW; ++u_) {                             \
      const int c_ = wave + KX_NW * u_;                                                         \
      if (c_ < nch_) X3_GLDS16(PCb + (size_t)(M0) * IMG + c_ * 64 + lane, &ldsP[BUF][c_ * 64]); \
    }                                                                                           \
    /* squared norms of the step's candidates: 64 per instruction */                            \
    if (MODE == 0 && wave < (TPS + 1) / 2 && (M0) + 2 * wave < t_end && (TPS > 1 || lane < 32)) \
      __builtin_amdgcn_global_load_lds((x3_gptr)(xxcb + ((M0) + 2 * wave) * 32 + lane),         \
                                       (x3_lptr)(&lds_xx[BUF][64 * wave]), 4, 0, 0);            \
  }
  int cur = 0;
  if (t_begin < t_end) KX_STAGE(t_begin, 0);
  // resident queries as B operands: k-step s = channels 16 s + 8 h + e
  bf16x8 qh[QSETS][KS], qm[QSETS][KS], ql[QSETS][KS];
  float xxq[QSETS];
#pragma unroll
  for (int u = 0; u < QSETS; ++u) {
    const int q = q0 + 32 * u + col;
    const int qcl = q < Nqp ? q : Nqp - 1;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = xqb[(size_t)(16 * s + 8 * h + e) * Nqp + qcl];
      u32x4 vh, vm, vl;
      X3_SPLIT_TO(v[0], v[1], vh, vm, vl, 0);
      X3_SPLIT_TO(v[2], v[3], vh, vm, vl, 1);
      X3_SPLIT_TO(v[4], v[5], vh, vm, vl, 2);
      X3_SPLIT_TO(v[6], v[7], vh, vm, vl, 3);
      qh[u][s] = x3_as_bf16(vh);
      qm[u][s] = x3_as_bf16(vm);
      ql[u][s] = x3_as_bf16(vl);
    }
    xxq[u] = MODE == 0 ? xxq_[(size_t)b * Nqp + qcl] : 0.f;
  }

  for (int m0 = t_begin; m0 < t_end; m0 += TPS) {
    __syncthreads();  // the batch of step m0 landed; every wave is done with the previous one
    if (m0 + TPS < t_end) KX_STAGE(m0 + TPS, cur ^ 1);
    if (wave_on) {
      float tmv[QSETS][TPS];
#pragma unroll
      for (int u = 0; u < QSETS; ++u)
#pragma unroll
        for (int i = 0; i < TPS; ++i) tmv[u][i] = -__builtin_inff();
      // (not unrolled: overlapping the tiles of a step costs more registers than the two waves per
      // SIMD leave)
#pragma clang loop unroll(disable)
      for (int i = 0; i < TPS; ++i) {
        const int j0 = (m0 + i) * 32;
        if (m0 + i < t_end) {
          // one query set: two accumulators (small / large piece products) keep dependent MFMAs
          // apart; two sets already interleave
          constexpr int NACS = QSETS == 1 ? 1 : 0;
          f32x16 acc[QSETS], acs_[NACS + 1];
#define KX_ACS(U) (QSETS == 1 ? acs_[0] : acc[U])
#pragma unroll
          for (int u = 0; u < QSETS; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              acc[u][r] = 0.f;
              if (QSETS == 1) acs_[0][r] = 0.f;
            }
          const u32x4* __restrict__ lp = ldsP[cur] + i * IMG;
#pragma unroll
          for (int s = 0; s < KS; ++s) {
            const int slot = knx_slot<NCH>(col, 2 * s + h);
            const bf16x8 ah = x3_as_bf16(lp[slot]);
            const bf16x8 am = x3_as_bf16(lp[PIECE + slot]);
            const bf16x8 al = x3_as_bf16(lp[2 * PIECE + slot]);
#pragma unroll
            for (int u = 0; u < QSETS; ++u) {
              KX_ACS(u) = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, qh[u][s], KX_ACS(u), 0, 0, 0);
              acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, qh[u][s], acc[u], 0, 0, 0);
              KX_ACS(u) = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, ql[u][s], KX_ACS(u), 0, 0, 0);
              acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, qm[u][s], acc[u], 0, 0, 0);
              KX_ACS(u) = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, qm[u][s], KX_ACS(u), 0, 0, 0);
              acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, qh[u][s], acc[u], 0, 0, 0);
            }
          }
          // D[candidate = (r&3) + 8(r>>2) + 4h][query = col]
          float xxj[16];
          if (MODE == 0) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const float4 t4 = *reinterpret_cast<const float4*>(&lds_xx[cur][32 * i + 8 * g + 4 * h]);
              xxj[4 * g + 0] = t4.x;
              xxj[4 * g + 1] = t4.y;
              xxj[4 * g + 2] = t4.z;
              xxj[4 * g + 3] = t4.w;
            }
          }
          const bool tail = j0 + 32 > Nc;
#pragma unroll
          for (int u = 0; u < QSETS; ++u) {
            float tm = -__builtin_inff();
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              float v = QSETS == 1 ? acc[u][r] + acs_[0][r] : acc[u][r];
              if (MODE == 0) v = __builtin_fmaf(2.0f, v, -xxj[r]) - xxq[u];
              if (tail && j0 + (r & 3) + 8 * (r >> 2) + 4 * h >= Nc) v = -__builtin_inff();
              tm = fmaxf(tm, v);
            }
#pragma unroll
            for (int i2 = 0; i2 < TPS; ++i2)
              if (i2 == i) tmv[u][i2] = tm;   // static register indices
          }
#undef KX_ACS
        }
      }
      // tilemax[q][2 mt + h]: the two halves of a tile sit in lanes col and col + 32; pair them
      // so that a lane writes whole tiles (16 bytes: two tiles, or 8 bytes: one)
#pragma unroll
      for (int u = 0; u < QSETS; ++u) {
        const int q = q0 + 32 * u + col;
        float* row = tilemax + ((size_t)b * Nqp + (q < Nqp ? q : 0)) * T16 + 2 * m0;
        float o[TPS];
#pragma unroll
        for (int i = 0; i < TPS; ++i) o[i] = __shfl_xor(tmv[u][i], 32, 64);
        if (TPS == 4) {
          // ntiles, t_begin are even: a step ends after 2 or 4 tiles
          if (q < Nqp && m0 + 2 * h < t_end)
            *reinterpret_cast<float4*>(row + 4 * h) =
                h == 0 ? make_float4(tmv[u][0], o[0], tmv[u][1], o[1])
                       : make_float4(o[TPS - 2], tmv[u][TPS - 2], o[TPS - 1], tmv[u][TPS - 1]);
        } else if (TPS == 2) {
          if (q < Nqp && m0 + h < t_end)
            *reinterpret_cast<float2*>(row + 2 * h) =
                h == 0 ? make_float2(tmv[u][0], o[0]) : make_float2(o[TPS - 1], tmv[u][TPS - 1]);
        } else {
          if (q < Nqp && h == 0) *reinterpret_cast<float2*>(row) = make_float2(tmv[u][0], o[0]);
        }
      }
    }
    cur ^= 1;
  }
#undef KX_STAGE
}

// tau <- tau - eps_q (see the header of this file); one thread per query
__global__ void pn_knn_x3_margin_kernel(float* __restrict__ tau, const float* __restrict__ xxq, int Nq, int Nqp,
                                        const unsigned* __restrict__ xxmax, float A, int mode) {
  const int b = blockIdx.y;
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= Nq) return;
  const float nq = xxq[(size_t)b * Nqp + q], nc = __uint_as_float(xxmax[b]);
  const float cross = sqrtf(nq * nc) * 1.000001f;
  const float eps = mode == 0 ? 2.0f * A * cross + 0x1p-21f * (nq + nc) : A * cross;
  const float t = tau[(size_t)b * Nqp + q];
  // round down: one more ulp of |t| + eps on top
  tau[(size_t)b * Nqp + q] = t - eps * 1.0001f - 0x1p-22f * fabsf(t);
}
