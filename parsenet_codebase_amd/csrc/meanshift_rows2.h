// Row pass of the mean-shift backward (PASS 1 of meanshift_x3.h) with the work of a 32-row resident
// tile split between the TWO waves of a SIMD (round 4).
//
// The row pass holds two resident operands — q for S = X q^T and gu for T = X gu^T: 192 registers
// of split bf16 pieces — so the round-3 kernel runs ONE wave per SIMD in 512 registers, and nothing
// runs beside its elementwise stage or its LDS latencies: 59 % of its cycles issue MFMAs (the
// two-waves-per-SIMD column pass: 84 %).  Here a workgroup is 8 waves = 4 pairs; the waves of pair p
// (waves p and p + 4: the same SIMD) own the SAME 32 resident rows and share the tile:
//   role 0 keeps q  in registers and forms S,   role 1 keeps gu and forms T     (first GEMM: 1 unit each);
//   the second GEMM out += X^T W contracts over the 32 streamed points in two k-steps of 16: role r
//   takes k-step r.  It needs the weights W = K (T - c) alpha (src/mean_shift.py:45-79's gradient)
//   of ITS 16 streamed points only, i.e. half of S and half of T: the waves swap those halves
//   through LDS (2 KiB each way) and each evaluates HALF of the elementwise stage;
//   each wave accumulates its k-step into its own 32 x 128 accumulator, and the two are added
//   (role 0 + role 1, a fixed order) through LDS once per list fragment, when the rows are stored.
// A wave needs 96 operand registers instead of 192, two waves fit a SIMD, and both the matrix-pipe
// work (3 GEMM units = 144 MFMAs per resident tile, 72 per wave) and the elementwise work per tile
// are what the one-wave kernel does.  The products are the one-wave kernel's in another fixed order:
// the first GEMM adds small and large piece products into one accumulator (like the forward and the
// column pass), the second GEMM's output is the sum of two chains (k-step 0 and k-step 1 over all
// tiles) instead of one interleaved chain — fp32-grade and reproducible, not bit-identical to it.
// MEASURED AND NOT KEPT AS THE DEFAULT (PN_MS_ROWS2=1 selects it; profiles/r04_rows2_ab.txt, same box,
// alternating, B = 4 x 10 000, per launch): one-wave kernel 1.47 ms; a first version of this file (both
// waves evaluating the whole stage, second GEMM split by channels: bit-identical to the one-wave kernel)
// 1.59 ms — the duplicated stage is VALU time nothing hides while both waves of a SIMD sit in it; this
// version 1.96 ms — 256 registers with 9 spilled (scratch traffic in the tile loop waits behind the
// DMA), two 8-wave barriers per tile, and the waves of a SIMD still move in lockstep: both in the first
// GEMM, both in the stage, both in the second GEMM.  What the column pass gains from two waves per SIMD
// comes from running them half a tile APART (ping-pong); here each wave needs the other's half of the
// first GEMM before its stage, so a skew means running the first GEMM one tile ahead (a third image
// buffer, a second S / T accumulator live across the stage) — that does not fit 256 registers next to
// 96 operand and 64 output registers.
// LDS: two image buffers (48 KiB) + the exchange halves (16 KiB), reused for the pair sums; two
// workgroup barriers per tile.
#pragma once

#define R2_WAVES 8
__global__ __launch_bounds__(64 * R2_WAVES) __attribute__((amdgpu_waves_per_eu(2, 2))) void pn_ms3_rows2_kernel(
    const float* __restrict__ R, const float* __restrict__ R1, const u32x4* __restrict__ PA,
    const float* __restrict__ cs, const float* __restrict__ rs, const float* __restrict__ bsq_, int N, int ntiles,
    int tiles_per_slice, float* __restrict__ opart, const unsigned char* __restrict__ pairs,
    const int* __restrict__ lists, const int* __restrict__ offs, int nblk_total, int blk_off, int nbp, int nbB,
    int cmin, int sstride) {
  // 64 KiB: [0, 48 KiB) two tile images; [48, 64 KiB) exchange halves xch[pair][role][2][64] float4;
  // the whole block holds the pair sums psum[pair][16][64] float4 at the end of a fragment
  __shared__ __attribute__((aligned(16))) u32x4 lds_raw[4096];
  u32x4(*ldsP)[X3_IMG_U4] = reinterpret_cast<u32x4(*)[X3_IMG_U4]>(lds_raw);
  float4* xch = reinterpret_cast<float4*>(lds_raw + 2 * X3_IMG_U4);
  float4* psum = reinterpret_cast<float4*>(lds_raw);
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int pr = wave & 3, role = wave >> 2;
  const int col = lane & 31, h = lane >> 5;
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
  int nexec = 0;
  for (bool first_seg = true;; first_seg = false) {   // list fragments (exactly one without a flat plan)
    int b = blockIdx.z, rblk = blockIdx.y, slice = blockIdx.x;
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
      if (!first_seg) __syncthreads();  // role 0 has read the pair sums of the previous fragment
    } else {
      t_begin = slice * tiles_per_slice;
      t_end = min(ntiles, t_begin + tiles_per_slice);
    }
    const int i0 = (rblk * 4 + pr) * 32;
    const bool wave_on = i0 < N;
    const unsigned char* __restrict__ prow = nullptr;
    if (pairs) {
      const int wt = min(rblk * 4 + pr, ntiles - 1);
      prow = pairs + (size_t)b * ntiles * ntiles + (size_t)wt * ntiles;
    }
    const float bsqv = bsq_[b];
    const float hl = (0.5f / bsqv) * MS_LOG2E;
    const size_t bN = (size_t)b * N;
    const u32x4* __restrict__ PAb = PA + (size_t)b * ntiles * X3_IMG_U4;
    // a 24 KiB image = 24 chunks of 1 KiB (64 lanes x 16 B): three per wave
#define R2_STAGE(MT, BUF)                                                              \
  {                                                                                    \
    _Pragma("unroll") for (int u = 0; u < 3; ++u) {                                    \
      const int q_ = wave * 3 + u;                                                     \
      X3_GLDS16(PAb + (size_t)(MT) * X3_IMG_U4 + q_ * 64 + lane, &ldsP[BUF][q_ * 64]); \
    }                                                                                  \
  }
#define R2_TILE(E) (lst ? lst[E] : (E))
    int cur = 0;
    int mt_cur = t_begin < t_end ? R2_TILE(t_begin) : 0;
    int mt_nxt = t_begin + 1 < t_end ? R2_TILE(t_begin + 1) : 0;
    int on_cur = prow && t_begin < t_end ? (int)prow[mt_cur] : 1;
    if (t_begin < t_end) R2_STAGE(mt_cur, 0);
    // this wave's resident operand (q or gu) as B operand of the first GEMM: k-step s = channels 16 s + 8 h + e
    const int ires = min(i0 + col, N - 1);
    const float* __restrict__ Rw = role ? R1 : R;
    bf16x8 qh[8], qm[8], ql[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const float* src = Rw + (bN + ires) * MS_D + 16 * s + 8 * h;
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
    const float c_res = cs[bN + ires], a_res = rs[bN + ires];
    f32x16 acc_o[4];
#pragma unroll
    for (int fb = 0; fb < 4; ++fb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc_o[fb][r] = 0.f;

    for (int e_ = t_begin; e_ < t_end; ++e_) {
      const int mt = mt_cur;
      __syncthreads();   // image of tile mt landed; every wave is done with tile mt - 1 (image and exchange)
      int on_nxt = 1, mt_nn = 0;
      if (e_ + 1 < t_end) {
        if (prow) on_nxt = (int)prow[mt_nxt];
        if (e_ + 2 < t_end) mt_nn = R2_TILE(e_ + 2);
      }
      const bool pair_on = wave_on && on_cur != 0;     // the same for both waves of the pair
      const int mt_st = e_ + 1 < t_end ? mt_nxt : mt;
      f32x16 sa;
      if (pair_on) {
        if (role == 0) ++nexec;
        // ---- first GEMM: this wave's product X . (q or gu)^T; per k-step the three small piece products
        // first, then the large ones, into ONE accumulator (as the forward and the column pass do: a
        // second accumulator does not fit 256 registers next to the 64 output registers) ----
#pragma unroll
        for (int r = 0; r < 16; ++r) sa[r] = 0.f;
        const u32x4* __restrict__ lp = ldsP[cur];
        const int rowoff = col * 16, sw = x3_swz(col);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          const int slot = rowoff + ((2 * s + h) ^ sw);
          const bf16x8 ah = x3_as_bf16(lp[slot]);
          const bf16x8 am = x3_as_bf16(lp[X3_PIECE_U4 + slot]);
          const bf16x8 al = x3_as_bf16(lp[2 * X3_PIECE_U4 + slot]);
          X3_MFMA(sa, al, qh[s]);
          X3_MFMA(sa, ah, ql[s]);
          X3_MFMA(sa, am, qm[s]);
          X3_MFMA(sa, am, qh[s]);
          X3_MFMA(sa, ah, qm[s]);
          X3_MFMA(sa, ah, qh[s]);
          if (s < 3) {
            // the wave's three DMA pieces of the next image between the k-steps (unconditional: behind the
            // last tile of a fragment they re-read the current image into the free buffer, nobody reads it)
            const int q_ = wave * 3 + s;
            X3_GLDS16(PAb + (size_t)mt_st * X3_IMG_U4 + q_ * 64 + lane, &ldsP[cur ^ 1][q_ * 64]);
          }
        }
        // the half the sibling needs: role 0 hands out S of k-step 1 (registers 8..15), role 1 T of k-step 0
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int r0 = 8 * (role ^ 1) + 4 * q;
          xch[((pr * 2 + role) * 2 + q) * 64 + lane] = make_float4(sa[r0], sa[r0 + 1], sa[r0 + 2], sa[r0 + 3]);
        }
      } else if (e_ + 1 < t_end) {
        R2_STAGE(mt_nxt, cur ^ 1);
      }
      __syncthreads();   // the exchange halves of every pair are in LDS
      if (pair_on) {
        float ot[8];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const float4 v = xch[((pr * 2 + (role ^ 1)) * 2 + q) * 64 + lane];
          ot[4 * q] = v.x;
          ot[4 * q + 1] = v.y;
          ot[4 * q + 2] = v.z;
          ot[4 * q + 3] = v.w;
        }
        // ---- this wave's half of the elementwise stage: D registers 8 role .. 8 role + 7, i.e. the
        // streamed points of k-step `role` of the second GEMM ----
        float gs[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const float own = sa[8 * role + r];
          const float sv = role ? ot[r] : own;
          const float tv = role ? own : ot[r];
          const float dist = __builtin_fmaf(-2.0f, sv, 2.0f);
          const float a2 = -dist * hl;
          const float a2c = __builtin_amdgcn_fmed3f(a2, -MS_LIM2, MS_LIM2);
          const float k = __builtin_amdgcn_exp2f(a2c);
          const float d_ = (tv - c_res) * a_res;
          float g = k * d_;
          asm("" : "+v"(g));          /* keep the select a v_cndmask, not a branch */
          gs[r] = a2c == a2 ? g : 0.f;
        }
        u32x4 wh, wm, wl;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const X3Pieces p_ = x3_split2_t<false>(gs[2 * q], gs[2 * q + 1]);
          wh[q] = p_.h;
          wm[q] = p_.m;
          wl[q] = p_.l;
        }
        // ---- second GEMM, k-step `role`, all four feature blocks ----
        const char* lbase = reinterpret_cast<const char*>(ldsP[cur]);
        const int li = lane & 15, rb = 4 * h + (li >> 2), cb = 16 * ((lane >> 4) & 1) + 4 * (li & 3);
        const int sz0 = (((li >> 2) & 3) << 2) | (h & 3), sz1 = (((li >> 2) & 3) << 2) | ((h + 2) & 3);
#define R2_TR(P_, W, FB)                                                                        \
  __builtin_amdgcn_ds_read_tr16_b64_v4i16((x3_lds_s16x4)(                                        \
      lbase + (P_) * (X3_PIECE_U4 * 16) + (16 * role + 8 * (W) + rb) * 256 +                    \
      ((((FB) * 4 + (cb >> 3)) ^ ((W) ? sz1 : sz0)) << 4) + ((cb & 7) << 1)))
        const bf16x8 bh = x3_as_bf16(wh), bm = x3_as_bf16(wm), bl = x3_as_bf16(wl);
#pragma unroll
        for (int fb = 0; fb < 4; ++fb) {
          u32x4 xc[3];
#pragma unroll
          for (int p_ = 0; p_ < 3; ++p_) {
            const s16x4 lo_ = R2_TR(p_, 0, fb), hi_ = R2_TR(p_, 1, fb);
            const s16x8 v_ = __builtin_shufflevector(lo_, hi_, 0, 1, 2, 3, 4, 5, 6, 7);
            xc[p_] = __builtin_bit_cast(u32x4, v_);
          }
          const bf16x8 xh = x3_as_bf16(xc[0]), xm = x3_as_bf16(xc[1]), xl = x3_as_bf16(xc[2]);
          X3_MFMA(acc_o[fb], xl, bh);
          X3_MFMA(acc_o[fb], xh, bl);
          X3_MFMA(acc_o[fb], xm, bm);
          X3_MFMA(acc_o[fb], xm, bh);
          X3_MFMA(acc_o[fb], xh, bm);
          X3_MFMA(acc_o[fb], xh, bh);
        }
#undef R2_TR
      }
      cur ^= 1;
      mt_cur = mt_nxt;
      mt_nxt = mt_nn;
      on_cur = on_nxt;
    }
#undef R2_TILE
#undef R2_STAGE
    // ---- rows of the fragment: role 0's k-step-0 sums + role 1's k-step-1 sums, added through LDS ----
    __syncthreads();   // every wave is done with the images (and every DMA piece has landed: the barrier's vmcnt(0))
    if (role == 1) {
#pragma unroll
      for (int fb = 0; fb < 4; ++fb)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          psum[((pr * 4 + fb) * 4 + g) * 64 + lane] =
              make_float4(acc_o[fb][4 * g], acc_o[fb][4 * g + 1], acc_o[fb][4 * g + 2], acc_o[fb][4 * g + 3]);
    }
    __syncthreads();
    const int ir = i0 + col;
    if (role == 0 && wave_on && ir < N) {
      float* o = opart + (((size_t)b * S + slice) * N + ir) * MS_D;
#pragma unroll
      for (int fb = 0; fb < 4; ++fb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 v = psum[((pr * 4 + fb) * 4 + g) * 64 + lane];
          *reinterpret_cast<float4*>(o + fb * 32 + 8 * g + 4 * h) =
              make_float4(acc_o[fb][4 * g] + v.x, acc_o[fb][4 * g + 1] + v.y, acc_o[fb][4 * g + 2] + v.z,
                          acc_o[fb][4 * g + 3] + v.w);
        }
    }
    if (!flat || e_lo >= e_hi) break;
    ++fblk;
    while (offs[fblk + 1] <= e_lo) ++fblk;  // empty lists
  }
  if (lane == 0 && nexec) atomicAdd(&pn_ms3_exec[1], (unsigned long long)nexec);
}
