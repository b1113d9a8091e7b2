// Operand-split helpers shared by the 16-bit matrix-core kernels (meanshift_x3.h, meanshift_h2.h,
// the fp16 x 2 selection pass of knn_mfma.hip): vector typedefs, the LDS-DMA wrapper, the chunk
// swizzle of the tile images and the scaled two-piece fp16 split.
#pragma once
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4* x3_lds_s16x4;

typedef const __attribute__((address_space(1))) void* x3_gptr;
typedef __attribute__((address_space(3))) void* x3_lptr;
#define X3_GLDS16(G, L) __builtin_amdgcn_global_load_lds((x3_gptr)(G), (x3_lptr)(L), 16, 0, 0)

// chunk swizzle of the tile images: chunk c (8 channels) of row j is stored at c ^ x3_swz(j)
__host__ __device__ static inline int x3_swz(int j) { return ((j & 3) << 2) | ((j >> 2) & 3); }

// ---- fp16 x 2 (meanshift_h2.h has the scaling rules) ----
#define H2_IMG_U4 1024            // uint4 (16 B) units per 16 KiB tile image
#define H2_PIECE_U4 512           // per piece
#define H2_SX 4096.0f             // 2^12: unit rows
#define H2_ISX2 0x1p-24f          // 1 / H2_SX^2

struct H2Pieces {
  uint32_t h, m;
};
__device__ static inline H2Pieces h2_split2(float a, float b) {
  f32x2 v = {a, b};
  f16x2 ph = __builtin_convertvector(v, f16x2);
  // a - float(h) (exact) as ONE v_fma_mix_f32 that reads the fp16 half in place, instead of
  // v_cvt_f32_f16 + v_sub_f32: fma(half, -1.0f, a); op_sel picks the half, op_sel_hi marks src0 as fp16
  const uint32_t hb = __builtin_bit_cast(uint32_t, ph);
  f32x2 r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r[0]) : "v"(hb), "v"(a));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r[1]) : "v"(hb), "v"(b));
  f16x2 pm = __builtin_convertvector(r, f16x2);
  H2Pieces o;
  o.h = __builtin_bit_cast(uint32_t, ph);
  o.m = __builtin_bit_cast(uint32_t, pm);
  return o;
}
#define H2_SPLIT_TO(A, B, VH, VM, Q)     \
  {                                      \
    const H2Pieces _p = h2_split2(A, B); \
    VH[Q] = _p.h;                        \
    VM[Q] = _p.m;                        \
  }

__device__ static inline f16x8 h2_as_f16(u32x4 v) { return __builtin_bit_cast(f16x8, v); }

#define H2_MFMA(ACC, A, B) ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, ACC, 0, 0, 0)

// ---- bf16 x 3 ----
// error-free split of two floats into three packed bf16 pairs (element 0 in the low half)
struct X3Pieces {
  uint32_t h, m, l;
};
// PACKED: the two residuals as v_pk_add_f32 (one instruction per pair); otherwise as SCALAR
// subtractions (the empty asm keeps the compiler from packing them again): beside MFMAs a packed
// fp32 VALU instruction costs ~13 cycles more than a plain one (MI355X_MICROARCH.md).  Measured in
// the mean-shift passes (tools/jobs/r3zb.sh, r3zc.sh; profiles/r03_pingpong_ab.txt): scalar -4 / -11 %
// cycles per tile in the two 8-wave passes (two waves per SIMD: the VALU work of one runs beside the
// MFMAs of the other), +2.4 % in the one-wave-per-SIMD row pass (one more issue slot per pair,
// nothing beside it) — so the callers choose.  Same arithmetic either way.
template <bool PACKED>
__device__ static inline X3Pieces x3_split2_t(float a, float b) {
  f32x2 v = {a, b};
  bf16x2 ph = __builtin_convertvector(v, bf16x2);
  f32x2 r, r2;
  if (PACKED) {
    r = f32x2{a - (float)ph[0], b - (float)ph[1]};
  } else {
    float r0 = a - (float)ph[0], r1 = b - (float)ph[1];
    asm("" : "+v"(r0));
    r = f32x2{r0, r1};
  }
  bf16x2 pm = __builtin_convertvector(r, bf16x2);
  if (PACKED) {
    r2 = f32x2{r[0] - (float)pm[0], r[1] - (float)pm[1]};
  } else {
    float s0 = r[0] - (float)pm[0], s1 = r[1] - (float)pm[1];
    asm("" : "+v"(s0));
    r2 = f32x2{s0, s1};
  }
  bf16x2 pl = __builtin_convertvector(r2, bf16x2);
  X3Pieces o;
  o.h = __builtin_bit_cast(uint32_t, ph);
  o.m = __builtin_bit_cast(uint32_t, pm);
  o.l = __builtin_bit_cast(uint32_t, pl);
  return o;
}
__device__ static inline X3Pieces x3_split2(float a, float b) { return x3_split2_t<true>(a, b); }
#define X3_SPLIT_TO(A, B, VH, VM, VL, Q) \
  {                                      \
    const X3Pieces _p = x3_split2(A, B); \
    VH[Q] = _p.h;                        \
    VM[Q] = _p.m;                        \
    VL[Q] = _p.l;                        \
  }

__device__ static inline bf16x8 x3_as_bf16(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

