// Mean-shift backward restricted to a few rows (round 4).
//
// One mean-shift step (src/mean_shift.py:45-79) maps row i of the iterate to
//   y_i = u_i / ||u_i||,  u_i = sum_j K_ij x_j / r_i,  K_ij = exp(clamp((q_i.x_j - 1) / b^2, +-75)),  r_i = sum_j K_ij:
// row i of the result depends on row i of the previous iterate and on the data X — on no other row.
// The training path reads the final iterate only at the cluster centres the NMS picked
// (centres = new_X[indices], src/mean_shift.py:36-43; <= 64 rows of 10 000), so the gradient that
// reaches the iterations is zero outside those rows and STAYS zero there on the way back through
// all ten steps: the backward pass needs the R centre rows against the N data points, R x N
// kernel values per step instead of N x N.  (The dense passes of meanshift_x3.h serve callers whose
// gradient is dense.)
//
// Per step, for the R rows (gathered into compact (B,R,D) arrays by the caller) — meanshift.hip's formulas:
//   gu = (gy - y (y.gy)) / ||u|| ; c = gu.u ; alpha = 1 / (r b^2)
//   gs_ij = K_ij (gu_i.x_j - c_i) alpha_i                 (zero where the clamp is active)
//   gq_i  = sum_j gs_ij x_j                                -> gradient w.r.t. the previous iterate's row
//   gX_j += sum_i gs_ij q_i + sum_i K_ij gu_i / r_i        -> gradient w.r.t. the data
// Arithmetic: plain fp32 fma on the vector ALUs (the whole backward of a cfg5 step is 26 GFLOP this
// way; the dense passes it replaces executed 2 x 10 launches of ~1 TFLOP each).
//
// pn_ms_rows_prep_kernel: gu, c, alpha, 1 / r of the R rows, once per step.
// pn_ms_rows_bwd_kernel: one workgroup per block of 64 data points j and batch item: stages the 64
// x rows, the R q rows and the R gu rows in LDS, forms the 64 x 64 blocks of
// q.x and gu.x, the kernel values, gs and K / r, then its own 64 rows of gX (exclusive owner: read,
// add, write) and its partial of gq, which pn_ms_rows_reduce_kernel adds over the blocks in fixed
// order.  No atomics: results are bit-reproducible.
// Measured in round 4 (cfg5: 4 shapes x 10 000 points, 64 rows, same box, alternating): 96 us per step of the backward.  The
// same block on the fp32 matrix cores (v_mfma_f32_32x32x2_f32 for all four products: 20 000 instead of
// 50 000 cycles of arithmetic per workgroup) took 101 us: with 150 KiB of staged operands there is one
// 4-wave workgroup per CU and nothing hides its load -> barrier -> product -> barrier -> read-add-write chain;
// the matrix-core version was removed again.
#include "common.h"
#include <cstdlib>

#define MR_D 128
#define MR_R 64        // rows per batch item (zero padded)
#define MR_CB 64       // data points per workgroup
#define MR_LD 132      // row stride of the staged operands (floats): 16-byte aligned, 4 banks apart
#define MR_LG 68       // row stride of the 64 x 64 blocks
#define MR_LOG2E 1.4426950408889634f
#define MR_LIM2 (75.0f * MR_LOG2E)

__device__ static inline float4 mr_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// gu, c, alpha and 1 / r of the R rows, one wave per row (as pn_ms_prep_bwd_kernel): once per step, not once
// per workgroup of the main kernel (there the sixteen rows of a wave were sixteen dependent round trips to
// memory: 60 % of the launch)
__global__ __launch_bounds__(256) void pn_ms_rows_prep_kernel(const float* __restrict__ gy, const float* __restrict__ y,
                                                              const float* __restrict__ rsum, const float* __restrict__ unorm,
                                                              const float* __restrict__ bsq, int R, float* __restrict__ gu,
                                                              float* __restrict__ scal) {
  const int b = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + wave;
  if (r >= MR_R) return;
  float u0 = 0.f, u1 = 0.f, c = 0.f, al = 0.f, ri = 0.f;
  if (r < R) {
    const size_t base = ((size_t)b * R + r) * MR_D;
    const float y0 = y[base + lane], y1 = y[base + lane + 64];
    const float g0 = gy[base + lane], g1 = gy[base + lane + 64];
    const float nn = unorm[(size_t)b * R + r], rr = rsum[(size_t)b * R + r];
    const float yg = pn_wave_sum(y0 * g0 + y1 * g1);
    u0 = (g0 - y0 * yg) / nn;
    u1 = (g1 - y1 * yg) / nn;
    c = pn_wave_sum(u0 * (y0 * nn) + u1 * (y1 * nn));
    al = 1.0f / (rr * bsq[b]);
    ri = 1.0f / rr;
  }
  float* g = gu + ((size_t)b * MR_R + r) * MR_D;
  g[lane] = u0;
  g[lane + 64] = u1;
  if (lane == 0) {
    float* sp = scal + ((size_t)b * MR_R + r) * 4;
    sp[0] = c;
    sp[1] = al;
    sp[2] = ri;
    sp[3] = 0.f;
  }
}

__global__ __launch_bounds__(256) void pn_ms_rows_bwd_kernel(
    const float* __restrict__ gu, const float* __restrict__ scal, const float* __restrict__ q,
    const float* __restrict__ x, const float* __restrict__ bsq, int N, int R, int nblk, float* __restrict__ gx,
    float* __restrict__ gq_part) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Xs = lds;                          // [64][MR_LD]
  float* Qs = Xs + MR_CB * MR_LD;           // [64][MR_LD]
  float* Us = Qs + MR_R * MR_LD;            // [64][MR_LD]  gu
  float* GS = Us + MR_R * MR_LD;            // [i][j]  gs
  float* KR = GS + MR_R * MR_LG;            // [i][j]  K / r_i
  float* GT = KR + MR_R * MR_LG;            // [j][i]  gs transposed
  float* sc = GT + MR_CB * MR_LG;           // c_i
  float* sa = sc + MR_R;                    // alpha_i
  float* sr = sa + MR_R;                    // 1 / r_i
  const int b = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;

  const int j0 = blk * MR_CB;
  const float bs = bsq[b];
  const float hl = (0.5f / bs) * MR_LOG2E;
  const float* __restrict__ xb = x + (size_t)b * N * MR_D;

  // ---- stage: x rows (zero past N), q rows, gu rows + the per-row scalars
  for (int it = tid; it < MR_CB * (MR_D / 4); it += 256) {
    const int r = it >> 5, c4 = it & 31;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j0 + r < N) v = mr_ld4(xb + (size_t)(j0 + r) * MR_D + 4 * c4);
    *reinterpret_cast<float4*>(Xs + r * MR_LD + 4 * c4) = v;
    float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < R) w = mr_ld4(q + ((size_t)b * R + r) * MR_D + 4 * c4);
    *reinterpret_cast<float4*>(Qs + r * MR_LD + 4 * c4) = w;
  }
  for (int it = tid; it < MR_R * (MR_D / 4); it += 256) {      // gu rows and the per-row scalars (pn_ms_rows_prep_kernel)
    const int r = it >> 5, c4 = it & 31;
    *reinterpret_cast<float4*>(Us + r * MR_LD + 4 * c4) = mr_ld4(gu + ((size_t)b * MR_R + r) * MR_D + 4 * c4);
  }
  if (tid < MR_R) {
    const float4 sv = mr_ld4(scal + ((size_t)b * MR_R + tid) * 4);
    sc[tid] = sv.x;
    sa[tid] = sv.y;
    sr[tid] = sv.z;
  }
  __syncthreads();

  // ---- S = q.x, T = gu.x for rows i = ti + 16 a, columns j = tj + 16 e (a, e = 0..3)
  {
    const int ti = tid >> 4, tj = tid & 15;
    float s[4][4], t[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s[a][e] = 0.f;
        t[a][e] = 0.f;
      }
    for (int k = 0; k < MR_D; k += 4) {
      float4 qv[4], uv[4], xv[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        qv[a] = mr_ld4(Qs + (ti + 16 * a) * MR_LD + k);
        uv[a] = mr_ld4(Us + (ti + 16 * a) * MR_LD + k);
        xv[a] = mr_ld4(Xs + (tj + 16 * a) * MR_LD + k);
      }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s[a][e] = __builtin_fmaf(qv[a].x, xv[e].x, s[a][e]);
          s[a][e] = __builtin_fmaf(qv[a].y, xv[e].y, s[a][e]);
          s[a][e] = __builtin_fmaf(qv[a].z, xv[e].z, s[a][e]);
          s[a][e] = __builtin_fmaf(qv[a].w, xv[e].w, s[a][e]);
          t[a][e] = __builtin_fmaf(uv[a].x, xv[e].x, t[a][e]);
          t[a][e] = __builtin_fmaf(uv[a].y, xv[e].y, t[a][e]);
          t[a][e] = __builtin_fmaf(uv[a].z, xv[e].z, t[a][e]);
          t[a][e] = __builtin_fmaf(uv[a].w, xv[e].w, t[a][e]);
        }
    }
    // kernel values (meanshift.hip's MS_EW): exp2 of the clamped argument, zero past N
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int i = ti + 16 * a;
      const float ci = sc[i], ai = sa[i], ri = sr[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int j = tj + 16 * e;
        const float dist = __builtin_fmaf(-2.0f, s[a][e], 2.0f);
        const float a2 = -dist * hl;
        const float a2c = __builtin_amdgcn_fmed3f(a2, -MR_LIM2, MR_LIM2);
        float kv = __builtin_amdgcn_exp2f(a2c);
        if (j0 + j >= N) kv = 0.f;
        const float g = a2c == a2 ? kv * ((t[a][e] - ci) * ai) : 0.f;
        GS[i * MR_LG + j] = g;
        GT[j * MR_LG + i] = g;
        KR[i * MR_LG + j] = kv * ri;
      }
    }
  }
  __syncthreads();

  const int f4 = tid & 31, grp = tid >> 5;    // 4 features, 8 rows or columns per thread
  // ---- gX rows j = 8 grp .. 8 grp + 7: sum_i gs_ij q_i + (K_ij / r_i) gu_i
  {
    float4 acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = 0; i < MR_R; ++i) {
      const float4 qv = mr_ld4(Qs + i * MR_LD + 4 * f4);
      const float4 uv = mr_ld4(Us + i * MR_LD + 4 * f4);
      const float4 g0 = mr_ld4(GS + i * MR_LG + 8 * grp), g1 = mr_ld4(GS + i * MR_LG + 8 * grp + 4);
      const float4 k0 = mr_ld4(KR + i * MR_LG + 8 * grp), k1 = mr_ld4(KR + i * MR_LG + 8 * grp + 4);
      const float gv[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
      const float kv[8] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        acc[e].x = __builtin_fmaf(gv[e], qv.x, acc[e].x);
        acc[e].y = __builtin_fmaf(gv[e], qv.y, acc[e].y);
        acc[e].z = __builtin_fmaf(gv[e], qv.z, acc[e].z);
        acc[e].w = __builtin_fmaf(gv[e], qv.w, acc[e].w);
        acc[e].x = __builtin_fmaf(kv[e], uv.x, acc[e].x);
        acc[e].y = __builtin_fmaf(kv[e], uv.y, acc[e].y);
        acc[e].z = __builtin_fmaf(kv[e], uv.z, acc[e].z);
        acc[e].w = __builtin_fmaf(kv[e], uv.w, acc[e].w);
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int j = j0 + 8 * grp + e;
      if (j < N) {
        float4* o = reinterpret_cast<float4*>(gx + ((size_t)b * N + j) * MR_D + 4 * f4);
        float4 v = *o;
        v.x += acc[e].x;
        v.y += acc[e].y;
        v.z += acc[e].z;
        v.w += acc[e].w;
        *o = v;
      }
    }
  }
  // ---- partial of gq rows i = 8 grp .. 8 grp + 7 over this block's columns
  {
    float4 acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = 0; j < MR_CB; ++j) {
      const float4 xv = mr_ld4(Xs + j * MR_LD + 4 * f4);
      const float4 g0 = mr_ld4(GT + j * MR_LG + 8 * grp), g1 = mr_ld4(GT + j * MR_LG + 8 * grp + 4);
      const float gv[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        acc[e].x = __builtin_fmaf(gv[e], xv.x, acc[e].x);
        acc[e].y = __builtin_fmaf(gv[e], xv.y, acc[e].y);
        acc[e].z = __builtin_fmaf(gv[e], xv.z, acc[e].z);
        acc[e].w = __builtin_fmaf(gv[e], xv.w, acc[e].w);
      }
    }
    float* pb = gq_part + ((size_t)b * nblk + blk) * MR_R * MR_D;
#pragma unroll
    for (int e = 0; e < 8; ++e) *reinterpret_cast<float4*>(pb + (8 * grp + e) * MR_D + 4 * f4) = acc[e];
  }
}

// gq (B,R,D) = sum over the column blocks, in block order; one float4 per thread
__global__ __launch_bounds__(256) void pn_ms_rows_reduce_kernel(const float* __restrict__ gq_part, int R, int nblk,
                                                                float* __restrict__ gq) {
  const int b = blockIdx.y;
  const int e = blockIdx.x * 256 + threadIdx.x;      // float4 index inside (R, D)
  if (e >= R * (MR_D / 4)) return;
  const float* p = gq_part + (size_t)b * nblk * MR_R * MR_D + 4 * e;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  // eight partials in flight, added in block order (one load -> add per trip waited a memory latency per
  // block: 40 us for the 157 blocks of 10 000 points)
  int k = 0;
  for (; k + 8 <= nblk; k += 8) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = mr_ld4(p + (size_t)(k + u) * MR_R * MR_D);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      s.x += v[u].x;
      s.y += v[u].y;
      s.z += v[u].z;
      s.w += v[u].w;
    }
  }
  for (; k < nblk; ++k) {
    const float4 v = mr_ld4(p + (size_t)k * MR_R * MR_D);
    s.x += v.x;
    s.y += v.y;
    s.z += v.z;
    s.w += v.w;
  }
  *reinterpret_cast<float4*>(gq + (size_t)b * R * MR_D + 4 * e) = s;
}

// gx[b, rows[b, r], :] += g[b, r, :] for r = 0 .. R-1 IN ORDER (rows may repeat among the padded
// entries): one workgroup per batch item, 32 threads per row of 128 floats
__global__ __launch_bounds__(256) void pn_ms_rows_scatter_kernel(const float* __restrict__ g, const int64_t* __restrict__ rows,
                                                                 int N, int R, float* __restrict__ gx) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const int f4 = tid & 31, sub = tid >> 5;
  // eight rows in flight only when they are distinct; the simple safe form: one row at a time per
  // 32-thread group would race on repeats, so the groups take turns — r ascending, one barrier each
  for (int r0 = 0; r0 < R; r0 += 8) {
    for (int turn = 0; turn < 8; ++turn) {
      const int r = r0 + turn;
      if (sub == turn && r < R) {
        const int64_t row = rows[(size_t)b * R + r];
        if (row >= 0 && row < N) {
          float4* o = reinterpret_cast<float4*>(gx + ((size_t)b * N + row) * MR_D + 4 * f4);
          const float4 v = mr_ld4(g + ((size_t)b * R + r) * MR_D + 4 * f4);
          float4 w = *o;
          w.x += v.x;
          w.y += v.y;
          w.z += v.z;
          w.w += v.w;
          *o = w;
        }
      }
      __syncthreads();
    }
  }
}

static size_t mr_part_bytes(int B, int N) { return pn_align_up((size_t)B * pn_cdiv(N, MR_CB) * MR_R * MR_D * sizeof(float), 256); }
static size_t mr_gu_bytes(int B) { return pn_align_up((size_t)B * MR_R * MR_D * sizeof(float), 256); }
extern "C" size_t pn_meanshift_rows_bwd_workspace(int B, int N) {
  return mr_part_bytes(B, N) + mr_gu_bytes(B) + pn_align_up((size_t)B * MR_R * 4 * sizeof(float), 256);
}

// One step of the row-restricted backward.  gy, y, q (B,R,D), rsum, unorm (B,R): the R rows of the
// incoming gradient, of the step's result, of its input iterate and of its saved row sums / norms;
// x (B,N,D) the data; bsq (B).  Writes gq (B,R,D) and ADDS the step's contribution into gx (B,N,D).
extern "C" int pn_meanshift_rows_bwd_f32(const float* gy, const float* y, const float* q, const float* rsum,
                                         const float* unorm, const float* x, const float* bsq, int B, int N, int D,
                                         int R, float* gq, float* gx, void* workspace, size_t workspace_bytes,
                                         hipStream_t stream) {
  PN_CHECK_ARG(D == MR_D, "pn_meanshift_rows_bwd_f32: D must be %d, got %d", MR_D, D);
  PN_CHECK_ARG(R >= 1 && R <= MR_R, "pn_meanshift_rows_bwd_f32: 1 <= R <= %d, got %d", MR_R, R);
  PN_CHECK_ARG(B >= 1 && N >= 1, "pn_meanshift_rows_bwd_f32: empty input");
  const int nblk = pn_cdiv(N, MR_CB);
  if (workspace_bytes < pn_meanshift_rows_bwd_workspace(B, N)) {
    pn_set_error("pn_meanshift_rows_bwd_f32: workspace too small");
    return PN_ERR_WORKSPACE;
  }
  const size_t lds_valu = (size_t)(3 * MR_R * MR_LD + 3 * MR_R * MR_LG + 3 * MR_R) * sizeof(float);
  static unsigned attr_devs = 0;
  if (pn_first_on_device(&attr_devs)) {
    PN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(pn_ms_rows_bwd_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_valu));
  }
  float* part = static_cast<float*>(workspace);
  float* gu = reinterpret_cast<float*>(static_cast<char*>(workspace) + mr_part_bytes(B, N));
  float* scal = reinterpret_cast<float*>(static_cast<char*>(workspace) + mr_part_bytes(B, N) + mr_gu_bytes(B));
  {
    PN_PROF("meanshift_rows_bwd", stream);
    hipLaunchKernelGGL(pn_ms_rows_prep_kernel, dim3(MR_R / 4, B), dim3(256), 0, stream, gy, y, rsum, unorm, bsq, R, gu, scal);
    hipLaunchKernelGGL(pn_ms_rows_bwd_kernel, dim3(nblk, B), dim3(256), lds_valu, stream, (const float*)gu,
                       (const float*)scal, q, x, bsq, N, R, nblk, gx, part);
  }
  PN_CHECK_LAUNCH();
  hipLaunchKernelGGL(pn_ms_rows_reduce_kernel, dim3(pn_cdiv(R * (MR_D / 4), 256), B), dim3(256), 0, stream, part, R, nblk,
                     gq);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

extern "C" int pn_meanshift_rows_scatter_add_f32(const float* g, const int64_t* rows, int B, int N, int D, int R,
                                                 float* gx, hipStream_t stream) {
  PN_CHECK_ARG(D == MR_D, "pn_meanshift_rows_scatter_add_f32: D must be %d, got %d", MR_D, D);
  PN_CHECK_ARG(B >= 1 && R >= 1, "pn_meanshift_rows_scatter_add_f32: empty input");
  hipLaunchKernelGGL(pn_ms_rows_scatter_kernel, dim3(B), dim3(256), 0, stream, g, rows, N, R, gx);
  PN_CHECK_LAUNCH();
  return PN_OK;
}
