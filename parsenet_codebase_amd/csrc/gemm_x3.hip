// fp32-grade GEMM of the per-point layers (nn.Conv1d(kernel_size = 1): src/model.py:56-180,
// src/PointNet.py:143-289) on the bf16 matrix cores: y[b] = W x[b] (+ bias) with W (M,K) row-major and
// x (B,K,N) channel-first, and the same product with W transposed (the gradient w.r.t. x).
//
// gfx950 runs v_mfma_f32_32x32x2_f32 at 1/16 of the bf16 rate.  Both operands are split error-free into
// three bf16 pieces (split_common.h: x = xh + xm + xl exactly) and a product is evaluated as
//   xh yh + xh ym + xm yh + xm ym + xh yl + xl yh      (fp32 accumulation on v_mfma_f32_32x32x16_bf16);
// the three dropped terms are below 2^-25 |x y|: an fp32 dot product with another summation order, six
// bf16 MFMAs (32 cycles each) instead of eight fp32 ones (64 cycles each) per 32 x 32 x 16 block.
//
// Operands reach the kernel as IMAGES, written once per call by the two split kernels below: for every
// tile of 32 rows (rows of W, or 32 consecutive points of one batch item) and every block of 16
// contraction indices one 3 KiB unit  [piece 3][chunk 2][row 32] x 16 bytes  (chunk = 8 consecutive
// contraction indices of a row as packed bf16).  Chunk-major: the 32 lanes of a half wave read 32
// consecutive 16-byte units with ds_read_b128 — no bank conflicts, no swizzle — and a unit is a linear
// 3 KiB copy for the LDS DMA.
//
// pn_gemm_x3_kernel: 256 threads = 2 x 2 waves, a wave owns 64 x 64 of the 128 x 128 block (four 32 x 32
// accumulators).  A stage = one block of 16 contraction indices of the 4 + 4 tiles (24 KiB), double
// buffered (48 KiB: three workgroups per CU); the DMA of stage s + 1 runs under the 24 MFMAs per wave of
// stage s; one barrier per stage.
#include "split_common.h"

typedef float gx_f32x16 __attribute__((ext_vector_type(16)));

#define GX_UNIT 192             // u32x4 per unit: 3 pieces x 2 chunks x 32 rows
#define GX_PIECE 64             // u32x4 per piece of a unit

// rows image of a row-major matrix src (R, K): unit (row tile, k block); rows >= R and k >= K are zero
// (blockIdx.y: one matrix of a batch of them — the weight gradient images gy[b] (M, N) and x[b] (K, N) with the
// POINTS as contraction index; image b follows image b - 1)
__global__ __launch_bounds__(256) void pn_gx_img_rows_kernel(const float* __restrict__ src, int R, int K, int nkb,
                                                             u32x4* __restrict__ img) {
  const int unit = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int rt = unit / nkb, kb = unit - rt * nkb;
  const int lane = threadIdx.x & 63, r = lane & 31, c = lane >> 5;
  if (rt * 32 >= R) return;
  src += (size_t)blockIdx.y * R * K;
  img += (size_t)blockIdx.y * ((R + 31) / 32) * nkb * GX_UNIT;
  const int row = rt * 32 + r, k0 = kb * 16 + 8 * c;
  float v[8];
  if (row < R && k0 + 8 <= K && (K & 3) == 0) {
    // (a lane's 8 contraction indices are contiguous in memory: two 16-byte loads)
    const float4 a = *reinterpret_cast<const float4*>(src + (size_t)row * K + k0);
    const float4 b = *reinterpret_cast<const float4*>(src + (size_t)row * K + k0 + 4);
    v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = b.x, v[5] = b.y, v[6] = b.z, v[7] = b.w;
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (row < R && k0 + e < K) ? src[(size_t)row * K + k0 + e] : 0.f;
  }
  u32x4 vh, vm, vl;
  X3_SPLIT_TO(v[0], v[1], vh, vm, vl, 0);
  X3_SPLIT_TO(v[2], v[3], vh, vm, vl, 1);
  X3_SPLIT_TO(v[4], v[5], vh, vm, vl, 2);
  X3_SPLIT_TO(v[6], v[7], vh, vm, vl, 3);
  u32x4* dst = img + (size_t)unit * GX_UNIT + c * 32 + r;
  dst[0] = vh;
  dst[GX_PIECE] = vm;
  dst[2 * GX_PIECE] = vl;
}

// image of a channel-first tensor src (B, C, N) with the POINTS as rows and the channels as the
// contraction index: unit ((b, point tile), k block); points >= N and channels >= C are zero.
// (A row-major matrix W (M, K) read this way — B = 1, C = M, N = K — gives the rows image of W^T.)
// The channels may come from up to four tensors (B, C_s, N) laid end to end — the concatenation of the edge-conv
// layers' outputs that src/model.py:150 builds with torch.cat is never written out: source s holds the channels
// [cbeg[s], cbeg[s + 1]), every boundary a multiple of 8 (one lane's chunk comes from one source).
struct GxSources {
  const float* ptr[4];
  int cbeg[4];      // first channel of source s (unused sources: INT_MAX)
  int ccnt[4];      // its channel count (the batch stride is ccnt * N)
};
__global__ __launch_bounds__(256) void pn_gx_img_cf_kernel(GxSources S, int C, int N, int ntile, int nkb,
                                                           u32x4* __restrict__ img) {
  const int b = blockIdx.y;
  const int unit = blockIdx.x * 4 + (threadIdx.x >> 6);      // inside the batch item
  const int pt = unit / nkb, kb = unit - pt * nkb;
  const int lane = threadIdx.x & 63, r = lane & 31, c = lane >> 5;
  if (pt >= ntile) return;
  const int n = pt * 32 + r, k0 = kb * 16 + 8 * c;
  int si = 0;
#pragma unroll
  for (int t = 1; t < 4; ++t) si += (k0 >= S.cbeg[t]) ? 1 : 0;
  const float* __restrict__ sb = S.ptr[si] + (size_t)b * S.ccnt[si] * N - (size_t)S.cbeg[si] * N;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (n < N && k0 + e < C) ? sb[(size_t)(k0 + e) * N + n] : 0.f;
  u32x4 vh, vm, vl;
  X3_SPLIT_TO(v[0], v[1], vh, vm, vl, 0);
  X3_SPLIT_TO(v[2], v[3], vh, vm, vl, 1);
  X3_SPLIT_TO(v[4], v[5], vh, vm, vl, 2);
  X3_SPLIT_TO(v[6], v[7], vh, vm, vl, 3);
  u32x4* dst = img + ((size_t)b * ntile * nkb + unit) * GX_UNIT + c * 32 + r;
  dst[0] = vh;
  dst[GX_PIECE] = vm;
  dst[2 * GX_PIECE] = vl;
}

// out[b][m][n] = sum_k A[m][k] X[b][n][k] (+ bias[m]); imgA: rows image with mt tiles, imgX: point image with
// B * nt tiles (tile t of batch item b at b * nt + t), nkb blocks of 16 contraction indices in both.
// grid (ceil(mt / 4), ceil(B nt / 4)): consecutive workgroups share the activation tiles (L2).
// blockIdx.z = zb * zs + s (the weight gradient; the forward product has ONE z): operand pair zb — imgA and imgX
// advance by a_z / x_z units per pair — and slice s of the contraction, k blocks [s kb_per, (s + 1) kb_per); the
// block's result goes to out + z * out_z (partial sums, added up in z order by pn_gx_reduce_kernel).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void pn_gemm_x3_kernel(
    const u32x4* __restrict__ imgA, const u32x4* __restrict__ imgX, int M, int N, int mt, int nt, int ntot, int nkb,
    const float* __restrict__ bias, float* __restrict__ out, int zs, int kb_per, size_t a_z, size_t x_z,
    size_t out_z) {
  __shared__ __attribute__((aligned(16))) u32x4 lds[2][8 * GX_UNIT];      // per stage: 4 A units, 4 X units
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  const int at0 = blockIdx.x * 4, xt0 = blockIdx.y * 4;
  const int zb = blockIdx.z / zs, sl = blockIdx.z - zb * zs;
  const int kb0 = sl * kb_per, kb1 = min(nkb, kb0 + kb_per);
  imgA += zb * a_z * GX_UNIT;
  imgX += zb * x_z * GX_UNIT;
  out += blockIdx.z * out_z;
  // DMA: a stage is 24 chunks of 1 KiB (8 units x 3); wave w moves chunks 6 w .. 6 w + 5 = units 2 w, 2 w + 1
  const u32x4* src[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int unit = 2 * wave + u;                 // 0..3: A tiles, 4..7: X tiles
    if (unit < 4) {
      const int t = min(at0 + unit, mt - 1);
      src[u] = imgA + (size_t)t * nkb * GX_UNIT;
    } else {
      const int t = min(xt0 + unit - 4, ntot - 1);
      src[u] = imgX + (size_t)t * nkb * GX_UNIT;
    }
  }
#define GX_STAGE(KB, BUF)                                                                     \
  {                                                                                           \
    _Pragma("unroll") for (int u = 0; u < 2; ++u) _Pragma("unroll") for (int c = 0; c < 3; ++c) \
        X3_GLDS16(src[u] + (size_t)(KB) * GX_UNIT + c * 64 + lane, &lds[BUF][(2 * wave + u) * GX_UNIT + c * 64]); \
  }
  gx_f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
  if (kb0 < kb1) GX_STAGE(kb0, 0);
  int cur = 0;
  for (int kb = kb0; kb < kb1; ++kb) {
    __builtin_amdgcn_s_waitcnt(0x0f70);     // vmcnt(0): this wave's share of stage kb has landed
    __syncthreads();                        // ... everybody's; and everybody is done with the other buffer
    if (kb + 1 < kb1) GX_STAGE(kb + 1, cur ^ 1);
    const u32x4* __restrict__ L = lds[cur];
    bf16x8 ah[2], am[2], al[2], bh[2], bm[2], bl[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const u32x4* pa = L + (2 * wm + i) * GX_UNIT + h * 32 + r;
      ah[i] = x3_as_bf16(pa[0]);
      am[i] = x3_as_bf16(pa[GX_PIECE]);
      al[i] = x3_as_bf16(pa[2 * GX_PIECE]);
      const u32x4* pb = L + (4 + 2 * wn + i) * GX_UNIT + h * 32 + r;
      bh[i] = x3_as_bf16(pb[0]);
      bm[i] = x3_as_bf16(pb[GX_PIECE]);
      bl[i] = x3_as_bf16(pb[2 * GX_PIECE]);
    }
    // (products outermost: consecutive MFMAs go to different accumulators; small terms first)
#define GX_P(A_, B_)                                    \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_[i], B_[j], acc[i][j], 0, 0, 0)
    GX_P(al, bh);
    GX_P(ah, bl);
    GX_P(am, bm);
    GX_P(am, bh);
    GX_P(ah, bm);
    GX_P(ah, bh);
#undef GX_P
    cur ^= 1;
  }
#undef GX_STAGE
  // D[row = (v & 3) + 8 (v >> 2) + 4 h][col = r]: row = output channel, col = point
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int t = xt0 + 2 * wn + j;
    if (t >= ntot) continue;
    const int b = t / nt, p = t - b * nt;
    const int n = p * 32 + r;
    if (n >= N) continue;
    float* __restrict__ ob = out + (size_t)b * M * N + n;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m0 = (at0 + 2 * wm + i) * 32;
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int m = m0 + (v & 3) + 8 * (v >> 2) + 4 * h;
        if (m < M) ob[(size_t)m * N] = acc[i][j][v] + (bias ? bias[m] : 0.f);
      }
    }
  }
}

static inline int gx_nkb(int K) { return (K + 15) / 16; }
static inline GxSources gx_one_source(const float* p, int C) {
  GxSources S;
  for (int t = 0; t < 4; ++t) {
    S.ptr[t] = p;
    S.cbeg[t] = t == 0 ? 0 : 0x7fffffff;
    S.ccnt[t] = C;
  }
  return S;
}

extern "C" size_t pn_gemm_x3_weight_image_bytes(int M, int K) {
  return (size_t)pn_cdiv(M, 32) * gx_nkb(K) * GX_UNIT * 16;
}
extern "C" size_t pn_gemm_x3_points_image_bytes(int B, int C, int N) {
  return (size_t)B * pn_cdiv(N, 32) * gx_nkb(C) * GX_UNIT * 16;
}

// image of W (M, K) for  y = W x  (transposed = 0), or of W^T for  gx = W^T gy  (transposed = 1: the
// rows of the image are then the K columns of W, the contraction runs over M)
extern "C" int pn_gemm_x3_weight_image_f32(const float* w, int M, int K, int transposed, void* img, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PN_CHECK_ARG(M >= 1 && K >= 1, "pn_gemm_x3_weight_image_f32: empty matrix");
  if (!transposed) {
    const int units = pn_cdiv(M, 32) * gx_nkb(K);
    hipLaunchKernelGGL(pn_gx_img_rows_kernel, dim3(pn_cdiv(units, 4)), dim3(256), 0, stream, w, M, K, gx_nkb(K),
                       (u32x4*)img);
  } else {
    const int ntile = pn_cdiv(K, 32), nkb = gx_nkb(M);
    hipLaunchKernelGGL(pn_gx_img_cf_kernel, dim3(pn_cdiv(ntile * nkb, 4), 1), dim3(256), 0, stream, gx_one_source(w, M), M,
                       K, ntile, nkb, (u32x4*)img);
  }
  PN_CHECK_LAUNCH();
  return PN_OK;
}

// out (B, M, N) = A x (+ bias (M), may be NULL).  img_a: pn_gemm_x3_weight_image_f32 of an (M, K) operand
// (i.e. of W, or of W^T with M and K exchanged); x (B, K, N) channel-first fp32; workspace:
// pn_gemm_x3_points_image_bytes(B, K, N) bytes for the image of x.
static int gemm_x3_sources(const void* img_a, GxSources S, const float* bias, int B, int M, int K, int N, float* out,
                           void* workspace, size_t workspace_bytes, hipStream_t stream);

extern "C" int pn_gemm_x3_f32(const void* img_a, const float* x, const float* bias, int B, int M, int K, int N, float* out,
                              void* workspace, size_t workspace_bytes, void* stream_) {
  PN_CHECK_ARG(x, "pn_gemm_x3_f32: null input");
  return gemm_x3_sources(img_a, gx_one_source(x, K), bias, B, M, K, N, out, workspace, workspace_bytes,
                         (hipStream_t)stream_);
}

// the same product with the K input channels spread over nsrc <= 4 tensors xs[s] (B, cs[s], N), every cs[s] a
// multiple of 8, sum cs = K: W applied to the concatenation without writing it
extern "C" int pn_gemm_x3_cat_f32(const void* img_a, const float* const* xs, const int* cs, int nsrc, const float* bias,
                                  int B, int M, int N, float* out, void* workspace, size_t workspace_bytes,
                                  void* stream_) {
  PN_CHECK_ARG(xs && cs && nsrc >= 1 && nsrc <= 4, "pn_gemm_x3_cat_f32: 1 to 4 sources");
  GxSources S = gx_one_source(xs[0], cs[0]);
  int K = 0;
  for (int t = 0; t < nsrc; ++t) {
    PN_CHECK_ARG(xs[t] && cs[t] >= 8 && cs[t] % 8 == 0, "pn_gemm_x3_cat_f32: source %d has %d channels (multiples of 8)", t,
                 cs[t]);
    S.ptr[t] = xs[t];
    S.cbeg[t] = K;
    S.ccnt[t] = cs[t];
    K += cs[t];
  }
  return gemm_x3_sources(img_a, S, bias, B, M, K, N, out, workspace, workspace_bytes, (hipStream_t)stream_);
}

static int gemm_x3_sources(const void* img_a, GxSources S, const float* bias, int B, int M, int K, int N, float* out,
                           void* workspace, size_t workspace_bytes, hipStream_t stream) {
  PN_CHECK_ARG(B >= 1 && M >= 1 && K >= 1 && N >= 1, "pn_gemm_x3_f32: empty operand");
  if (workspace_bytes < pn_gemm_x3_points_image_bytes(B, K, N)) {
    pn_set_error("pn_gemm_x3_f32: workspace too small");
    return PN_ERR_WORKSPACE;
  }
  const int nt = pn_cdiv(N, 32), nkb = gx_nkb(K), mt = pn_cdiv(M, 32);
  PN_CHECK_ARG((long long)B * nt < (1ll << 30), "pn_gemm_x3_f32: too many point tiles");
  {
    PN_PROF("gemm_x3_image", stream);
    hipLaunchKernelGGL(pn_gx_img_cf_kernel, dim3(pn_cdiv(nt * nkb, 4), B), dim3(256), 0, stream, S, K, N, nt, nkb,
                       (u32x4*)workspace);
  }
  PN_CHECK_LAUNCH();
  {
    PN_PROF("gemm_x3", stream);
    hipLaunchKernelGGL(pn_gemm_x3_kernel, dim3(pn_cdiv(mt, 4), pn_cdiv(B * nt, 4)), dim3(256), 0, stream,
                       (const u32x4*)img_a, (const u32x4*)workspace, M, N, mt, nt, B * nt, nkb, bias, out, 1, nkb,
                       (size_t)0, (size_t)0, (size_t)0);
  }
  PN_CHECK_LAUNCH();
  return PN_OK;
}

// ---- the weight gradient: gw (M, K) = sum_b gy[b] (M, N) x[b]^T (N, K), contraction over the POINTS -----------
// (src/model.py:157-176, src/PointNet.py:196-284: what autograd's conv1d backward computes with a rocBLAS product
// over B N = 40 000 points.)  Rows images of gy[b] and x[b] (the points are contiguous in both: 16-byte loads),
// the kernel above on (M / 128) x (K / 128) output blocks x B operand pairs x S slices of the points — S so that
// the launch has ~3 workgroups per CU — and a FIXED-ORDER sum of the B S partial results: bit-reproducible, no
// atomics.  gb (M) or NULL: the bias gradient sum_b sum_n gy[b][m][n] from the same pass (fixed order as well).
static inline int gx_wgrad_slices(int B, int M, int K, int nkb) {
  const int blocks = pn_cdiv(pn_cdiv(M, 32), 4) * pn_cdiv(pn_cdiv(K, 32), 4) * B;
  int S = pn_cdiv(768, blocks);
  const int smax = nkb / 16 > 0 ? nkb / 16 : 1;       // at least 16 k blocks (256 points) per slice
  if (S > smax) S = smax;
  if (S < 1) S = 1;
  return S;
}

// out[i] = sum_z part[z][i] in z order; grid-stride, 4 elements per thread
__global__ __launch_bounds__(256) void pn_gx_reduce_kernel(const float* __restrict__ part, int Z, size_t n,
                                                           float* __restrict__ out) {
  for (size_t i = (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * blockDim.x * 4) {
    if (i + 4 <= n) {
      float4 acc = *reinterpret_cast<const float4*>(part + i);
      for (int z = 1; z < Z; ++z) {
        const float4 v = *reinterpret_cast<const float4*>(part + (size_t)z * n + i);
        acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
      }
      *reinterpret_cast<float4*>(out + i) = acc;
    } else {
      for (size_t j = i; j < n; ++j) {
        float acc = part[j];
        for (int z = 1; z < Z; ++z) acc += part[(size_t)z * n + j];
        out[j] = acc;
      }
    }
  }
}

// gb[m] = sum_b sum_n gy[b][m][n]: one 256-thread workgroup per output channel; a thread adds the elements
// 4 t .. 4 t + 3 (mod 1024) of every batch item's row in order (four independent 16-byte loads in flight), the 256
// partial sums meet in a fixed LDS tree: the same bits on every run.
// (Round 6, first version: one WAVE per channel striding the row 64 floats at a time — 625 dependent trips of a
// memory latency each, 0.3 ms per layer whatever its size; it made the whole weight gradient slower than rocBLAS.)
__global__ __launch_bounds__(256) void pn_gx_bias_grad_kernel(const float* __restrict__ gy, int B, int M, int N,
                                                              float* __restrict__ gb) {
  __shared__ float red[256];
  const int m = blockIdx.x, t = threadIdx.x;
  float acc = 0.f;
  const bool vec = (N & 3) == 0;
  for (int b = 0; b < B; ++b) {
    const float* row = gy + ((size_t)b * M + m) * N;
    if (vec) {
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      int n = 4 * t;
      for (; n + 3 * 1024 < N; n += 4 * 1024) {
        const float4 v0 = *reinterpret_cast<const float4*>(row + n);
        const float4 v1 = *reinterpret_cast<const float4*>(row + n + 1024);
        const float4 v2 = *reinterpret_cast<const float4*>(row + n + 2048);
        const float4 v3 = *reinterpret_cast<const float4*>(row + n + 3072);
        a0 += (v0.x + v0.y) + (v0.z + v0.w);
        a1 += (v1.x + v1.y) + (v1.z + v1.w);
        a2 += (v2.x + v2.y) + (v2.z + v2.w);
        a3 += (v3.x + v3.y) + (v3.z + v3.w);
      }
      for (; n < N; n += 1024) {
        const float4 v0 = *reinterpret_cast<const float4*>(row + n);
        a0 += (v0.x + v0.y) + (v0.z + v0.w);
      }
      acc += (a0 + a1) + (a2 + a3);
    } else {
      for (int n = t; n < N; n += 256) acc += row[n];
    }
  }
  red[t] = acc;
  __syncthreads();
#pragma unroll
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) red[t] += red[t + o];
    __syncthreads();
  }
  if (t == 0) gb[m] = red[0];
}

extern "C" size_t pn_gemm_x3_wgrad_workspace(int B, int M, int K, int N) {
  const int nkb = gx_nkb(N);
  const size_t img = ((size_t)pn_cdiv(M, 32) + pn_cdiv(K, 32)) * B * nkb * GX_UNIT * 16;
  const size_t part = (size_t)B * gx_wgrad_slices(B, M, K, nkb) * M * K * sizeof(float);
  return img + part;
}

extern "C" int pn_gemm_x3_wgrad_f32(const float* gy, const float* x, int B, int M, int K, int N, float* gw, float* gb,
                                    void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PN_CHECK_ARG(gy && x && gw, "pn_gemm_x3_wgrad_f32: null pointer");
  PN_CHECK_ARG(B >= 1 && M >= 1 && K >= 1 && N >= 1, "pn_gemm_x3_wgrad_f32: empty operand");
  if (workspace_bytes < pn_gemm_x3_wgrad_workspace(B, M, K, N)) {
    pn_set_error("pn_gemm_x3_wgrad_f32: workspace too small");
    return PN_ERR_WORKSPACE;
  }
  const int nkb = gx_nkb(N), mt = pn_cdiv(M, 32), kt = pn_cdiv(K, 32);
  const int S = gx_wgrad_slices(B, M, K, nkb), kb_per = pn_cdiv(nkb, S);
  u32x4* imgG = (u32x4*)workspace;
  u32x4* imgX = imgG + (size_t)B * mt * nkb * GX_UNIT;
  float* part = (float*)(imgX + (size_t)B * kt * nkb * GX_UNIT);
  {
    PN_PROF("gemm_x3_image", stream);
    hipLaunchKernelGGL(pn_gx_img_rows_kernel, dim3(pn_cdiv(mt * nkb, 4), B), dim3(256), 0, stream, gy, M, N, nkb, imgG);
    hipLaunchKernelGGL(pn_gx_img_rows_kernel, dim3(pn_cdiv(kt * nkb, 4), B), dim3(256), 0, stream, x, K, N, nkb, imgX);
  }
  PN_CHECK_LAUNCH();
  {
    PN_PROF("gemm_x3_wgrad", stream);
    // "points" of the kernel = the K rows of x[b]: out[z][m][k]
    hipLaunchKernelGGL(pn_gemm_x3_kernel, dim3(pn_cdiv(mt, 4), pn_cdiv(kt, 4), B * S), dim3(256), 0, stream,
                       (const u32x4*)imgG, (const u32x4*)imgX, M, K, mt, kt, kt, nkb, (const float*)nullptr, part, S,
                       kb_per, (size_t)mt * nkb, (size_t)kt * nkb, (size_t)M * K);
    const size_t n = (size_t)M * K;
    hipLaunchKernelGGL(pn_gx_reduce_kernel, dim3((unsigned)pn_cdiv((long long)pn_cdiv((long long)n, 4), 256)), dim3(256),
                       0, stream, (const float*)part, B * S, n, gw);
    if (gb) hipLaunchKernelGGL(pn_gx_bias_grad_kernel, dim3(M), dim3(256), 0, stream, gy, B, M, N, gb);
  }
  PN_CHECK_LAUNCH();
  return PN_OK;
}
