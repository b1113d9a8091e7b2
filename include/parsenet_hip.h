/* parsenet_hip.h — C ABI of the MI355X (gfx950) ParSeNet hot-path library.
 *
 * The reference (Hippogriff/parsenet-codebase) is pure Python on PyTorch: it has no FFI
 * of its own, so every entry point below replaces a *composition of stock torch ops*
 * inside one reference function (cited as file:line, paths relative to the upstream
 * repository).  The binding a maintainer adds is a ctypes stub — see INTEGRATION.md.
 *
 * Conventions
 *   - plain pointers + sizes only; all pointers are DEVICE pointers unless named h_*;
 *   - every tensor is dense row-major ("contiguous") in the stated shape;
 *   - floating data is fp32, index data is int64 (torch defaults), unless stated;
 *   - `stream` is a hipStream_t passed as void*; launches are asynchronous on it;
 *   - buffers are caller-owned; scratch space is passed in, its size queried with the
 *     matching pn_*_workspace();
 *   - return value 0 = success, negative = error (PN_ERR_*); pn_last_error() returns a
 *     thread-local description.  Functions are re-entrant and keep no global state.
 */
#ifndef PARSENET_HIP_H
#define PARSENET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PN_OK 0
#define PN_ERR_ARG (-1)
#define PN_ERR_HIP (-2)
#define PN_ERR_WORKSPACE (-3)
#define PN_ERR_UNSUPPORTED (-4)

const char* pn_last_error(void);
int pn_abi_version(void);

/* ---- per-kernel timing (HIP events on the launch stream; off by default) -----------
 * enable(1) brackets every kernel family launched through this library with an event pair;
 * count() resolves outstanding events and returns the number of families seen; get(i, ...)
 * returns name, accumulated milliseconds and number of launches; reset() clears. */
void pn_prof_enable(int on);
void pn_prof_reset(void);
int pn_prof_count(void);
int pn_prof_get(int i, char* name, int name_len, double* total_ms, long long* calls);

/* ---- kNN graph ------------------------------------------------------------------
 * pn_knn_f32 replaces src/model.py:9-22 knn(x,k) and src/PointNet.py:9-26 knn(x,k1,k2)
 * (k1 == k2; a dilation k2 > k1 is a strided slice of the k2 result done by the caller);
 * pn_knn_pn_f32 replaces src/PointNet.py:29-69 knn_points_normals (rows 0:3 xyz, 3:6 unit
 * normals, metric |dp|^2 * (1 + (2 - 2 ni.nj))).
 * x (B,C,N) channel-first.  idx (B,N,k): for every point the k candidates with the LARGEST
 * negated distance, best first, the point itself included; equal values -> smaller index.
 * Arithmetic: dot products / squared norms are channel-ordered fp32 fma chains, combined as
 * (-xx[j] - (-2*dot)) - xx[i] like the reference, every step rounded to fp32.
 * Limits: 1 <= k <= 128, k <= N. */
size_t pn_knn_workspace(int B, int C, int N, int k);
int pn_knn_f32(const float* x, int B, int C, int N, int k, int64_t* idx, void* workspace,
               size_t workspace_bytes, void* stream);
int pn_knn_pn_f32(const float* x6, int B, int N, int k, int64_t* idx, void* workspace,
                  size_t workspace_bytes, void* stream);
/* The same graph with int32 indices — the form the library's own edge-conv kernels consume
 * (pn_edgeconv_reduce_fwd_i32, pn_edgeconv_bwd_i32: half the index bytes of the torch dtype the
 * reference's topk returns, src/model.py:19).  metric 0: feature space (pn_knn_f32), 1: points +
 * normals (pn_knn_pn_f32, C = 6).  k <= 16 (the SplineNets' graphs, src/model.py:9-22 with k = 10)
 * takes ONE distance pass with the k best candidates of a lane in registers (csrc/knn_smallk.h);
 * same workspace query. */
int pn_knn_graph_i32(const float* x, int B, int C, int N, int k, int metric, int32_t* idx,
                     void* workspace, size_t workspace_bytes, void* stream);

/* ---- neighbours of 3-D points from coordinate differences (evaluation-mode fitting) ----------
 * Replaces the broadcast difference + topk of src/fitting_utils.py:150-164, 202-237
 * (up_sample_points_torch(_in_range): k = 5) and the KD-tree search behind open3d's
 * remove_statistical_outlier (src/fitting_utils.py:704-710: k = 20, float64) for a RAGGED batch of
 * segments: pts (total,3) fp32, off (S+1) int32 row offsets, max_n >= the largest segment.
 * d = ((dx^2 + dy^2) + dz^2), each operation rounded once (f64 = 1: in float64 on the fp32 points).
 * idx (total,k) int32 indices LOCAL to the segment, nearest first, the point itself first, equal
 * distances -> smaller index; a segment with fewer than k points is padded with the point itself.
 * dist (total,k) float / double (per f64) or NULL: the Euclidean distances of those neighbours.
 * Limits: k <= 64; segments up to 10 240 points (float64: 5 120). */
int pn_knn3_ragged(const float* pts, const int* off, int S, int max_n, int k, int f64, int32_t* idx,
                   void* dist, void* stream);

/* ---- dot-product selection between two point sets ------------------------------------
 * Replaces src/mean_shift.py:125-137 (compute_bandwidth: 2 - 2 X X^T, topk(K, largest=False),
 * the K-th smallest distance per row) and :146-149 (nms: argmin over centres of 2 - 2 C X^T).
 * 2 - 2*dot is an exact decreasing map of dot for unit vectors, so both are selections on the
 * dot products.  q (B,Nq,C), c (B,Nc,C) point-major.  Exactly one of out_idx (B,Nq,k) int64
 * (indices of the k largest dots, best first, ties -> smaller index, k <= 128) and out_val
 * (B,Nq) fp32 (the k-th largest dot, k <= 512) is non-NULL.  flags (B,Nq) int32 is set to 1
 * for queries whose result is invalid (survivor overflow on massively tied data): the caller
 * recomputes those rows.  PN_ERR_UNSUPPORTED if the shape is outside the fast path
 * (needs Nc/16 >= 2k and C <= 256); pn_dot_select_workspace then returns 0. */
size_t pn_dot_select_workspace(int B, int C, int Nq, int Nc, int k, int want_value);
int pn_dot_select_f32(const float* q, int Nq, const float* c, int Nc, int B, int C, int k,
                      int64_t* out_idx, float* out_val, int* flags, void* workspace,
                      size_t workspace_bytes, void* stream);

/* ---- mean-shift iterations ----------------------------------------------------------
 * Replaces src/mean_shift.py:45-79 (mean_shift_, gaussian kernel) and the autograd graph the
 * reference keeps through it.  All tensors point-major (B,N,D) with D = 128; bsq (B) = b^2.
 *   slices   : upper bound S of the column slices the launches use for (B,N) (sizes the scratch).
 *   pack     : x (B,N,D) -> xt (B,D,Np), Np = N rounded up to 64, zero padded.
 *   iter_fwd : y = normalise(q + ((K X) / rowsum(K) - q)), K = exp(clamp(-(2 - 2 q x^T)/b^2/2)).
 *              Scratch opart (B,S,N,D), rpart (B,S,N).  Saves rsum, unorm (B,N).
 *   iter_bwd : given gy = dL/dy: gq = dL/dq (overwritten), gx += this iteration's contribution
 *              to dL/dx (recomputes K; nothing of size N x N is stored).
 *              Scratch gu, go (B,N,D), cs (B,2,N), qt, gut (B,D,Np), opart_q, opart_x (B,S,N,D). */
int pn_meanshift_slices(int B, int N);
int pn_meanshift_pack_f32(const float* x, int B, int N, int D, float* xt, void* stream);
int pn_meanshift_iter_fwd_f32(const float* q, const float* x, const float* xt, const float* bsq,
                              int B, int N, int D, float* opart, float* rpart, float* y,
                              float* rsum, float* unorm, void* stream);
int pn_meanshift_iter_bwd_f32(const float* gy, const float* y, const float* q, const float* x,
                              const float* xt, const float* rsum, const float* unorm,
                              const float* bsq, int B, int N, int D, float* gu, float* go,
                              float* cs, float* qt, float* gut, float* opart_q, float* opart_x,
                              float* gq, float* gx, void* stream);

/* ---- mean-shift, fp32-grade products on the bf16 matrix cores ("bf16 x 3") ---------------
 * Same contract and outputs as the entry points above; every fp32 operand is split without
 * error into three bf16 pieces and each product is formed from the six significant piece
 * products with fp32 accumulation (an fp32 dot product in a different summation order).
 *   image_bytes : size of one tile-image array for (B,N).
 *   split       : x (B,N,D) -> img (pre-split, LDS-ready image of every 32-point tile; both
 *                 GEMMs read it, the second one through the hardware transpose read).
 *   iter_fwd    : as pn_meanshift_iter_fwd_f32 with img_x in place of (x, xt).
 *   iter_bwd    : as pn_meanshift_iter_bwd_f32 with img_x in place of xt and two scratch image
 *                 arrays (images of q and gu) in place of (go, qt, gut). */
size_t pn_meanshift_x3_image_bytes(int B, int N);
int pn_meanshift_x3_split_f32(const float* x, int B, int N, int D, void* img, void* stream);
int pn_meanshift_x3_iter_fwd_f32(const float* q, const void* img_x, const float* bsq, int B, int N,
                                 int D, float* opart, float* rpart, float* y, float* rsum,
                                 float* unorm, void* stream);
int pn_meanshift_x3_iter_bwd_f32(const float* gy, const float* y, const float* q, const float* x,
                                 const void* img_x, const float* rsum, const float* unorm,
                                 const float* bsq, int B, int N, int D, float* gu, float* cs,
                                 void* img_q, void* img_gu, float* opart_q, float* opart_x, float* gq,
                                 float* gx, void* stream);

/* Block-sparse variant.  K_ij = exp((q_i . x_j - 1) / b^2) of src/mean_shift.py:58-64 decays fast
 * on a clustered embedding; the (32-row tile of q) x (32-row tile of x) pairs a plan skips are
 * chosen, from rigorous bounds on the tiles' bounding caps on the unit sphere, such that for EVERY
 * row of q all skipped terms together stay below rel_eps of that row's sum (the caller's choice;
 * the host side uses 1e-6, below the rounding of an fp32 sum of N terms) — the result is the dense
 * one to fp32 noise.  Per q cap A: L_A = lower bound of its rows' best dot product, R_A = sum over
 * the data caps B of n_B exp((Lo_AB - L_A) / b^2) (n_B rows, Lo_AB <= every dot product of the
 * pair) bounds every row sum from below by exp((L_A - 1) / b^2) R_A, and the data caps with the
 * smallest upper bounds U_AB are dropped as long as sum n_B exp((U_AB - L_A) / b^2) <= rel_eps R_A.
 * The caller orders the points so that tiles are local (any order is valid; mean-shift is
 * permutation-equivariant).
 *   tileinfo : z (B,N,D) unit rows -> two bounding caps per tile (its rows dealt to two far-apart
 *              seeds): cen (B,T,2,D) normalised means, rho (B,T,2) angular radii (+1e-3 slack;
 *              < 0: empty cap), cnt (B,T,2) rows per cap (may be NULL), T = align_up(N,64)/32.
 *   plan     : caps of the iterate q and of the data x (cntX: the data caps' row counts; NULL: one
 *              row per non-empty cap in R, 32 in the dropped mass — rigorous, keeps more pairs)
 *              -> plan (pn_meanshift_x3_plan_bytes), N <= 32768:
 *              the pair predicate (T x T bytes), per resident block of each pass the compact
 *              list of streamed tiles with at least one pair set, and the prefix sums of the list
 *              lengths (the launches cut the concatenated lists into one equal range per CU).
 *   iter_fwd_plan / iter_bwd_plan : as iter_fwd / iter_bwd, skipping what the plan of THIS
 *              iteration excludes (plan == NULL: dense; the backward must get the forward's plan).
 *   chain_order : sim (B,P,P) similarities of P = 128 cell centres -> rank (B,P) int32, the position
 *              of every cell in the greedy nearest-neighbour chain from cell 0 (the order the
 *              caller lays the cells, hence the points, out in: neighbours in the sequence are
 *              neighbours on the sphere).  Only a heuristic for locality; never affects results.
 */
int pn_meanshift_chain_order_f32(const float* sim, int B, int P, int* rank, void* stream);
/* Executed work of the bf16 x 3 mean-shift launches, counted by the launches themselves: every wave
 * counts the (resident 32-row tile, streamed 32-row tile) pairs whose GEMMs it runs (all pairs in a
 * dense launch, the pairs the plan keeps in a planned one).  out3[0..2] = pairs executed by the
 * forward / row-pass / column-pass launches since the previous call; reading clears the counters
 * and synchronises with the device.  One pair is 2 * 32 * 32 * 128 FLOP per GEMM unit (2 / 3 / 4
 * units) x 6 piece products on the bf16 matrix cores. */
int pn_meanshift_x3_exec_tiles(unsigned long long* out3);
/* Nearest candidate of every query — arg-max_j xq_i . xc_j, the first step of MeanShift.nms
 * (src/mean_shift.py:146-149) — by EXACT pruning: xq, xc (B,N,D) unit rows in one common order that
 * keeps neighbours together (the locality order of the iterations), their tile caps from
 * pn_meanshift_x3_tileinfo_f32.  A candidate tile whose upper bound lies below the query tile's
 * guaranteed best cannot hold a maximum; only the remaining tile pairs are evaluated, as fp32 fma
 * chains over the channels in order (the arithmetic of pn_dot_select_f32, bit for bit).  perm (B,N)
 * position -> original index (NULL: identity): nearest (B,N) is indexed by the query's ORIGINAL
 * index and holds the candidate's ORIGINAL index, ties to the smaller one.  workspace:
 * pn_meanshift_x3_plan_bytes(B, N). */
int pn_meanshift_x3_nearest_f32(const float* xq, const float* xc, const float* cenQ, const float* rhoQ,
                                const float* cenC, const float* rhoC, const int64_t* perm, int B, int N, int D,
                                int64_t* nearest, void* workspace, size_t workspace_bytes, void* stream);
/* fp32-grade GEMM of the per-point layers (nn.Conv1d(kernel_size=1): src/model.py:56-180,
 * src/PointNet.py:143-289) on the bf16 matrix cores (csrc/gemm_x3.hip): operands split error-free into
 * three bf16 pieces, six piece products per fp32 product, fp32 accumulation.
 * pn_gemm_x3_weight_image_f32: the pre-split image of a weight W (M,K) row-major for y = W x
 * (transposed = 0) or of W^T for gx = W^T gy (transposed = 1; the image then belongs to a (K,M) operand);
 * img: pn_gemm_x3_weight_image_bytes(M, K) resp. (K, M) bytes.
 * pn_gemm_x3_f32: out (B,M,N) = A x (+ bias (M) or NULL), img_a the image of the (M,K) operand A,
 * x (B,K,N) channel-first; workspace: pn_gemm_x3_points_image_bytes(B, K, N) bytes. */
size_t pn_gemm_x3_weight_image_bytes(int M, int K);
size_t pn_gemm_x3_points_image_bytes(int B, int C, int N);
int pn_gemm_x3_weight_image_f32(const float* w, int M, int K, int transposed, void* img, void* stream);
int pn_gemm_x3_f32(const void* img_a, const float* x, const float* bias, int B, int M, int K, int N, float* out,
                   void* workspace, size_t workspace_bytes, void* stream);
/* pn_gemm_x3_cat_f32: the same product with the K input channels spread over nsrc <= 4 tensors xs[s] (B, cs[s], N)
 * (every cs[s] a multiple of 8, K = sum cs): W applied to the concatenation torch.cat builds in src/model.py:150
 * without writing it.  xs, cs: HOST arrays of nsrc device pointers / channel counts. */
int pn_gemm_x3_cat_f32(const void* img_a, const float* const* xs, const int* cs, int nsrc, const float* bias, int B,
                       int M, int N, float* out, void* workspace, size_t workspace_bytes, void* stream);
/* The weight gradient of such a layer (what autograd's conv1d backward forms over the B N points:
 * src/model.py:157-176, src/PointNet.py:196-284 under loss.backward()): gw (M,K) = sum_b gy[b] (M,N) x[b]^T (N,K)
 * in the same arithmetic, split over the points with a FIXED-ORDER sum of the partial results (bit-reproducible,
 * no atomics); gb (M) or NULL: the bias gradient sum_b sum_n gy[b][m][n].  gy (B,M,N), x (B,K,N) channel-first;
 * workspace: pn_gemm_x3_wgrad_workspace(B, M, K, N) bytes. */
size_t pn_gemm_x3_wgrad_workspace(int B, int M, int K, int N);
int pn_gemm_x3_wgrad_f32(const float* gy, const float* x, int B, int M, int K, int N, float* gw, float* gb,
                         void* workspace, size_t workspace_bytes, void* stream);
/* Mean-shift backward restricted to R <= 64 rows per batch item (csrc/meanshift_rows.hip).  A step of
 * src/mean_shift.py:45-79 maps row i of the iterate to a function of that row and of the data alone, and
 * the training path reads the final iterate only at the cluster centres (src/mean_shift.py:36-43): the
 * gradient is zero outside those rows in every step on the way back.  One call = one step: gy, y, q
 * (B,R,D) are the R rows of the incoming gradient, of the step's result and of its input iterate, rsum,
 * unorm (B,R) the saved row sums and pre-normalisation norms of those rows, x (B,N,D) the data, bsq (B)
 * the squared bandwidths.  Writes gq (B,R,D), the gradient w.r.t. the input rows, and ADDS the step's
 * gradient w.r.t. the data into gx (B,N,D).  D = 128.  No atomics.  workspace:
 * pn_meanshift_rows_bwd_workspace(B, N) bytes. */
size_t pn_meanshift_rows_bwd_workspace(int B, int N);
int pn_meanshift_rows_bwd_f32(const float* gy, const float* y, const float* q, const float* rsum, const float* unorm,
                              const float* x, const float* bsq, int B, int N, int D, int R, float* gq, float* gx,
                              void* workspace, size_t workspace_bytes, void* stream);
/* gx[b, rows[b,r], :] += g[b, r, :], r ascending (rows (B,R) int64 may repeat; entries outside [0, N)
 * are skipped): the first iterate of the mean-shift iterations IS the data. */
int pn_meanshift_rows_scatter_add_f32(const float* g, const int64_t* rows, int B, int N, int D, int R, float* gx,
                                      void* stream);
size_t pn_meanshift_x3_plan_bytes(int B, int N);
/* the part of a plan its consumers read (the rest is scratch of the plan call: plans of T iterations may be placed
 * core_bytes apart in one buffer of T * core + (plan_bytes - core) bytes) */
size_t pn_meanshift_x3_plan_core_bytes(int B, int N);
/* Spherical k-means steps of the locality order the block-sparse iterations run on (no counterpart in the
 * reference: mean-shift, src/mean_shift.py:45-79, is permutation-equivariant and the order is free).
 * x (B,N,128) unit rows, cen / old (B,K,128); lab (B,N) int32 = arg-max over the centres of the dot product
 * (ties -> the smaller centre); new centre = normalised sum of the cell's points in index order, the old
 * centre when the cell is empty.  Deterministic. */
int pn_kmeans_assign_f32(const float* x, const float* cen, int B, int N, int D, int K, int* lab, void* stream);
int pn_kmeans_centres_f32(const float* x, const int* lab, const float* old, int B, int N, int D, int K,
                          float* cen, void* stream);
/* The locality order itself (mean_shift.locality_order; no counterpart in the reference): perm (B,N) int64 = the
 * STABLE argsort of key[n] = rank[home[fine[n]]] * F + fine[n] — rank (B,P) int32 position of a coarse cell in the
 * chain, home (B,F) int32 coarse cell of a fine cell, fine (B,N) int32 fine cell of a point — as a counting sort
 * over the F <= 768 cells, one launch.  Identical to the tensor library's argsort(stable=True) of the keys. */
int pn_cell_order_i32(const int* rank, const int* home, const int* fine, int B, int N, int P, int F,
                      long long* perm, void* stream);
int pn_meanshift_x3_tileinfo_f32(const float* z, int B, int N, int D, float* cen, float* rho, float* cnt,
                                 void* stream);
int pn_meanshift_x3_plan_f32(const float* cenQ, const float* rhoQ, const float* cenX, const float* rhoX,
                             const float* cntX, const float* bsq, int B, int N, float rel_eps, void* plan,
                             void* stream);
int pn_meanshift_x3_iter_fwd_plan_f32(const float* q, const void* img_x, const float* bsq, int B, int N,
                                      int D, float* opart, float* rpart, float* y, float* rsum,
                                      float* unorm, const void* plan, void* stream);
/* the same, and the bounding caps of the result's tiles (what pn_meanshift_x3_tileinfo_f32 of y returns:
 * the q caps of the NEXT iteration's plan) out of the launch that combines the partial results; cen / rho
 * NULL: exactly pn_meanshift_x3_iter_fwd_plan_f32.  src/mean_shift.py:58-64, one iteration. */
int pn_meanshift_x3_iter_fwd_info_f32(const float* q, const void* img_x, const float* bsq, int B, int N,
                                      int D, float* opart, float* rpart, float* y, float* rsum,
                                      float* unorm, const void* plan, float* cen, float* rho, float* cnt,
                                      void* stream);
int pn_meanshift_x3_iter_bwd_plan_f32(const float* gy, const float* y, const float* q, const float* x,
                                      const void* img_x, const float* rsum, const float* unorm,
                                      const float* bsq, int B, int N, int D, float* gu, float* cs,
                                      void* img_q, void* img_gu, float* opart_q, float* opart_x,
                                      float* gq, float* gx, const void* plan, void* stream);

/* K-th largest dot product of every query row with both distance passes in bf16 x 3 arithmetic
 * (error-free operand split on the bf16 matrix cores, fp32 accumulate: fp32-grade values, not the
 * fma chain of pn_dot_select_f32) — the bandwidth statistic of src/mean_shift.py:125-137 under the
 * default mean-shift arithmetic.  q (B,Nq,C), c (B,Nc,C) point-major; workspace and flags as
 * pn_dot_select_f32 with want_value; PN_ERR_UNSUPPORTED outside C <= 128, Nc >= 2048. */
int pn_dot_kth_x3_f32(const float* q, int Nq, const float* c, int Nc, int B, int C, int k, float* out_val,
                      int* flags, void* workspace, size_t workspace_bytes, void* stream);

/* ---- K-th largest dot product between unit vectors, fp16 x 2 matrix-core passes -----------
 * The bandwidth statistic of src/mean_shift.py:125-137 only needs the VALUE of the K-th nearest
 * neighbour (to 1e-5 after averaging): same engine as pn_dot_select_f32(out_val), with both
 * distance passes on the fp16 matrix cores (scaled fp16 x 2 split, |error| ~1e-7 on the dot
 * products).  q (B,Nq,128) unit rows; img_c = pn_meanshift_h2_split_f32 of the candidates
 * (B,Nc,128); workspace as pn_dot_select_workspace(B,128,Nq,Nc,k,1); flags as pn_dot_select_f32. */
int pn_dot_kth_unit_h2_f32(const float* q, int Nq, const void* img_c, int Nc, int B, int D, int k,
                           float* out_val, int* flags, void* workspace, size_t workspace_bytes,
                           void* stream);

/* ---- mean-shift, fp32-grade products on the fp16 matrix cores ("fp16 x 2") ---------------
 * Same contract and outputs again; every operand is scaled by a power of two into the fp16
 * range and split into two fp16 pieces, each product is formed from the three significant
 * piece products with fp32 accumulation (error below that of an fp32 fma chain; see
 * csrc/meanshift_h2.h for the scaling rules).  Rows of x and of the iterates must be unit
 * vectors.  Entry points as the x3 ones; the backward takes ``rowsc`` (3 B N + B ntiles floats of
 * scratch, ntiles = 2 ceil(N / 64): per-row scalars and per-tile maxima) in place of ``cs``. */
size_t pn_meanshift_h2_image_bytes(int B, int N);
int pn_meanshift_h2_split_f32(const float* x, int B, int N, int D, void* img, void* stream);
int pn_meanshift_h2_iter_fwd_f32(const float* q, const void* img_x, const float* bsq, int B, int N,
                                 int D, float* opart, float* rpart, float* y, float* rsum,
                                 float* unorm, void* stream);
int pn_meanshift_h2_iter_bwd_f32(const float* gy, const float* y, const float* q, const float* x,
                                 const void* img_x, const float* rsum, const float* unorm,
                                 const float* bsq, int B, int N, int D, float* gu, float* rowsc,
                                 void* img_q, void* img_gu, float* opart_q, float* opart_x, float* gq,
                                 float* gx, void* stream);

/* ---- GroupNorm (+ReLU) (+max over points) of the per-point heads -------------------------
 * Replaces torch GroupNorm -> ReLU (-> max over N) of src/PointNet.py:216-218, 274-283 on
 * channel-first (B,C,N) tensors.  One block per (b,c) row.
 *   rows_fwd      : per-row sum and sum of squares (B,C); with rmax != NULL also the row maximum /
 *                   minimum and their positions (int32) for the max-over-N variant.
 *   group_moments : (B,groups) mean and rstd = 1/sqrt(var+eps) from the row sums (fp64 inside).
 *   apply_fwd     : out = [relu](gamma*(y-mean)*rstd + beta).
 *   rows_bwd      : ra = sum_n gz, rb = sum_n gz*yhat per row, gz = gout*[z>0] (or gout if !relu).
 *   group_bwd     : c1c2 (B,groups,2) = group means of gamma*gz and gamma*gz*yhat.
 *   apply_bwd     : dy = rstd*(gamma*gz - c1 - yhat*c2); with gsp/arg non-NULL gz is the sparse
 *                   gradient gsp (B,C) placed at position arg (B,C) of every row (gout ignored).
 * rowbias (NULL, or C floats with rb_bstride = 0: one value per channel, or B * C floats with rb_bstride = C: one per
 * (item, channel)): the four row kernels read y + rowbias[b * rb_bstride + c] — the preceding convolution's bias
 * (src/PointNet.py:196, 268-284) added at load instead of by a pass of its own; the same fp32 addition. */
int pn_gn_rows_fwd_f32(const float* y, int B, int C, int N, float* rsum, float* rsq, float* rmax,
                       int* amax, float* rmin, int* amin, const float* rowbias, int rb_bstride, void* stream);
int pn_gn_group_moments_f32(const float* rsum, const float* rsq, int B, int C, int groups, int N,
                            float eps, float* mean, float* rstd, void* stream);
int pn_gn_apply_fwd_f32(const float* y, const float* mean, const float* rstd, const float* gamma,
                        const float* beta, int B, int C, int groups, int N, int relu, float* out,
                        const float* rowbias, int rb_bstride, void* stream);
int pn_gn_rows_bwd_f32(const float* gout, const float* y, const float* mean, const float* rstd,
                       const float* gamma, const float* beta, int B, int C, int groups, int N,
                       int relu, float* ra, float* rb, const float* rowbias, int rb_bstride, void* stream);
/* The (B,C) tail of max_n relu(GroupNorm(y)) (src/PointNet.py:199-201: bnmlp1 + relu + max over the points, taken
 * on the row extrema): extremum by the sign of gamma, (ext - mean) * rstd, gamma * yhat + beta, relu — and of its
 * backward pass: gz = g * (z > 0), rb = gz * yhat.  The tensor-library operations they replace, separately rounded. */
int pn_gn_max_finish_f32(const float* rmax, const int* amax, const float* rmin, const int* amin,
                         const float* mean, const float* rstd, const float* gamma, const float* beta,
                         int B, int C, int groups, float* yhat, float* z, int* arg, float* out, void* stream);
int pn_gn_max_bwd_prep_f32(const float* g, const float* z, const float* yhat, int B, int C, float* gz, float* rb,
                           void* stream);
int pn_gn_group_bwd_f32(const float* ra, const float* rb, const float* gamma, int B, int C,
                        int groups, int N, float* c1c2, void* stream);
int pn_gn_apply_bwd_f32(const float* gout, const float* y, const float* mean, const float* rstd,
                        const float* gamma, const float* beta, const float* c1c2, int B, int C,
                        int groups, int N, int relu, const float* gsp, const int* arg, float* dy,
                        const float* rowbias, int rb_bstride, void* stream);

/* ---- batched symmetric 3x3 eigen-decomposition (fp64) ---------------------------------
 * Serves the right singular vectors the primitive fits need (torch.svd of a tall n x 3 matrix in
 * src/fitting_utils.py:440 used by src/primitive_forward.py:725,794): they are the eigenvectors
 * of the 3x3 Gram matrix.  G (M,3,3) row-major -> evals (M,3) descending, evecs (M,3,3) with
 * eigenvectors in COLUMNS, each signed so that its largest-magnitude component is positive. */
int pn_sym3_eig_f64(const double* G, int M, double* evals, double* evecs, void* stream);

/* ---- layout helper: (B,R,C) -> (B,C,R) --------------------------------------------- */
int pn_transpose_f32(const float* in, float* out, int B, int R, int C, void* stream);

/* ---- edge features, API form ------------------------------------------------------
 * Replaces the gather + repeat + cat of src/model.py:25-53 (get_graph_feature) and
 * src/PointNet.py:72-103, :106-140.  xt (B,N,C) is the point-major copy of x (the
 * reference builds it with x.transpose(2,1).contiguous(), model.py:42); idx (B,N,k) are
 * per-item indices (the reference's idx_base offset is applied internally).
 * feat (B,N,k,2C): [x_j - x_i | x_i] — the memory behind the (B,2C,N,k) permuted view the
 * reference returns.  The backward reduces a gradient of that shape into gxt (B,N,C): it
 * transposes the kNN graph (sorted CSR by target point, built in `workspace` of
 * pn_edgeconv_bwd_workspace(B, N, k) bytes) and every row of gxt is summed by one wave in list
 * order — no floating-point atomics, bit-reproducible. */
int pn_edge_feature_fwd_f32(const float* xt, const int64_t* idx, int B, int N, int k, int C,
                            float* feat, void* stream);
int pn_edge_feature_bwd_f32(const float* gfeat, const int64_t* idx, int B, int N, int k, int C,
                            float* gxt, void* workspace, size_t workspace_bytes, void* stream);

/* ---- fused edge convolution ---------------------------------------------------------
 * Replaces get_graph_feature -> Conv2d 1x1 (no bias) -> GroupNorm / BatchNorm2d ->
 * LeakyReLU -> max over k of src/PointNet.py:157-165,180-191,203-214 and
 * src/model.py:75-86,146-159, without the (B,2C,N,k) tensor: the convolution is applied to
 * points (PQ = xt @ [Wa ; Wb-Wa]^T, a plain GEMM done by the caller), the edge stage reduces
 * y = P[idx] + Q over the k neighbours.
 *   reduce_fwd : PQ (B,N,2*Cout), idx (B,N,k), gamma (Cout; only its sign is used: max where
 *                gamma >= 0, min otherwise) -> yext, s1 = sum_k y (B,N,Cout) fp32, argk uint8
 *                (B,N,Cout), stats fp64 [(per_sample ? B : 1)][groups][2] = sum y, sum y^2
 *                (per-workgroup partials in `workspace`, combined in index order: no atomics).
 *                per_sample = 1: GroupNorm statistics; 0: BatchNorm statistics (groups = Cout).
 *   moments    : stats -> mean, rstd = 1/sqrt(var + eps) (fp32), n = number of groups in total.
 *   finalize   : out (B,Cout,N) = LeakyReLU(gamma*(yext-mean)*rstd + beta).
 *   bwd_prep   : gout (B,Cout,N) -> gz = gout * LeakyReLU'(z), yhat, both (B,N,Cout).
 *   bwd        : exact Group/BatchNorm gradient on every edge -> dPQ (B,N,2*Cout);
 *                t = gamma*gz, c1c2 fp32 [(per_sample?B:1)][groups][2] = group means of t and
 *                t*yhat over the edge activations; dense = 0 when the statistics were constants
 *                (eval-mode BatchNorm).  Both edge terms are evaluated by transposing the kNN
 *                graph (CSR by target point, lists sorted by (source, slot), built inside the
 *                call in `workspace`) and gathering rows of Q, t and argk: every dP row is summed
 *                by one wave in list order.  No floating-point atomics: two calls on the same
 *                inputs return the same bits (the reference on one device is deterministic). */
size_t pn_edgeconv_reduce_workspace(int B, int N, int Cout, int groups);
int pn_edgeconv_reduce_fwd_f32(const float* PQ, const int64_t* idx, const float* gamma, int B,
                               int N, int k, int Cout, int groups, int per_sample, float* yext,
                               uint8_t* argk, float* s1, double* stats, void* workspace,
                               size_t workspace_bytes, void* stream);
int pn_edgeconv_reduce_fwd_i32(const float* PQ, const int32_t* idx, const float* gamma, int B,
                               int N, int k, int Cout, int groups, int per_sample, float* yext,
                               uint8_t* argk, float* s1, double* stats, void* workspace,
                               size_t workspace_bytes, void* stream);
int pn_moments_f32(const double* stats, int n, double count, float eps, float* mean, float* rstd,
                   void* stream);
int pn_edgeconv_finalize_fwd_f32(const float* yext, const float* mean, const float* rstd,
                                 const float* gamma, const float* beta, int B, int N, int Cout,
                                 int groups, int per_sample, float slope, float* out,
                                 void* stream);
int pn_edgeconv_bwd_prep_f32(const float* gout, const float* yext, const float* mean,
                             const float* rstd, const float* gamma, const float* beta, int B,
                             int N, int Cout, int groups, int per_sample, float slope, float* gz,
                             float* yhat, void* stream);
size_t pn_edgeconv_bwd_workspace(int B, int N, int k);
int pn_edgeconv_bwd_f32(const float* PQ, const int64_t* idx, const float* t, const float* s1,
                        const uint8_t* argk, const float* mean, const float* rstd,
                        const float* c1c2, int B, int N, int k, int Cout, int groups,
                        int per_sample, int dense, float* dPQ, void* workspace,
                        size_t workspace_bytes, void* stream);
int pn_edgeconv_bwd_i32(const float* PQ, const int32_t* idx, const float* t, const float* s1,
                        const uint8_t* argk, const float* mean, const float* rstd,
                        const float* c1c2, int B, int N, int k, int Cout, int groups,
                        int per_sample, int dense, float* dPQ, void* workspace,
                        size_t workspace_bytes, void* stream);
/* The transposed graph is a function of idx alone: pn_edgeconv_csr_build writes it into a workspace of
 * pn_edgeconv_bwd_workspace(B, N, k) bytes (idx_is_i32: int32 / int64 indices) — e.g. during the forward pass, on a
 * side stream — and pn_edgeconv_bwd_prebuilt runs the backward on it (same kernels, same result). */
int pn_edgeconv_csr_build(const void* idx, int idx_is_i32, int B, int N, int k, void* workspace, size_t workspace_bytes,
                          void* stream);
int pn_edgeconv_bwd_prebuilt(const float* PQ, const void* idx, int idx_is_i32, const float* t, const float* s1,
                             const uint8_t* argk, const float* mean, const float* rstd, const float* c1c2, int B, int N,
                             int k, int Cout, int groups, int per_sample, int dense, float* dPQ, void* workspace,
                             size_t workspace_bytes, void* stream);

/* ---- Chamfer nearest neighbour ---------------------------------------------------
 * Replaces the (M,N,3) broadcast + torch.min of src/utils.py:286-296 (chamfer_distance),
 * :313-323 (chamfer_distance_one_side), :338-358 (chamfer_distance_single_shape).
 * a (B,Na,3), b (B,Nb,3).  minA/argA (B,Na): squared distance to / index of the nearest
 * point of b for every point of a; minB/argB (B,Nb) the other way.  Any output pointer
 * may be NULL (a side whose two outputs are NULL is skipped).  Ties -> smallest index.
 * d = ((dx*dx + dy*dy) + dz*dz), each operation rounded to fp32. */
size_t pn_chamfer_nn_workspace(int B, int Na, int Nb);
int pn_chamfer_nn_f32(const float* a, const float* b, int B, int Na, int Nb, float* minA,
                      int64_t* argA, float* minB, int64_t* argB, void* workspace,
                      size_t workspace_bytes, void* stream);

/* Ragged batch: item i is a[offA[i] .. offA[i+1]) against b[offB[i] .. offB[i+1]) of two
 * concatenated clouds (totalA / totalB rows; offsets: B+1 device ints; maxA / maxB = largest item,
 * used to size the grid).  Outputs are concatenated like the inputs, indices are local to the
 * item.  Used for the spline segments of a step (src/primitives.py:204-206: 900 | 930 samples
 * against a different number of ground-truth points per segment) in ONE launch per direction. */
size_t pn_chamfer_nn_ragged_workspace(int totalA, int totalB);
int pn_chamfer_nn_ragged_f32(const float* a, const int* offA, int totalA, int maxA, const float* b,
                             const int* offB, int totalB, int maxB, int B, float* minA,
                             int64_t* argA, float* minB, int64_t* argB, void* workspace,
                             size_t workspace_bytes, void* stream);

/* Reduced two-sided distance of every item of the ragged batch and its backward (round 4):
 *   out[s] = (mean_i minA_i + mean_j minB_j) / 2      (src/utils.py:326-358, reduce=True)
 *   gpred[i] = (pred_i - gt[argA_i]) g_s / nA + sum_{j: argB_j = i} (pred_i - gt_j) g_s / nB
 * pred / minA / argA are concatenated over the items like the clouds (offA, offB: S+1 ints),
 * indices local to the item.  Fixed summation order, no atomics. */
int pn_chamfer_ragged_reduce_f32(const float* minA, const int* offA, const float* minB, const int* offB, int S,
                                 float* out, void* stream);
int pn_chamfer_ragged_bwd_f32(const float* pred, const int* offA, int maxA, const float* gt, const int* offB,
                              const int64_t* argA, const int64_t* argB, const float* g, int S, float* gpred,
                              void* stream);

/* Gradient of out[b,m,:] = src[b,idx[b,m],:] for rows of 3 floats (the nearest-neighbour gather of
 * src/utils.py:273-358's min over the broadcast): gsrc[b,i,:] = sum_{m: idx[b,m] = i} g[b,m,:] in
 * ascending m, gathered per row — no atomics.  g (B,M,3), idx (B,M), gsrc (B,N,3) fully written. */
int pn_gather_rows3_bwd_f32(const float* g, const int64_t* idx, int B, int M, int N, float* gsrc, void* stream);

/* ---- batched per-segment fitting (pn_weighted_moments, pn_small_lstsq,
 *      pn_bspline_eval) ------------------------------------------------------------------
 * Replace the serial per-segment Python loop of src/primitive_forward.py:925-1047
 * (fit_one_shape_torch), the four fits :708-843 (Fit.fit_{plane,sphere,cylinder,cone}_torch),
 * src/fitting_utils.py:32-85 (LeastSquares.lstsq, best_lambda), :385-455 (CustomSVD and its
 * custom backward), src/primitives.py:58-206 (ComputePrimitiveDistance) and the per-segment
 * loop of src/residual_utils.py:154-208 — for ALL analytic segments of ALL shapes of a step.
 *
 * Segment table (device int32 arrays of S entries): seg_shape = shape index b, seg_row = row of
 * the membership matrix W (B,Cp,N) holding the segment's soft weights, seg_type = 0 plane,
 * 1 sphere, 2 cylinder, 3 cone, seg_rows = number of fitted points (rank tolerance).  The fit
 * uses the points stride*j of the shape (fit_one_shape_torch keeps every 4th point for
 * analytic primitives) with weight W[b,row,stride*j] + eps.
 *
 *   pn_weighted_moments_f64   partial (S, pn_weighted_moments_chunks(), pn_weighted_moments_count())
 *                             fp64: sums of w^e * monomial(p, n), e <= 3, degree <= 3 (table in
 *                             csrc/fit_math.h) — everything the four fits need, one pass.
 *   pn_primitive_fit_f64      (the "pn_small_lstsq" of the survey, fused with the 3x3
 *                             eigen-decomposition) moments -> params (S,16) fp64 and the Jacobian
 *                             jac (S,16,64) = d params / d moments; 3x3 Jacobi, normal equations,
 *                             torch.matrix_rank-style rank test and the 1e-6*10^i ridge search on the
 *                             device.  params: plane a(3),d | sphere c(3),r | cylinder axis(3),c(3),r |
 *                             cone apex(3),axis(3),theta; slot 15 = ridge lambda used.  status bits:
 *                             1 = non-finite / no full-rank system (the reference raises),
 *                             2 = null cone (cond > 1e5: zero cone, no gradient), 4 = NaN residual.
 *   pn_cone_angle_f64         second pass of fit_cone_torch: theta (params slot 6), its Jacobian
 *                             row, and cone_direct (S) = factor of the direct path d theta / d w_i.
 *   pn_primitive_residual_f32 mean (squared, or guard_sqrt'ed if sqrt_flag) distance of the
 *                             ground-truth points gt_idx[gt_off[s] .. gt_off[s+1]) (indices into
 *                             the shape's points) to the primitive: dist (S) fp32, and
 *                             dparam (S,16) fp64 = d dist / d params.
 *   pn_weighted_moments_bwd_f32  gW (B,Cp,N) fp32 (zero-initialised by the caller) receives
 *                             d loss / d W at the fitted points given g_dist (S) = d loss / d dist. */
int pn_weighted_moments_chunks(void);
int pn_weighted_moments_count(void);
int pn_weighted_moments_f64(const float* P, const float* Nrm, const float* W, int B, int N, int Cp,
                            int stride, float eps, const int* seg_shape, const int* seg_row, int S,
                            double* partial, void* stream);
int pn_primitive_fit_f64(const double* partial, const int* seg_type, const int* seg_rows, int S,
                         double* params, double* jac, int* status, void* stream);
int pn_cone_angle_f64(const float* P, const float* W, int B, int N, int Cp, int stride, float eps,
                      const int* seg_shape, const int* seg_row, const int* seg_type,
                      const int* status, int S, double* params, double* jac, double* cone_direct,
                      void* stream);
int pn_primitive_residual_f32(const float* P, int B, int N, const int* seg_shape, const int* seg_type,
                              const int* gt_off, const int* gt_idx, int S, const double* params,
                              int sqrt_flag, float* dist, double* dparam, int* status, void* stream);
int pn_weighted_moments_bwd_f32(const float* P, const float* Nrm, const float* W, int B, int N, int Cp,
                                int stride, float eps, const int* seg_shape, const int* seg_row,
                                const int* seg_type, int S, const float* g_dist, const double* dparam,
                                const double* jac, const double* params, const double* cone_direct,
                                float* gW, void* stream);

/* B-spline surface evaluation, src/fitting_utils.py:609-622 sample_points_from_control_points_:
 * out[s,u,v,:] = A_s (sum_ij nu[u,i] nv[v,j] ctrl[s,i,j,:]) + t_s with nu (gu,cu), nv (gv,cv),
 * ctrl (S,cu,cv,3); affine (S,3,4) = [A_s | t_s] or NULL (the de-standardisation of
 * src/primitive_forward.py:60-72 folded in); wrap = 1 appends the first u-row again (closed
 * splines, :377-385): out (S,(gu+wrap)*gv,3).  _bwd: gctrl (S,cu,cv,3) from gout. */
int pn_bspline_eval_f32(const float* nu, const float* nv, const float* ctrl, const float* affine, int S,
                        int gu, int gv, int cu, int cv, int wrap, float* out, void* stream);
int pn_bspline_eval_bwd_f32(const float* nu, const float* nv, const float* gout, const float* affine,
                            int S, int gu, int gv, int cu, int cv, int wrap, float* gctrl, void* stream);

/* ---- round-3 fusions (csrc/fused.hip) ------------------------------------------------------
 * pn_edgeconv_bwd_stats_f32: step A of the fused edge-conv backward WITH its reductions (what
 * src/model.py:127-156 / src/PointNet.py:186-205 leave to autograd through Conv2d -> Norm ->
 * LeakyReLU -> max): from gout (B,Cout,N) and the forward's yext (B,N,Cout), mean / rstd writes
 *   t (B,N,Cout) = gamma * gout * LeakyReLU'(z),  dgamma (Cout), dbeta (Cout) and the group means
 *   c1c2 ((per_sample ? B : 1), groups, 2) that pn_edgeconv_bwd_f32 consumes (zeros when !dense:
 *   evaluation-mode BatchNorm).  Fixed-order partial sums, fp64 combination. */
size_t pn_edgeconv_bwd_stats_workspace(int B, int N, int Cout);
int pn_edgeconv_bwd_stats_f32(const float* gout, const float* yext, const float* mean, const float* rstd,
                              const float* gamma, const float* beta, int B, int N, int k, int Cout,
                              int groups, int per_sample, int dense, float slope, float* t, float* dgamma,
                              float* dbeta, float* c1c2, void* workspace, size_t workspace_bytes,
                              void* stream);

/* Triplet embedding loss, src/segment_loss.py:85-123 (EmbeddingLoss.triplet_loss, the arithmetic
 * after the numpy sampling): E (rows,D) unit-row embedding (D = 128); item p of P has num <= 32
 * anchor/positive rows ia[p][:] and negative rows ib[p][:] (row indices into E) and a weight w[p]
 * (1 / (pairs of its shape + 1e-8));  c_ij = relu(|a_i - p_j|^2 - |a_i - n_j|^2 + margin),
 * item_loss[p] = w[p] * (sum_ij c_ij - sum_i c_ii) / (#(c_ij > 0) + 1), loss[0] = sum_p item_loss[p];
 * item_scale[p] = w[p] / (# + 1) is kept for the backward.  _bwd STORES gout[0] * d loss / d E into
 * the rows of gE (rows,D) that some item names and leaves the others as they are (the caller
 * zeroes gE).  A point can be sampled more than once: the per-item gradient rows go to `workspace`
 * (pn_triplet_bwd_workspace bytes) and rows naming the same point are added in item order — no
 * atomics, bit-reproducible. */
int pn_triplet_fwd_f32(const float* E, int rows, int D, const int64_t* ia, const int64_t* ib, const float* w,
                       int P, int num, float margin, float* item_loss, float* item_scale, float* loss,
                       void* stream);
size_t pn_triplet_bwd_workspace(int P, int num, int D);
int pn_triplet_bwd_f32(const float* E, int rows, int D, const int64_t* ia, const int64_t* ib,
                       const float* item_scale, const float* gout, int P, int num, float margin, float* gE,
                       void* workspace, size_t workspace_bytes, void* stream);

/* Memberships of the fitting stage in one pass: src/residual_utils.py:120 (weights = center @
 * embedding^T), src/fitting_utils.py:306-325 (weights_normalize) and the labels of
 * src/mean_shift.py:176-178 (arg-max over the centres, first index on ties).
 * cen (B,CP,D) centre rows padded to CP in {16,32,64} (rows >= ncl[b] ignored), emb (B,N,D), D = 128,
 * bw (B), ncl (B) int64.  Outputs Wraw, prob, Wn (B,CP,N) (padding rows of prob / Wn are zero),
 * rowstat (B,CP,4) = (min_n prob, max_n (prob - min) + eps, arg-min, arg-max as int bits), labels
 * (B,N) int64 or NULL.  _bwd: gWraw (B,CP,N) from gWn, the exact gradient of weights_normalize
 * (min / max route to their first arg-min / arg-max column; clamp passes inclusively); rowgrad
 * (B,CP,2) is scratch.  d cen and d emb are then two GEMMs on gWraw (the caller's). */
int pn_membership_fwd_f32(const float* cen, const float* emb, const float* bw, const int64_t* ncl, int B,
                          int CP, int N, int D, float eps, float* Wraw, float* prob, float* Wn,
                          float* rowstat, int64_t* labels, void* stream);
int pn_membership_bwd_f32(const float* gWn, const float* Wraw, const float* prob, const float* rowstat,
                          const float* bw, const int64_t* ncl, int B, int CP, int N, float* rowgrad,
                          float* gWraw, void* stream);

/* Non-maximum suppression of the shifted points, src/mean_shift.py:139-179, without leaving the
 * device.  pn_nms_occupied_f32: membership (B,N) int64 = nearest shifted point of every input point
 * (pn_dot_select_f32, k = 1) -> counts (B,N) int32 members per shifted point (np.unique's counts,
 * :150-153), uq (B,U) int64 the occupied ones in ascending order (zero-padded), nocc (B) their number
 * (may exceed U: the caller retries with a larger U).  pn_nms_vote_f32: G (B,U,U) = Cu Cu^T of the
 * occupied centres -> every occupied centre votes for the FIRST arg-max of [2 - 2 G < bw] * counts
 * (:160-168; distance < b, not b^2, like the reference; only occupied columns can win: an
 * unoccupied one scores 0 and the centre itself scores its own count) -> hits (B,N) int32 scratch,
 * cid (B,cmax) int64 the voted centres in ascending order (zero-padded), ncl (B) their number. */
int pn_nms_occupied_f32(const int64_t* membership, int B, int N, int U, int* counts, int64_t* uq, int64_t* nocc,
                        void* stream);
int pn_nms_vote_f32(const float* G, const int64_t* uq, const int64_t* nocc, const int* counts, const float* bw,
                    int B, int N, int U, int cmax, int* hits, int64_t* cid, int64_t* ncl, void* stream);

/* y = act(x * scale[c] + shift[c]) on (B,C,N): evaluation-mode BatchNorm1d folded with the
 * activation that follows it (src/model.py:160-176: conv5/bn5 LeakyReLU(0.2), conv6/bn6 and
 * conv7/bn7 ReLU of the frozen SplineNets).  act: 0 none, 1 ReLU, 2 LeakyReLU(slope).
 * _bwd: gx = gy * scale[c] * act'(y). */
int pn_affine_act_fwd_f32(const float* x, const float* scale, const float* shift, int B, int C, int N,
                          int act, float slope, float* y, void* stream);
int pn_affine_act_bwd_f32(const float* gy, const float* y, const float* scale, int B, int C, int N, int act,
                          float slope, float* gx, void* stream);

/* out[s][c] = max_n act(x[s][c][n] * scale[c] + shift[c]) * w[s][n]: the frozen SplineNet's
 * conv5 -> bn5 (evaluation mode) -> LeakyReLU, "x *= weights" and the max pool over the points
 * (src/model.py:160-170) in one pass; idx (S,C) int32 first arg-max, val (S,C) the activation there.
 * _bwd: gw (S,N) = d / d w from g (S,C) (no gradient to x: the network is frozen); N <= 13 600 (12 bytes of LDS per point). */
int pn_weighted_max_fwd_f32(const float* x, const float* scale, const float* shift, const float* w, int S, int C,
                            int N, int act, float slope, float* out, int* idx, float* val, void* stream);
int pn_weighted_max_bwd_f32(const float* g, const int* idx, const float* val, int S, int C, int N, float* gw,
                            void* stream);

/* Standardisation of S spline segments (src/fitting_utils.py:512-553, standardize_point_torch): the two steps that
 * are not arithmetic on the points (mean, covariance and rotation stay tensor expressions: csrc/fused.hip says why).
 *   select: sel (S,n) bytes = w > 0.8, or — fewer than 400 such points — the kf largest memberships (ties to the
 *           smaller index);
 *   scale : std (S,3) = | max - min | over the selected points of Pr * w per axis, pts (S,n,3) = Pr / (std + eps),
 *           Pr (S,n,3) the rotated centred points.
 * w (S,n) fp32; one 256-thread workgroup per segment; exact operations, bit-identical to the tensor-library form. */
int pn_standardize_select_f32(const float* w, int S, int n, int kf, unsigned char* sel, void* stream);
int pn_standardize_scale_f32(const float* Pr, const float* w, const unsigned char* sel, int S, int n, float eps,
                             float* pts, float* stdv, void* stream);
/* Adam (torch.optim.Adam's defaults and update rule: train_parsenet.py:96, train_parsenet_e2e.py:88,
 * train_open_splines.py:81) on ONE flat fp32 buffer of n parameters: p, the gradients g and both moments m, v are
 * contiguous arrays of n floats; step = the 1-based count of this update.  One launch for the whole model. */
int pn_adam_flat_f32(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                     float eps, int step, void* stream);

/* Gradient tensors into the flat data-parallel bucket (the buffer the ONE all-reduce of a step works on; the
 * reference's DataParallel reduces per parameter, train_parsenet.py:90-91): tensor e = ns[e] floats at srcs[e] goes
 * to flat + offs[e].  srcs / offs / ns are HOST arrays of count entries; 64 tensors per launch. */
int pn_gather_flat_f32(const float* const* srcs, const long long* offs, const long long* ns, int count, float* flat,
                       void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PARSENET_HIP_H */
