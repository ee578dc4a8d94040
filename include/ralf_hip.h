/* libralf_hip.so -- C ABI of the MI355X-native (gfx950) RALF hot path.
 *
 * The reference (CyberAgentAILab/RALF @ 2024_08_07) is 100 % Python on stock torch.nn; it has no
 * FFI of its own.  Each entry point below replaces the torch / faiss call the reference makes at
 * the cited file:line (paths relative to the reference root) and is what a ctypes binding in the
 * reference would bind.
 *
 * Conventions (all entry points):
 *   - return 0 (RALF_OK) or a negative RALF_ERR_* code; message via ralf_last_error() (thread-local)
 *   - never throw, never allocate or free caller memory, never synchronise the stream
 *   - every pointer is a caller-owned DEVICE pointer (HBM), contiguous, 16-byte aligned
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream)
 *   - scratch memory is caller-provided: ask ralf_<op>_workspace_bytes() first
 */
#ifndef RALF_HIP_H
#define RALF_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RALF_ABI_VERSION 26
#define RALF_OK 0
#define RALF_ERR_INVALID (-1)   /* bad argument / unsupported shape */
#define RALF_ERR_WORKSPACE (-2) /* workspace too small */
#define RALF_ERR_LAUNCH (-3)    /* HIP launch failure */

const char* ralf_last_error(void);
int ralf_abi_version(void);

/* ---------------------------------------------------------------------------------------------
 * Exact inner-product top-k scan (replaces faiss.IndexFlat(d, METRIC_INNER_PRODUCT).search reached
 * through datasets.Dataset.get_nearest_examples at
 * image2layout/train/models/retrieval/retriever.py:200-202, :270-274, :322-324 and
 * image2layout/train/models/retrieval/cross_retriever.py:191-193).
 *   index   float32 [n_db, dim] row-major, HBM resident        queries float32 [nq, dim]
 *   out_idx int64 [nq, k]   out_score float32 [nq, k], sorted by (score desc, index asc);
 *   rows beyond n_db (k > n_db) are (-inf, -1).
 * score(q, n) = fmaf chain over ascending dimension (one rounding per product, fp32 accumulate) --
 * bit-identical to oracle/knn_oracle.c.  Requires dim % 4 == 0, 1 <= k <= 1024.
 * ------------------------------------------------------------------------------------------- */
size_t ralf_knn_topk_ip_workspace_bytes(int64_t n_db, int dim, int nq, int k);
int ralf_knn_topk_ip(const float* index, int64_t n_db, int dim, const float* queries, int nq, int k,
                     int64_t* out_idx, float* out_score, void* workspace, size_t workspace_bytes, void* stream);
/* The two phases separately (tests / profiling): raw score matrix [nq, n_db] and selection from it. */
int ralf_knn_scores(const float* index, int64_t n_db, int dim, const float* queries, int nq, float* scores, void* stream);
int ralf_knn_select(const float* scores, int64_t n_db, int nq, int k, int64_t* out_idx, float* out_score,
                    void* workspace, size_t workspace_bytes, void* stream);
/* exact re-scoring for the two-stage search (bf16 coarse ranking by ralf_gemm, then this, then ralf_knn_select):
 * out[q][j] = <Q[q], X[cand[q][j]]> for cand int64 [nq, pool] (row indices, every value in [0, N)); same accumulation chain as
 * ralf_knn_scores, hence bit-identical scores */
int ralf_knn_rescore(const float* X, int64_t N, int D, const float* Q, int nq, const int64_t* cand, int pool, float* out, void* stream);
/* last step of the two-stage search (replaces the sort / gather / norm arithmetic around faiss-style re-ranking): the k best of
 * `pool` re-scored candidates per query by (score desc, row index asc) -> out_idx / out_score [nq, k]; when `bad` (int32 [nq]) is
 * given it also receives the per-query certificate  bad[q] = !(k-th exact score >= bound[q * bound_ld] + eps[q])  with
 * eps = |q - qb| max|x| + |qb| max|x - xb| + 2 D 2^-24 |q| max|x| (bf16 rounding of both operands + fp32 accumulation of both
 * passes): qnorms fp32 [nq, 3] and xnorms fp32 [3] from ralf_knn_rownorms.  pool <= 1024. */
int ralf_knn_select_cand(const float* exact, const int64_t* cand, int nq, int pool, int k, int64_t* out_idx, float* out_score,
                         const float* bound, int64_t bound_ld, const float* qnorms, const float* xnorms, int D, int32_t* bad, void* stream);
/* per row of X fp32 [R, D] and its bf16 copy Xb: norms[r] = {|x|, |xb|, |x - xb|} (fp32 [R, 3], may be NULL); maxes fp32 [3]
 * (may be NULL, zero on entry) = column maxima over the rows.  Values are rounded UP (they feed an upper bound). */
int ralf_knn_rownorms(const float* X, const void* Xb_bf16, int64_t R, int D, float* norms, float* maxes, void* stream);
/* The whole two-stage search as ONE call (what `index.search(x, k)` costs for a batch of queries: retriever.py:200-202 loops faiss over single
 * queries; cross_retriever.py:133-207): queries -> bf16 + norms, coarse scores on the bf16 matrix cores, the pool + 1 best coarse rows, their exact
 * fp32 scores, the k best with the per-query certificate.  bad[q] = 1: not certified -- the caller sends those queries through ralf_knn_topk_ip
 * (the results of certified queries ARE ralf_knn_topk_ip's, bit for bit).  Xb: the bf16 copy of the index (ralf_copy2d), xnorms fp32 [3] = the
 * column maxima of ralf_knn_rownorms(X, Xb).  dim % 64 == 0, 1 <= k <= pool, pool + 1 <= 1024, pool < n_db; workspace 256-byte aligned. */
size_t ralf_knn_two_stage_workspace_bytes(int64_t n_db, int dim, int nq, int pool);
int ralf_knn_topk_ip_two_stage(const float* X, const void* Xb_bf16, int64_t n_db, int dim, const float* Q, int nq, int k, int pool, const float* xnorms,
                               int64_t* out_idx, float* out_score, int32_t* bad, void* workspace, size_t workspace_bytes, void* stream);
/* The same search without the [nq, n_db] coarse score matrix (written and re-read by the selection: 252 MB at BASELINE config 4's 1024 queries): a first
 * product against the first min(n_db, 4096) rows gives every query a lower bound of its (pool+1)-th best coarse score, the product over the whole index keeps
 * the scores at or above it in slot lists (RalfGemmDesc.flt_*, 16 slots per query and column tile), the selection reads the slots.  Same arguments, workspace
 * size and results; bad[q] is ALSO set when a column tile of the query had more hits than slots or fewer than pool + 1 rows reached the bound (an index whose
 * rows are ordered by similarity floods tiles: the caller should route such an index through ralf_knn_topk_ip_two_stage). */
int ralf_knn_topk_ip_two_stage_filtered(const float* X, const void* Xb_bf16, int64_t n_db, int dim, const float* Q, int nq, int k, int pool, const float* xnorms,
                                        int64_t* out_idx, float* out_score, int32_t* bad, void* workspace, size_t workspace_bytes, void* stream);
/* the candidate slots a filtered coarse pass wrote (RalfGemmDesc.flt_*: list int32 [nq][T][cap][2] = {row, score bits}, count int32 [nq][T]) as
 * the dense pair the selection kernels take: rows int64 [nq][T * cap] (0 in unused slots), scores fp32 [nq][T * cap] (-inf in unused slots);
 * over int32 [nq] (may be NULL, ZEROED by the caller) is set to 1 where a tile's count exceeds cap (the list lost candidates).
 * ralf_knn_gather_rows: out[q][j] = rows[q][pos[q][j]] (rows with a row length of cap). */
int ralf_knn_list_unpack(const int* list, const int* count, int nq, int T, int cap, int64_t* rows, float* scores, int* over, void* stream);
int ralf_knn_gather_rows(const int64_t* rows, int cap, const int64_t* pos, int nq, int m, int64_t* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * element types of activation / weight buffers (accumulation is always fp32)
 * ------------------------------------------------------------------------------------------- */
#define RALF_F32 0  /* parity mode: exact-fp32 matrix cores (v_mfma_f32_32x32x2_f32)            */
#define RALF_BF16 1 /* throughput mode: bf16 operands (v_mfma_f32_32x32x16_bf16), fp32 accum    */

#define RALF_ACT_NONE 0
#define RALF_ACT_RELU 1
#define RALF_ACT_GELU 2          /* exact erf GELU (nn.GELU default, common/attention.py:23)      */
#define RALF_ACT_RELU_POST 3     /* ReLU applied AFTER the residual: relu(colscale*acc + bias + res), the tail of a ResNet bottleneck in inference */
#define RALF_AUX_NONE 0
#define RALF_AUX_RELU_MASK 1     /* v = aux > 0 ? v * aux_scale : 0   (ReLU gradient from output) */
#define RALF_AUX_GELU_GRAD 2     /* v *= gelu'(aux)                   (aux = pre-activation)      */

/* implicit-im2col geometry: matrix rows = pixels of an (RH x RW) grid per image, columns =
 * (kh, kw, c) with c fastest, gathered from an NHWC source [*, SH, SW, SC].
 *   mode 0 (forward / weight-gradient): rows = output pixels, source = input x:
 *           sy = ry*stride - pad + kh
 *   mode 1 (data-gradient):            rows = input pixels,  source = dY:
 *           sy = (ry + pad - kh) / stride when divisible                                         */
typedef struct RalfConvGeom {
    int RH, RW, SH, SW, SC, KH, KW, stride, pad, mode;
} RalfConvGeom;

/* C[z][m][n] = epi(alpha * sum_k A[z][m][k] * B[z][k][n]),  z = z1*nb0 + z0  (two batch levels).
 * Replaces F.linear / F.conv2d / torch.bmm and their backward products, see ralf_amd/csrc/gemm.hip.
 *   a_kcontig: A stored [M][lda] (k contiguous) else [K][lda] (m contiguous)
 *   b_kcontig: B stored [N][ldb] (k contiguous, e.g. nn.Linear.weight) else [K][ldb]
 *   gather:    0 none; 1 = A is the im2col matrix of g (needs a_kcontig=b_kcontig=1);
 *              2 = B is the im2col matrix of g (weight gradient, needs a_kcontig=b_kcontig=0)
 *   epilogue order: *alpha, +bias[n] (fp32), C2 = v (optional pre-activation copy), act, dropout,
 *              aux mask/gradient, +res, (+C if accumulate), store as dtype or fp32 (out_f32).
 *   splitk > 1: deterministic split of the K range through `workspace` (fp32 partial slabs).   */
typedef struct RalfGemmDesc {
    const void* A; const void* B; void* C; void* C2;
    const float* bias; const void* res; const void* aux;
    int64_t lda, ldb, ldc, ldr;
    int64_t sA0, sA1, sB0, sB1, sC0, sC1, sR0, sR1;
    int M, N, K, nb0, nb1;
    int dtype, a_kcontig, b_kcontig, gather;
    int act, aux_mode, out_f32, accumulate, splitk;
    float alpha, aux_scale;
    RalfConvGeom g;
    /* fused dropout after the activation: v = keep ? v/(1-p) : 0 with keep = f(seed[0], call_id, m*N+n) -- the
     * same mask ralf_dropout produces on the contiguous [M,N] tensor (so the backward can regenerate it) */
    const int64_t* seed; uint64_t call_id; float drop_p;
    /* atomic_out: C (fp32) += alpha*A@B with fp32 atomics, split-K without partial slabs / reduce kernel
     * (weight gradients accumulated straight into the flat gradient buffer; summation order not fixed) */
    int atomic_out;
    /* colstats (fp32 [ceil(M/64)][2][N], may be NULL): per 64-row block of C, the column sums and sums of squares of
     * the STORED output (after rounding to its dtype) -- the BatchNorm batch statistics of a convolution output
     * come out of the convolution's own epilogue (ralf_bn_stats_from_partials) instead of a pass over the tensor.
     * Needs a plain epilogue (alpha 1, no bias/act/res/aux/dropout/accumulate), splitk 1, one batch, N % 64 == 0. */
    float* colstats;
    /* sBias0: element stride of `bias` per z0 batch entry (0 = one bias vector shared by the batch).
     * kseg / sBk (0 = off): the K range of B is a chain of segments of kseg rows (a multiple of the k-tile: 64 bf16 / 32 fp32),
     * segment s starting sBk elements beyond where a contiguous B would put it -- several weight matrices that live apart in
     * one flat parameter buffer act as ONE stacked operand (cross-attention K/V projections of all decoder layers over the
     * same memory: one data-gradient product instead of a product + accumulate per layer).  Needs the aligned interior path
     * (K a multiple of the k-tile, 16-byte aligned operands) and splitk == 1. */
    int64_t sBias0;
    int64_t sBk;
    int kseg;
    /* inference-time BatchNorm folded into the producing convolution: acc * colscale[n] (fp32 [N], may be NULL) before the bias,
     * i.e. y = act(conv(x) * gamma/sqrt(var+eps) + (beta - mean*gamma/sqrt(var+eps)) (+ res)) with scale/shift from ralf_bn_fold_batched */
    const float* colscale;
    /* BatchNorm-backward statistics out of a data-gradient GEMM (all four NULL = off).  C is the gradient dz of a tensor
     * z = relu(BN(x) (+ res)): the epilogue zeroes dz where the ReLU was inactive (bnb_mask: the 1-bit mask ralf_bn_apply wrote,
     * bit (m*N + n) & 7 of byte (m*N + n) >> 3; NULL = no ReLU) and writes, per 64-row block of C, the column sums of the STORED dz
     * and of dz * (x - mean[n]) to bnb_part (fp32 [ceil(M/64)][2][N]) -- the reductions of the BatchNorm backward
     * (ralf_bn_bwd_stats_from_partials) without a pass over dz and x (replaces the first half of torch's native_batch_norm_backward).
     * bnb_x: the BatchNorm INPUT x, contiguous [M][N] in the GEMM's dtype.  Needs splitk 1, one batch, N % 64 == 0, a contiguous
     * 16-byte aligned [M][N] output in the GEMM's dtype, epilogue limited to alpha / bias / res / accumulate. */
    const void* bnb_x; const unsigned char* bnb_mask; const float* bnb_mean; float* bnb_part;
    /* A-operand transform with write-through (at_mode 0 = off): the matrix cores see a'(m, k) = f(A(m, k), ...) instead of A, i.e. the
     * element-wise kernel that would have produced this GEMM's input runs in its operand loader -- no pass of its own over the tensor:
     *   at_mode 1  a' = relu?( A * at_c1[k] + at_c2[k] (+ at_a2) )      BatchNorm apply (+ residual, + ReLU) of the consumer's INPUT:
     *              A = the producing convolution's raw output, c1 / c2 = scale / shift (ralf_bn_stats_from_partials), at_a2 = the
     *              residual branch or NULL (replaces ralf_bn_apply in front of a 1x1 convolution; torch: F.batch_norm + relu, + add)
     *   at_mode 2  a' = A * at_c1[k] + (at_a2 * at_c2[k] + at_c3[k])     BatchNorm backward apply on the consumer's OUTPUT GRADIENT:
     *              A = the masked gradient dz, at_a2 = the BatchNorm input x, c1..c3 from ralf_bn_bwd_stats_from_partials
     *              (replaces ralf_bn_bwd_apply in front of a 1x1 convolution's data gradient; torch: native_batch_norm_backward)
     * a' is rounded to bf16 (what the separate kernel would have stored).  The workgroups of the first column tile also store a' to
     * at_out (layout of A; may be NULL) and, in mode 1 with at_relu, the bits a' > 0 to at_mask (bit (m*K + k) & 7 of byte
     * (m*K + k) >> 3; may be NULL), so every later reader (weight gradient, residual, BatchNorm backward) finds the tensor and the
     * mask exactly as the separate kernel leaves them.  Needs bf16, a plain k-contiguous A (no gather) with lda == K on the aligned
     * interior path, K % 64 == 0, K <= 512, one batch, no split-K; at_a2 / at_out share A's layout and 16-byte alignment. */
    int at_mode, at_relu;
    const void* at_a2; const float* at_c1; const float* at_c2; const float* at_c3;
    void* at_out; unsigned char* at_mask;
    /* Threshold filter instead of an output matrix (flt_list NULL = off): element (m, n) of the product is kept when value >= flt_thresh[m].
     * Every column tile t of the launch (tiles of flt_tile columns, as ralf_gemm_filter_tile() reports for the shape) owns flt_cap slots
     * of row m's candidate list: hit number p < flt_cap of the tile is stored as flt_list[(m * T + t) * flt_cap + p] = {n as int32, value as
     * fp32} (8 bytes), and flt_count[m * T + t] = the tile's number of hits (T = ceil(N / flt_tile); every entry is written exactly once:
     * no zero-fill; a count above flt_cap means the tile lost hits and the caller must redo that row).  Nothing else is written (C is
     * ignored).  The coarse bf16 pass of the two-stage top-k search (ralf_amd/retrieval/knn.py; replaces writing and re-reading the [nq, N]
     * score matrix, 252 MB at BASELINE config 4): with a per-query lower bound of its (pool+1)-th best score as threshold the lists hold a
     * superset of the pool.  bf16, A and B k-contiguous, aligned interior path, one batch, no split-K, plain epilogue (alpha only). */
    const float* flt_thresh; int* flt_count; void* flt_list; int flt_cap, flt_thresh_ld;   /* row m's threshold = flt_thresh[m * flt_thresh_ld] (0 = 1) */
    /* LayerNorm in front of a FEW-ROW product (ln_g NULL = off): A = LayerNorm(A rows; ln_g, ln_b, ln_eps) rounded to bf16, applied by every tile's
     * wave to the 32 rows it has just loaded -- replaces ralf_layernorm_fwd + ralf_gemm for the LayerNorm -> linear pairs of a KV-cached decode
     * step (norm3 -> linear1, head LayerNorm -> vocabulary matrix: common/common.py:25-34,52-56), one launch instead of two at batch 256.
     * bf16, K == 256 (the whole row in the wave's registers), M <= 512, A and B k-contiguous, one batch, no split-K: refused otherwise. */
    const float* ln_g; const float* ln_b; float ln_eps;
    /* few_row_split = 1: a few-row product (M <= 512, bf16 NT) with K % 512 == 0 may run FOUR waves per tile, a quarter of the k range each,
     * partial tiles summed in wave order -- deterministic, but not the single MFMA chain every other path of this product writes (the fused
     * layer kernels reproduce that chain bit for bit), so the caller asks for it: the decode step's 256 x 256 x 1024 product does. */
    int few_row_split;
} RalfGemmDesc;
size_t ralf_gemm_workspace_bytes(const RalfGemmDesc* d);
int ralf_gemm(const RalfGemmDesc* d, void* workspace, size_t workspace_bytes, void* stream);
int ralf_gemm_filter_tile(const RalfGemmDesc* d);   /* column-tile width (64, 128 or 256) ralf_gemm will use for this flt_* product; <= 0 on error */
/* Which form ralf_gemm takes for a gather = 1 product that is a 3 x 3 / stride-1 / pad-1 convolution (timm Bottleneck.conv2, common/image.py:39-48; forward
 * or data gradient): 0 = the tap gather, 1 / 2 / 3 = the tile's input patch resident in LDS on 128 x 128 / 256 x 128 / 256 x 64 tiles (same results bit for
 * bit; tests and tools ask).  Only shape, dtype, layout and geometry fields of the descriptor are read.  < 0 on error. */
int ralf_gemm_patch_variant(const RalfGemmDesc* d);

/* Weight gradients of MANY linear layers in one launch (the `dW += dy^T x` products autograd issues one by one for nn.Linear /
 * nn.MultiheadAttention in_proj / out_proj, e.g. 24 per encoder stack): job j adds dy_j^T x_j into the fp32 matrix dw_j.
 *   dy bf16 [rows, n_out] (leading dimension ld_dy), x bf16 [rows, n_in] (ld_x), dw fp32 [n_out, n_in] (ld_dw)
 *   splitk <= 1: every output tile walks the whole `rows` reduction (deterministic, no workspace);
 *   splitk  > 1: the reduction is cut into slabs in `workspace` and summed in split order by a second launch.
 * rows % 64 == 0; n_out, n_in and the leading dimensions % 8 == 0; 16-byte aligned pointers. */
typedef struct RalfWgradJob {
    const void* dy; const void* x; float* dw;
    int64_t rows, ld_dy, ld_x, ld_dw;
    int n_out, n_in, splitk, pad;
    float* db;   /* may be NULL: fp32 [n_out] += column sums of dy (the bias gradient of the same layer: LinearFn.backward of
                    train/models/common/common.py's nn.Linear / in_proj), taken from the dy tiles the product reads anyway; n_out % 256 == 0 */
} RalfWgradJob;
size_t ralf_wgrad_grouped_workspace_bytes(const RalfWgradJob* jobs, int njobs);
int ralf_wgrad_grouped(const RalfWgradJob* jobs, int njobs, int dtype, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * LayerNorm (nn.LayerNorm, eps 1e-5) -- x,y [rows, cols] dtype; gamma/beta/statistics fp32.
 * bwd ACCUMULATES into dgamma/dbeta (fp32 [cols], may be NULL).   ralf_amd/csrc/norm.hip
 * ------------------------------------------------------------------------------------------- */
int ralf_layernorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                       int rows, int cols, float eps, void* stream);
/* skip (may be NULL): gradient of the residual branch of a pre-norm block, added into dx.
 * dx_drop (may be NULL): second output = ralf_dropout(dx, p_drop, seed, call_id) -- the block in front of this norm is
 * `x + dropout(f(x))`, whose backward starts by masking dx: done here while dx is in registers. */
int ralf_layernorm_bwd(int dtype, const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                       void* dx, float* dgamma, float* dbeta, const void* skip, int rows, int cols,
                       void* dx_drop, float p_drop, const int64_t* seed, uint64_t call_id, void* stream);
/* out[c] += sum_r x[r*ld + c]   (bias gradients) */
int ralf_colsum(int dtype, const void* x, int64_t ld, float* out, int rows, int cols, void* stream);
/* the same for many matrices in one launch (the bias gradients that go with ralf_wgrad_grouped); bf16, cols % 256 == 0, ld % 8 == 0 */
typedef struct RalfColsumJob {
    const void* x; float* out;
    int64_t ld;
    int rows, cols;
} RalfColsumJob;
int ralf_colsum_grouped(const RalfColsumJob* jobs, int njobs, int dtype, void* stream);

/* BatchNorm2d on NHWC viewed as [M = B*H*W, C] (timm ResNet-50 BN, momentum 0.1, eps 1e-5).
 * train: bn_stats (s1 = sum x, s2 = sum x^2; zero on entry) -> bn_finalize(training=1) -> bn_apply
 * eval : bn_finalize(training=0) uses the running statistics.
 * apply: y = relu?(x*scale + shift (+ res)).   backward: bn_bwd_reduce (s1 = sum g, s2 = sum g*xhat,
 * g = dy*(y>0) when relu) then bn_bwd_apply -> dx (and dres = g).  dgamma = s2, dbeta = s1. */
#define RALF_BN_MAX_PARTIALS 1024 /* reduction workspace = RALF_BN_MAX_PARTIALS * 2 * C floats */
int ralf_bn_stats(int dtype, const void* x, float* s1, float* s2, int64_t M, int C, float* workspace, void* stream);
int ralf_bn_finalize(const float* s1, const float* s2, const float* gamma, const float* beta, float* running_mean, float* running_var,
                     float* mean, float* rstd, float* scale, float* shift, int64_t M, int C, float eps, float momentum, int training, void* stream);
/* training-mode statistics in two launches (partial sums; reduce + finalize): replaces nn.BatchNorm2d's batch-stat path
 * incl. the running_mean / running_var / num_batches_tracked updates (torch/nn/modules/batchnorm.py semantics, momentum 0.1);
 * workspace as ralf_bn_stats; running_* and num_batches_tracked may be NULL */
int ralf_bn_batch_stats(int dtype, const void* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                        int64_t* num_batches_tracked, float* mean, float* rstd, float* scale, float* shift, int64_t M, int C,
                        float eps, float momentum, float* workspace, void* stream);
/* the same from per-64-row partial column sums written by ralf_gemm (RalfGemmDesc.colstats): partials fp32 [nrows][2][C];
 * workspace: RALF_BN_MAX_PARTIALS * 2 * C floats */
int ralf_bn_stats_from_partials(const float* partials, int nrows, const float* gamma, const float* beta, float* running_mean, float* running_var,
                                int64_t* num_batches_tracked, float* mean, float* rstd, float* scale, float* shift, int64_t M, int C,
                                float eps, float momentum, float* workspace, void* stream);
/* relu_mask (optional, M*C/8 bytes; bit i of byte j <-> element 8j+i of the [M,C] tensor) is written by bn_apply when relu and read by
 * the backward kernels INSTEAD of y (1/16 of the bytes); without it they test y > 0 (y required then). */
/* eval-mode scale / shift of many BatchNorm layers in ONE launch (the 53 of a ResNet-50): scale = gamma * rsqrt(running_var + eps),
 * shift = beta - running_mean * scale, for the colscale / bias of the folded convolution epilogue.  jobs_device: device array. */
typedef struct RalfBnFoldJob {
    const float* gamma; const float* beta; const float* mean; const float* var;
    float* scale; float* shift;
    int C; int pad_;
} RalfBnFoldJob;
int ralf_bn_fold_batched(const RalfBnFoldJob* jobs_device, int njobs, float eps, void* stream);
int ralf_bn_apply(int dtype, const void* x, const float* scale, const float* shift, const void* res, void* y, uint8_t* relu_mask,
                  int64_t M, int C, int relu, void* stream);
int ralf_bn_bwd_reduce(int dtype, const void* x, const void* dy, const void* y, const uint8_t* relu_mask, const float* mean, const float* rstd,
                       float* s1, float* s2, int64_t M, int C, int relu, float* workspace, void* stream);
/* the reductions from the partial rows a data-gradient GEMM wrote (RalfGemmDesc.bnb_part, nrows = ceil(M/64)):
 * s1[c] += sum dz, s2[c] += rstd[c] * sum dz*(x - mean) = sum dz*xhat; workspace: 128*2*C floats.  Then ralf_bn_bwd_apply with relu = 0
 * (dz is already masked; the gradient of the residual branch is dz itself). */
int ralf_bn_bwd_stats_from_partials(const float* partials, int nrows, const float* rstd, float* s1, float* s2, int C, float* workspace,
                                    const float* gamma, const float* mean, int64_t M, float* coef, void* stream);
/* coef (fp32 [3][C], may be NULL; needs gamma, mean, M and s1 / s2 zero on entry): dx = coef[0][c] * dz + coef[1][c] * x + coef[2][c], the whole
 * backward apply as one per-channel affine map -- applied by ralf_bn_bwd_apply_affine or inside the operand loader of the data-gradient GEMM
 * that consumes dx (RalfGemmDesc.at_mode 2) */
int ralf_bn_bwd_apply_affine(int dtype, const void* dz, const void* x, const float* c1, const float* c2, const float* c3, void* dx, int64_t M, int C, void* stream);
int ralf_bn_bwd_apply(int dtype, const void* x, const void* dy, const void* y, const uint8_t* relu_mask, const float* mean, const float* rstd,
                      const float* gamma, const float* s1, const float* s2, void* dx, void* dres, int64_t M, int C, int relu, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Gather / pointwise kernels (ralf_amd/csrc/pointwise.hip)
 * ------------------------------------------------------------------------------------------- */
/* out[r,:] = W[idx[r],:]*scale + pe[r % S,:]   (nn.Embedding -> PositionalEncoding1d, common/common.py:97-98) */
int ralf_embed_fwd(int dtype, const int64_t* idx, const float* W, const float* pe, void* out, int64_t rows, int S, int d, float scale, void* stream);
int ralf_embed_bwd(int dtype, const int64_t* idx, const void* dy, float* dW, int64_t rows, int d, float scale, void* stream);
/* y = x*keep/(1-p) (+ res), mask = f(seed[0], call_id, element index): the same call on dy (res = NULL) is the
 * backward; with p == 0 it is a plain residual add */
int ralf_dropout(int dtype, const void* x, const void* res, void* y, int64_t n, float p, const int64_t* seed, uint64_t call_id, void* stream);
/* nn.CrossEntropyLoss(label_smoothing, ignore_index) on fp32 logits [rows,V]: cnt_loss = {#valid, mean loss};
 * dlogits (dtype, may be NULL) = d loss / d logits */
int ralf_xent_fwd_bwd(int dtype, const float* logits, const int64_t* target, void* dlogits, float* cnt_loss, int64_t rows, int V,
                      int ignore_index, float label_smoothing, void* stream);
/* sequence concatenation of up to 4 sources [B, rows_i, d] into out [B, sum rows_i, d], each with an optional learned scalar added
 * (torch.cat + the nn.Embedding(2, 1) flags of retrieval_augmented_autoreg.py:963-994,1022-1028); backward = 1: `out` is the
 * gradient of the concatenation, src_i receive its pieces (contiguous) and dscalar_i[0] += the piece's sum. */
int ralf_concat_rows(int dtype, int backward, int nsrc, const void* const* src, const int* rows, const float* const* scalar, float* const* dscalar,
                     void* out, int B, int d, void* stream);
/* y = dropout_p(x * scale + pe[r % S, :]) on [rows, d]: PositionalEncoding1d (common/positional_encoding.py:92-107) of the K retrieved
 * features; pe fp32 [S, d] or NULL (the same call on a gradient with pe = NULL is the backward) */
int ralf_scale_pe_dropout(int dtype, const void* x, const float* pe, void* y, int64_t rows, int S, int d, float scale, float p,
                          const int64_t* seed, uint64_t call_id, void* stream);
int ralf_add_scalar(int dtype, const void* x, const float* s, void* y, int64_t rows, int cols, int64_t ldx, int64_t ldy, void* stream);
int ralf_sum_all(int dtype, const void* x, float* out, int64_t rows, int cols, int64_t ldx, void* stream);
/* *counter += inc on the device (int32 or int64): the optimizer's step count / the dropout seed of the next step, advanced inside a
 * captured graph (torch: `state["step"] += 1`, train/train.py:449-454; the generator state) */
int ralf_counter_add(void* counter, int is_int64, int64_t inc, void* stream);
/* input rows of the frozen layout encoder (fid/model.py:90-103: stack of (cx, cy, w, h)) and the key-padding mask of its sequence:
 * bbox [R*N, 8] (dtype; columns 4..7 zero), kpm uint8 [R, N+1] = (0, !mask[r, 0], ..., !mask[r, N-1]); cx..h fp32 [R, N], mask uint8/bool [R, N] */
int ralf_layout_pack(int dtype, const float* cx, const float* cy, const float* w, const float* h, const uint8_t* mask, void* bbox, uint8_t* kpm,
                     int64_t R, int N, void* stream);
/* y[i] (fp32) = x[i] (dtype) * s[0], s on the device: the chain rule through a scalar loss (`dlogits * grad_output` in autograd's
 * backward of nn.CrossEntropyLoss on fp32 logits, retrieval_augmented_autoreg.py:209-216) without a host read of the factor */
int ralf_scale_dev(int dtype, const void* x, const float* s, float* y, int64_t n, void* stream);
/* p[0 .. nbytes) = 0 (16-byte aligned, nbytes % 16 == 0): optimizer.zero_grad() on the flat gradient buffer (train/train.py:444) */
int ralf_zero(void* p, int64_t nbytes, void* stream);
int ralf_copy2d(int src_dtype, int dst_dtype, const void* src, void* dst, int64_t rows, int cols, int64_t lds, int64_t ldd, int accumulate, void* stream);
int ralf_permute4(int src_dtype, int dst_dtype, const void* in, void* out, int d0, int d1, int d2, int d3, int64_t s0, int64_t s1, int64_t s2, int64_t s3,
                  int valid3, void* stream);
/* the same for a table of jobs in ONE launch (per-step re-layout of all 3x3 convolution weights: Runtime.lp shadows).
 * jobs_device: njobs RalfPermuteJob records in DEVICE memory; job j runs on workgroups [first_block[j], first_block[j+1])
 * (ascending, first_block[0] = 0), total_blocks = grid size. */
typedef struct RalfPermuteJob {
    const void* in; void* out;
    int64_t s0, s1, s2, s3;
    int d0, d1, d2, d3;
    int valid3, src_dtype, dst_dtype, first_block;
} RalfPermuteJob;
int ralf_permute4_batched(const RalfPermuteJob* jobs_device, int njobs, int total_blocks, void* stream);
/* Both GEMM operand layouts of k x k convolution weights from one read of the fp32 OIHW masters, many weights per launch (every
 * optimizer step re-derives them): ohwi [Co][KK][Cip] (input channels zero-padded to Cip) and ikwo [Ci][KK][Co], KK = kh*kw <= 49,
 * dst_dtype RALF_F32 / RALF_BF16.  Job j owns workgroups [first_block[j], first_block[j+1]); jobs_device: device array. */
typedef struct RalfConvRelayoutJob {
    const void* w; void* ohwi; void* ikwo;
    int Co, Ci, KK, Cip, dst_dtype, first_block;
} RalfConvRelayoutJob;
int ralf_conv_relayout_batched(const RalfConvRelayoutJob* jobs_device, int njobs, int total_blocks, void* stream);
/* ResNet stem max-pool 3x3/s2/p1 (NHWC) with saved arg-max; FPN nearest up-sampling fused with the lateral add */
/* The ResNet stem convolution (7x7, stride 2, pad 3, 4 input channels stored as 8, 64 output channels) in direct form (ralf_amd/csrc/stem.hip;
 * timm resnet50 conv1 with the saliency channel, common/image.py:39-48,70-77; torch: F.conv2d): x [B,IH,IW,8] bf16 NHWC (channels 4..7 zero),
 * w [64][7][7][8] bf16 -> y [B,OH,OW,64] bf16 with OH = (IH-1)/2+1, OW = (IW-1)/2+1.  part (may be NULL): fp32 [B*OH*ceil(OW/128)][2][64], per
 * output-row tile the channel sums and sums of squares of y as stored: BatchNorm partial statistics for ralf_bn_stats_from_partials. */
int ralf_stem7x7_fwd(const void* x, const void* w, void* y, float* part, int B, int IH, int IW, void* stream);
/* its weight gradient in the same form (the weight half of torch's conv2d backward): dW [64][4][7][7] fp32 OIHW (= or +=) from x [B,IH,IW,8] and
 * dy [B,OH,OW,64] bf16; one persistent workgroup per CU keeps the 64 x 7 x 64 block in registers over its output-row tiles, a second kernel sums the
 * workgroups' blocks in order (deterministic) */
size_t ralf_stem7x7_wgrad_workspace_bytes(int B, int IH, int IW);
int ralf_stem7x7_wgrad(const void* x, const void* dy, float* dW, int B, int IH, int IW, int accumulate, void* workspace, size_t workspace_bytes, void* stream);
/* 1x1 convolution with FEW (K = 64 or 128) input channels on NHWC bf16 rows (ralf_amd/csrc/conv1x1_k64.hip): y [M, N] = x [M, K] w [N, K]^T, M % 64 == 0,
 * N = 64 * 2^k <= 2048.  Replaces torch's conv2d for timm Bottleneck.conv3 / downsample.0 / the first conv1 of ResNet-50's layer1
 * (image2layout/train/models/common/image.py:39-48) -- one k-tile of matrix work per output tile in front of a store four times the operand's
 * size, where ralf_gemm's tiled kernel spends the tile on its latency chain.  Same output bits as ralf_gemm.
 *   colstats (may be NULL): fp32 [M / 64, 2, N], per 64-row block the column sums / sums of squares of y as stored (= RalfGemmDesc.colstats)
 *   scale, shift (NULL together = off; not with colstats): y = act(acc * scale[n] + shift[n]) (+ res [M, N]); relu 0 / 1 / 2 (2 = after the
 *   residual: RALF_ACT_RELU_POST) -- the eval-mode BatchNorm of the inference backbone (RalfGemmDesc.colscale / bias / res). */
int ralf_conv1x1_k64(const void* x, const void* w, void* y, float* colstats, const float* scale, const float* shift, const void* res,
                     int relu, int64_t M, int N, int K, void* stream);

/* Weight gradient of a 3x3 / pad 1 convolution of stride s = 1 or 2, DIRECT form (ralf_amd/csrc/conv_wgrad.hip): dW[co][ci][kh][kw] (fp32, OIHW;
 * = or +=) = sum over pixels of dy[b,oy,ox,co] * x[b,s oy+kh-1,s ox+kw-1,ci]; dy [B,H,W,Co] and x [B,IH,IW,Ci] NHWC bf16 (H, W: the OUTPUT grid).  A workgroup keeps a 64 x 64 x 9
 * block of dW in registers and stages each 64-pixel tile of dy and the halo patch of x once for all nine taps (the implicit-GEMM form,
 * ralf_gemm gather = 2, moves every input pixel nine times).  W in {8, 16, 32, 64} (stride 2: <= 32), H % (64 / W) == 0, Ci % 64 == 0, Co % 64 == 0.  Replaces
 * the weight half of torch's conv2d backward for the bottlenecks' 3x3 convolutions (common/image.py:39-48).  Deterministic (splits summed
 * in order). */
size_t ralf_conv3x3_wgrad_workspace_bytes(int B, int H, int W, int Ci, int Co);
int ralf_conv3x3_wgrad(const void* dy, const void* x, float* dW, int B, int H, int W, int IH, int IW, int stride, int Ci, int Co, int accumulate,
                       void* workspace, size_t workspace_bytes, void* stream);
/* The stem's BatchNorm (batch statistics) + ReLU + 3x3 / stride 2 / pad 1 max-pool in one pass over the convolution output y (NHWC), and its
 * backward in two (timm resnet50 conv1 -> bn1 -> act1 -> maxpool, common/image.py:39-48,66-67; torch: F.batch_norm, relu, F.max_pool2d and their
 * autograd backward).  scale / shift from ralf_bn_stats_from_partials.  fwd: out [B,OH,OW,C], arg int8 = position 0..8 of the maximum;
 * pooled values and arg equal ralf_bn_apply -> ralf_maxpool3x3s2_fwd bit for bit, the normalised tensor is never stored.
 * bwd_reduce: part[nblk][2][C] = per-workgroup column sums of dz and dz * (y - mean), dz = the gradient reaching relu(BN(y)) through the
 * pooling (rounded to dtype, zero where the ReLU was inactive) -> ralf_bn_bwd_stats_from_partials(part, nblk, ..., coef);
 * bwd_apply: dy = c1 * dz + c2 * y + c3 with dz recomputed. */
int ralf_bn_relu_maxpool_fwd(int dtype, const void* y, const float* scale, const float* shift, void* out, int8_t* arg, int B, int H, int W, int C, void* stream);
int ralf_bn_relu_maxpool_bwd_reduce(int dtype, const void* dpool, const int8_t* arg, const void* y, const float* scale, const float* shift, const float* mean,
                                    float* part, int nblk, int B, int H, int W, int C, void* stream);
int ralf_bn_relu_maxpool_bwd_apply(int dtype, const void* dpool, const int8_t* arg, const void* y, const float* scale, const float* shift, const float* c1,
                                   const float* c2, const float* c3, void* dy, int B, int H, int W, int C, void* stream);
int ralf_maxpool3x3s2_fwd(int dtype, const void* x, void* y, int8_t* arg, int B, int H, int W, int C, void* stream);
int ralf_maxpool3x3s2_bwd(int dtype, const void* dy, const int8_t* arg, void* dx, int B, int H, int W, int C, void* stream);
int ralf_upsample_nearest_add(int dtype, const void* src, const void* lateral, void* up, int64_t ld_up, void* sum, int B, int IH, int IW, int OH, int OW, int C, void* stream);
int ralf_upsample_nearest_bwd(int dtype, const void* g_up, int64_t ld_up, const void* g_sum, void* dsrc, int B, int IH, int IW, int OH, int OW, int C, void* stream);

/* decode step tail (retrieval_augmented_autoreg.py:282-296; helpers/sampling.py:18-71): vocabulary mask of the
 * position (tokenizer.token_mask[i]), forced token from DECODE_SPACE_RESTRICTION (-1 = free), then the choice of sampling.py:18-71:
 * mode 0 argmax (`deterministic`), 1 `top_k`, 2 `top_p` (nucleus: candidates whose inclusive cumulative probability in descending order
 * exceeds top_p are dropped, the first stays), 3 `random` (softmax(x / T)), 4 `gumbel`; one multinomial draw with a counter-based
 * generator for modes 1-4.  logits fp32 [B,V], V <= 1024. */
int ralf_mask_sample(const float* logits, const uint8_t* allowed, const int64_t* forced, int mode, int top_k, float temperature,
                     const int64_t* seed, uint64_t call_id, int64_t* out, int B, int V, float top_p, void* stream);
/* the same, and the chosen token also lands where the NEXT decode step reads it (no slice / compare / concatenate kernels between
 * steps): seq_out[b * seq_ld] = token (column of the [B, max_len+1] sequence buffer, may be NULL), pad_flag_out[b * flag_ld] =
 * (token == pad_id) (column of the uint8 key-padding mask of the self-attention, may be NULL) */
int ralf_mask_sample_step(const float* logits, const uint8_t* allowed, const int64_t* forced, int mode, int top_k, float temperature,
                          const int64_t* seed, uint64_t call_id, int64_t* out, int64_t* seq_out, int64_t seq_ld, uint8_t* pad_flag_out,
                          int64_t flag_ld, int64_t pad_id, int B, int V, float top_p, int row0, void* stream);
/* row0: number of the first row in the WHOLE batch when a batch is decoded in slices (the counter-based draws are indexed by it) */

/* ---------------------------------------------------------------------------------------------
 * Fused attention (ralf_amd/csrc/attention.hip): O = dropout(softmax(scale*Q K^T + mask)) V.
 * Operands are [B, S, H*dh]-style views: element (b, s, h, c) at base + b*bs + s*rs + h*dh + c.
 * kpm: uint8 [B, Sk], 1 = padded key (masked).  lse/delta: fp32 [B, H, Sq] (saved for backward).
 * ------------------------------------------------------------------------------------------- */
typedef struct RalfAttnDesc {
    const void* q; const void* k; const void* v; void* o;
    const void* dout; void* dq; void* dk; void* dv;
    float* lse; float* delta;
    const uint8_t* kpm; const int64_t* seed;
    int64_t q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs;
    int64_t do_bs, do_rs, dq_bs, dq_rs, dk_bs, dk_rs, dv_bs, dv_rs;
    uint64_t call_id;
    int B, H, Sq, Sk, dh, dtype, causal;
    float scale, p_drop;
    int64_t kpm_bs; /* row stride of kpm in bytes; 0 = Sk (a decode loop keeps ONE [B, max_len] mask and grows Sk) */
} RalfAttnDesc;
int ralf_attention_fwd(const RalfAttnDesc* d, void* stream);
int ralf_attention_bwd(const RalfAttnDesc* d, void* stream);
/* One attention block of a KV-cached decode step with its projections inside (ralf_amd/csrc/attention_mfma.hip): for row b
 *   h = LayerNorm(x[b]);  q = h Wq^T + bq;  self_: (k, v) = h Wk^T + bk, h Wv^T + bv are written to cache row Sk first;
 *   o[b] = softmax(q K^T * scale, keys masked by kpm) V   per head, over the Sk cached rows (+ the new one when self_).
 * Replaces ralf_layernorm_fwd + 2 x ralf_gemm + ralf_attention_fwd of the reference's per-token decoder call
 * (retrieval_augmented_autoreg.py:274-279, nn.TransformerDecoderLayer norm_first) at batch 256: bf16, d = 256, H = 8.
 *   x [B, d] bf16 (row stride x_rs)   W: nn.MultiheadAttention.in_proj_weight as bf16 [3d, d] (rows 0..d-1 = q; self_ also k, v)
 *   bias fp32 [3d]   kv: bf16 cache [B, rows, 2d] (k at column 0, v at column d; strides in elements)   kpm uint8 [B, kpm_bs] or NULL */
typedef struct RalfDecodeAttnDesc {
    const void* x; const float* ln_g; const float* ln_b;
    const void* W; const float* bias;
    void* kv; const uint8_t* kpm; void* o;
    int64_t x_rs, kv_bs, kv_rs, kpm_bs, o_rs;
    int B, H, d, Sk, self_;
    float scale, eps;
    int64_t kv_hs, kv_vo; /* cross-attention only; 0, 0 = the [B, rows, 2d] cache above.  Head-pair-major cache (e.g. [B, 2, H/2, rows, 64], written by
                           * a batched ralf_gemm over the 8 column slices of the k | v projection): the keys of head pair hp of element b start at
                           * kv + b*kv_bs + hp*kv_hs, rows kv_rs (= 64) apart, its values kv_vo elements behind them -- a workgroup streams two
                           * contiguous blocks instead of 128-byte pieces 1 KB apart */
    const int32_t* pos;   /* self_ only, may be NULL: int32 [B] on the device, the number of cached rows PER ELEMENT (then Sk = their upper bound):
                           * the samples of a lock-step decode that rewind their prefixes independently (sample_relation's back-tracking,
                           * retrieval_augmented_autoreg.py:432-460) */
} RalfDecodeAttnDesc;
int ralf_decode_attn(const RalfDecodeAttnDesc* d, void* stream);
/* most keys (cached rows, + the new one when self_) ralf_decode_attn accepts: its scores live in LDS.  Callers with longer memories
 * (2*h*w + K + Lc rows: e.g. 512x512 canvases) use ralf_layernorm_fwd + ralf_gemm + ralf_attention_fwd instead. */
int ralf_decode_attn_max_keys(void);

/* One KV-cached decode step of the WHOLE decoder stack in ONE launch, a workgroup per sample (ralf_amd/csrc/decode_token.hip): token ids [B] at
 * positions pos (or pos_vec[b]) -> fp32 logits [B, V].  Replaces the per-token decoder call of the reference's sampling loop
 * (retrieval_augmented_autoreg.py:274-279; common/common.py:84-135: embedding, nlayers x nn.TransformerDecoderLayer(norm_first), head) and the
 * ~45 launches per token it took through ralf_embed_fwd / ralf_decode_attn / ralf_gemm.  bf16 weights row-major [n_out, n_in] (the
 * nn.Linear / in_proj layouts), fp32 biases and LayerNorm parameters, d = 256, 8 heads, feed-forward 1024.
 *   self_kv  bf16 [B, L, 2d] per layer: rows < pos are read, row pos is written (k at column 0, v at column d)
 *   cross_kv bf16 [B, 8, M, 64] per layer: slice 4 * kv + hp = keys (kv = 0) / values (1) of head pair hp (nn.decoder_init_cache)
 *   kpm uint8 [B, kpm_bs] or NULL: padded PREFIX tokens of the self-attention (tgt_key_padding_mask), columns 0 .. pos
 *   emb fp32 [vocab, d], pe fp32 [positions, d], emb_scale = sqrt(d); lnh_*, w_head [V, d]: the head (LayerNorm + Linear without bias) */
#define RALF_DECODE_TOKEN_MAX_LAYERS 8
typedef struct RalfDecodeTokenLayer {
    const void* w_qkv; const float* b_qkv; const float* ln1_g; const float* ln1_b;
    const void* w_o1; const float* b_o1;
    const float* ln2_g; const float* ln2_b; const void* w_q2; const float* b_q2;
    const void* w_o2; const float* b_o2;
    const float* ln3_g; const float* ln3_b; const void* w_f1; const float* b_f1; const void* w_f2; const float* b_f2;
    void* self_kv; const void* cross_kv;
} RalfDecodeTokenLayer;
typedef struct RalfDecodeTokenDesc {
    const int64_t* tok; const int32_t* pos_vec; const uint8_t* kpm;
    const float* emb; const float* pe; const float* lnh_g; const float* lnh_b; const void* w_head; float* logits;
    int64_t kpm_bs;
    int B, L, M, V, nlayers, pos;
    float emb_scale, eps;
    RalfDecodeTokenLayer layer[RALF_DECODE_TOKEN_MAX_LAYERS];
    /* optional (ABI 25; s_out != NULL): the token choice of ralf_mask_sample_step on this step's logits in the SAME launch -- one wave per sample runs the
     * same function on the same values (helpers/sampling.py: deterministic / top_k / top_p / random / gumbel; retrieval_augmented_autoreg.py:280-300).
     * Arguments as ralf_mask_sample_step's: allowed uint8 [V], forced int64 [B] or NULL, seed, call id, out int64 [B] (may be `tok`: a workgroup
     * reads its element at the start and writes it at the end), the sequence / pad-flag columns with their strides, pad id, row0.  V <= 1024. */
    const uint8_t* s_allowed; const int64_t* s_forced; const int64_t* s_seed; int64_t* s_out; int64_t* s_seq_out; uint8_t* s_flag_out;
    int64_t s_seq_ld, s_flag_ld, s_pad_id;
    uint64_t s_call;
    int s_mode, s_top_k, s_row0;
    float s_temperature, s_top_p;
} RalfDecodeTokenDesc;
int ralf_decode_token(const RalfDecodeTokenDesc* d, void* stream);
int ralf_decode_token_limits(int* max_self_rows, int* max_memory_rows);   /* most cached positions (L) and memory rows (M) ralf_decode_token takes */

/* Pre-norm transformer layers (forward, training) of SHORT sequences (ralf_amd/csrc/tlayer.hip): a workgroup per sample keeps its
 * S <= RALF_TLAYER_MAX_ROWS rows in LDS from a LayerNorm to the end of the block chain.  bf16, d = 256, 8 heads, ff = 1024:
 *   x1  = x  + drop(attn(LN1(x))      Wo^T  + bo)            self-attention (causal and / or key-padding mask)        parts 0, 1
 *   q   = LN2(x1) Wq^T + bq                                   the cross-attention's queries                            part 1
 *   x2  = x1 + drop(o2 Wo2^T + bo2)                           o2 = ralf_attention_fwd(q, memory k | v), run by the caller between parts 1 and 2   part 2
 *   out = r  + drop(W2 drop(relu(W1 LN3(r) + b1)) + b2)      r = x2 (part 2) or x1 (part 0)                          parts 0, 2
 * part 0 = a whole nn.TransformerEncoderLayer in one launch; parts 1 + ralf_attention_fwd + 2 = nn.TransformerDecoderLayer in three
 * (norm_first; image2layout/train/models/common/common.py:25-34,84-135,216-226), instead of the 7 (12) launches of ralf_layernorm_fwd /
 * ralf_gemm / ralf_attention_fwd, with the SAME arithmetic and dropout masks (call ids = the RalfGemmDesc.call_id / RalfAttnDesc.call_id
 * of those launches) and every tensor their backward passes read written on the way -- the backward is the unfused one.
 *   weights: bf16 in FRAGMENT ORDER (ralf_tlayer_pack of the row-major [n_out, n_in] matrix; w_q = the first 256 rows of the cross-attention's
 *   in_proj_weight); biases and LayerNorm parameters fp32; activations bf16 [B*S, width]; statistics fp32 [B*S]; lse fp32 [B, 8, S];
 *   kpm uint8 [B, kpm_bs] or NULL; seed int64[1] on the device (needed when a dropout probability is > 0)
 * Parts 2, 3 and 4 also run on 64-row STRIPS of any [rows, 256] tensor (B = rows / S "samples"): the long-sequence encoder layers (256 tokens
 * per sample: the attention keeps ralf_attention_fwd) are part 4 (h1 = LN1(x), qkv = h1 Win^T + bin), ralf_attention_fwd, part 2 (x1 = the
 * layer input, o2 = the attention output) -- or part 3 (out = x + drop(W2 drop(relu(W1 LN3(x) + b1)) + b2)) behind a separate out-projection.
 * Part 2 is also the tail of a KV-cached DECODE step (retrieval_augmented_autoreg.py:274-279: one new token per batch element): the "samples"
 * are then strips of S <= 32 batch rows (one 32-row MFMA block per workgroup), dropout off, and h3 / mean3 / rstd3 / hid may be NULL (nothing is
 * kept for a backward pass; x2 is still written -- the kernel reads it back for the last residual add). */
#define RALF_TLAYER_MAX_ROWS 64
typedef struct RalfTLayerDesc {
    const void* x;
    const float* ln1_g; const float* ln1_b; const void* w_in; const float* b_in; const void* w_o; const float* b_o;
    const uint8_t* kpm;
    const float* ln2_g; const float* ln2_b; const void* w_q; const float* b_q; const void* o2; const void* w_o2; const float* b_o2;
    const float* ln3_g; const float* ln3_b; const void* w1; const float* b1; const void* w2; const float* b2;
    void* h1; float* mean1; float* rstd1; void* qkv; void* o1; float* lse1; void* x1;   /* x1: output of parts 0, 1; input of part 2 */
    void* h2; float* mean2; float* rstd2; void* q; void* x2;
    void* h3; float* mean3; float* rstd3; void* hid; void* out;
    const int64_t* seed;
    uint64_t call_attn1, call_out1, call_out2, call_ffn1, call_ffn2;
    int64_t kpm_bs; /* row stride of kpm in bytes */
    int B, S, causal, part;
    float scale, p_attn, p_res, eps;
    /* part 3 only: act = RALF_ACT_GELU (erf form) instead of ReLU, z = the pre-activation W1 LN(x) + b1 (bf16 [rows, 1024], what the GELU
     * gradient reads; NULL with ReLU), no_res = 1: out = W2 act(..) + b2 without the residual add -- LN -> Linear -> GELU -> Linear, the
     * reference's FeedForward (image2layout/train/models/common/attention.py:15-30) */
    void* z; int act, no_res;
} RalfTLayerDesc;
int ralf_tlayer_fwd(const RalfTLayerDesc* d, void* stream);
/* Backward of the long-sequence layers' tail (ralf_tlayer_fwd part 2 on 64-row strips), data gradients only -- the launches
 * ralf_gemm (dz, with the ReLU mask) -> ralf_gemm (dh) -> ralf_layernorm_bwd -> ralf_gemm (d o) of functional.FFNFn / LayerNormSkipFn /
 * LinearFn.backward in one, with the same arithmetic; the weight / bias gradients are taken afterwards from the tensors it writes
 * (dz, g_m) by the grouped weight-gradient launches, as before.  B strips of S <= 64 rows:
 *   dz  = (dy_m W2) o [hid > 0] / (1 - p)        dy_m = the gradient of the layer output masked by the ffn2 dropout (dy itself when p = 0)
 *   dh  = dz W1                                   (stage 1 stops here and writes dh to g)
 *   g   = LayerNorm3 backward(dh; x2, mean3, rstd3, gamma) + dy     the gradient of r = x + drop(o Wo^T + bo); dgamma / dbeta += (atomics)
 *   g_m = g masked by the out-projection dropout (p, call_out);  d_o = g_m Wo
 * w2t / w1t / wot: ralf_tlayer_pack of the TRANSPOSED weights (RalfPackJob.transpose).
 * stage 4 = the backward of part 4 (LayerNorm 1 + q | k | v projection): dy_m = dqkv [rows, 768], w1t = Win^T, dh = dqkv Win, then the same
 * LayerNorm backward on (x2, mean3, rstd3, ln3_g) := (x, mean1, rstd1, gamma 1) with dy (may be NULL) as the skip gradient: g = dx,
 * g_m = dx masked by (p, call_out) -- the dropout of the block that produced x.  nk = the width of dy_m in 256-column chunks: 3 for dqkv, 1 for
 * the gradient of a 256 -> 256 projection (the decoder's cross-attention queries); with wot (and d_o) the out-projection's data gradient
 * d_o = g_m Wo follows as in stage 3 (decoder: q-projection + LayerNorm 2 backward + the self-attention out-projection's data gradient).
 * stage 5 = the backward of part 3 with GELU (FeedForward): hid = the pre-activation z, dz = (dy_m W2) o gelu'(z), dh = dz W1, the LayerNorm
 * backward (dy NULL: no skip gradient), no product behind it. */
typedef struct RalfTLayerBwdDesc {
    const void* dy_m; const void* dy; const void* hid; const void* x2;
    const float* mean3; const float* rstd3; const float* ln3_g;
    const void* w2t; const void* w1t; const void* wot;
    void* dz; void* g; void* g_m; void* d_o;
    float* dgamma; float* dbeta;
    const int64_t* seed; uint64_t call_out;
    int B, S, stage, nk;
    float p, pad2_;
} RalfTLayerBwdDesc;
int ralf_tlayer_bwd(const RalfTLayerBwdDesc* d, void* stream);
/* weights -> the fragment order ralf_tlayer_fwd streams: for the 32-row tile t and the 16-wide k-slice i of src [N][K] (row stride ld
 * elements), the 64 lanes' MFMA operands (lane (r, half) = src[32 t + r][16 i + 8 half .. + 7]) become 1 KiB of consecutive memory at
 * dst + ((t * K/16 + i) * 64 + lane) * 8 elements.  N % 32 == 0, K % 16 == 0; up to 96 matrices per launch (jobs is a HOST array). */
typedef struct RalfPackJob { const void* src; void* dst; int64_t ld; int N, K; int transpose, pad_; } RalfPackJob;   /* transpose: the packed
    matrix is src^T, i.e. element [n][k] = src[k * ld + n] (src [K][N] row-major): the data-gradient products of ralf_tlayer_bwd */
int ralf_tlayer_pack(const RalfPackJob* jobs, int njobs, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Optimizer (ralf_amd/csrc/optim.hip): clip_grad_norm_ + AdamW on flat fp32 buffers
 * ------------------------------------------------------------------------------------------- */
int ralf_sumsq(const float* g, int64_t n, float* out, void* stream);                 /* out[0] += sum g^2 (fp32 atomics: order not fixed) */
int ralf_clip_coef(const float* sumsq, float max_norm, float* coef, float* norm_out, void* stream);
/* the deterministic pair the train step uses: RALF_SUMSQ_PARTS per-workgroup partial sums (fixed element assignment), summed in a fixed
 * order -- the same gradient buffer gives the same clip coefficient bit for bit, on every run and on every data-parallel rank */
#define RALF_SUMSQ_PARTS 1024
int ralf_sumsq_partials(const float* g, int64_t n, float* partials, void* stream);
int ralf_clip_coef_partials(const float* partials, float max_norm, float* coef, float* norm_out, void* stream);
/* step_dev (int32[1] on the device, may be NULL) overrides `step` for the bias corrections (graph replay);
 * lr_scale (fp32[1] on the device, may be NULL) multiplies lr: the scheduler's factor (train/schedulers/multi_step_lr.py) without re-capturing */
int ralf_adamw(float* p, const float* g, float* m, float* v, void* shadow_bf16, int64_t n, float lr, float beta1, float beta2, float eps,
               float weight_decay, int step, const float* coef, const int* step_dev, const float* lr_scale, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Streams owned by the library (ralf_amd/csrc/optim.hip).  The host runtime runs its graph branches, weight-gradient
 * queue and captures on THESE, never on streams of the framework's shared pool: the pool also hands its streams to the
 * collective library, and an event of an in-flight collective that sits on a stream which later starts capturing makes the
 * collective's completion poll fail (hipErrorCapturedEvent) and takes the process down.  Non-blocking w.r.t. the null stream.
 * ------------------------------------------------------------------------------------------- */
int ralf_stream_create(void** out_stream);
/* the same with the device's highest (high != 0) or lowest stream priority: a prioritised stream gets a hardware queue of its own, so a
 * host-to-device copy of the NEXT batch issued on it is not parked behind the kernels of the running step (a plain stream shares one of
 * the 4 hardware queues with the step's graph branches: the loop's `.to(rank)`, train/train.py:434, waited 17-22 ms per iteration) */
int ralf_stream_create_priority(void** out_stream, int high);
int ralf_stream_destroy(void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RALF_HIP_H */
