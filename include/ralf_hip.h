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

#define RALF_ABI_VERSION 1
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

/* ---------------------------------------------------------------------------------------------
 * element types of activation / weight buffers (accumulation is always fp32)
 * ------------------------------------------------------------------------------------------- */
#define RALF_F32 0  /* parity mode: exact-fp32 matrix cores (v_mfma_f32_32x32x2_f32)            */
#define RALF_BF16 1 /* throughput mode: bf16 operands (v_mfma_f32_32x32x16_bf16), fp32 accum    */

#define RALF_ACT_NONE 0
#define RALF_ACT_RELU 1
#define RALF_ACT_GELU 2          /* exact erf GELU (nn.GELU default, common/attention.py:23)      */
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
 *   epilogue order: *alpha, +bias[n] (fp32), C2 = v (optional pre-activation copy), act,
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
} RalfGemmDesc;
size_t ralf_gemm_workspace_bytes(const RalfGemmDesc* d);
int ralf_gemm(const RalfGemmDesc* d, void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RALF_HIP_H */
