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

#ifdef __cplusplus
}
#endif
#endif /* RALF_HIP_H */
