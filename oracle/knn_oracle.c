/* ORACLE (test infrastructure, NOT product code): exact inner-product top-k scan on the CPU.
 *
 * Restates what the reference obtains from faiss-cpu ^1.7.4 (pyproject.toml:29; absent from
 * /root/reference and from this image) through HF-datasets:
 *   db_dataset.add_faiss_index_from_external_arrays(vectors, metric_type=faiss.METRIC_INNER_PRODUCT)
 *       image2layout/train/models/retrieval/retriever.py:79-84   -> faiss.IndexFlat(d, IP)
 *   db_dataset.get_nearest_examples("search_feat", query, k=top_k+1)
 *       image2layout/train/models/retrieval/retriever.py:200-202 -> IndexFlat.search(1 x D, k)
 * faiss IndexFlat(IP) = brute force: score(q,n) = <q, x_n> in fp32, the k largest scores returned
 * in descending order.  faiss's own summation order and tie order are not observable here
 * (PARITY UNPINNED against faiss itself); this oracle FIXES them:
 *   score = ((0 + x[0]q[0]) + x[1]q[1]) + ...   one fused multiply-add per dimension, ascending d
 *           (bit-identical to a chain of v_mfma_f32_*_f32 K-steps on gfx950)
 *   order = (score descending, database index ascending)
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define QB 16 /* queries per register block */

int knn_oracle_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* scores[q*N + n] = fmaf-chain inner product.  X [N,D] row-major, Q [nq,D] row-major. */
void knn_oracle_scores(const float* X, int64_t N, int D, const float* Q, int nq, float* scores) {
    int nqp = (nq + QB - 1) / QB * QB;
    float* Qt = (float*)calloc((size_t)D * nqp, sizeof(float)); /* [D][nqp] */
    for (int q = 0; q < nq; ++q)
        for (int d = 0; d < D; ++d) Qt[(size_t)d * nqp + q] = Q[(size_t)q * D + d];
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; ++n) {
        const float* x = X + (size_t)n * D;
        for (int q0 = 0; q0 < nqp; q0 += QB) {
            float acc[QB];
            for (int j = 0; j < QB; ++j) acc[j] = 0.0f;
            for (int d = 0; d < D; ++d) {
                const float xv = x[d];
                const float* qrow = Qt + (size_t)d * nqp + q0;
                for (int j = 0; j < QB; ++j) acc[j] = fmaf(xv, qrow[j], acc[j]);
            }
            for (int j = 0; j < QB && q0 + j < nq; ++j) scores[(size_t)(q0 + j) * N + n] = acc[j];
        }
    }
    free(Qt);
}

/* top-k of one score row under (score desc, idx asc); k <= N assumed by caller (else padded -1/-inf) */
static void select_topk(const float* s, int64_t N, int k, int64_t* idx, float* val) {
    int m = 0; /* filled */
    for (int64_t n = 0; n < N; ++n) {
        const float v = s[n];
        if (m == k && !(v > val[k - 1])) continue; /* equal score, larger index: never displaces */
        int pos = m < k ? m : k - 1;
        while (pos > 0 && v > val[pos - 1]) { /* strict: earlier index stays ahead on ties */
            if (pos < k) { val[pos] = val[pos - 1]; idx[pos] = idx[pos - 1]; }
            --pos;
        }
        val[pos] = v; idx[pos] = n;
        if (m < k) ++m;
    }
    for (int i = m; i < k; ++i) { val[i] = -INFINITY; idx[i] = -1; }
}

/* full search: idx [nq,k] int64, val [nq,k] float.  Processes queries in chunks to bound memory. */
void knn_oracle_topk_ip(const float* X, int64_t N, int D, const float* Q, int nq, int k, int64_t* idx, float* val) {
    const int chunk = 64;
    float* sc = (float*)malloc((size_t)chunk * N * sizeof(float));
    for (int q0 = 0; q0 < nq; q0 += chunk) {
        int c = nq - q0 < chunk ? nq - q0 : chunk;
        knn_oracle_scores(X, N, D, Q + (size_t)q0 * D, c, sc);
#pragma omp parallel for schedule(dynamic)
        for (int q = 0; q < c; ++q)
            select_topk(sc + (size_t)q * N, N, k, idx + (size_t)(q0 + q) * k, val + (size_t)(q0 + q) * k);
    }
    free(sc);
}

/* selection only (scores supplied): used to check the HIP select kernel in isolation */
void knn_oracle_select(const float* scores, int64_t N, int nq, int k, int64_t* idx, float* val) {
#pragma omp parallel for schedule(dynamic)
    for (int q = 0; q < nq; ++q) select_topk(scores + (size_t)q * N, N, k, idx + (size_t)q * k, val + (size_t)q * k);
}
