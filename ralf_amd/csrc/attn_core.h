// Per-score arithmetic of the bf16 matrix-core attention kernels (attention_mfma.hip, tlayer.hip): ONE place, so that the forward kernels agree
// bit for bit among themselves and the backward kernels among themselves.  These loops -- not the matrix cores -- bound every attention kernel
// (rocprofv3 SQ_INSTS_VALU: 25 vector instructions per score in the round-4 forward, 31-34 in the backward kernels, four cycles each on a SIMD
// that also has to issue the MFMAs), so they are written for instruction count:
//   * scores stay RAW until the exponent: max over raw scores (v_max3), p = exp2(fma(s, scale * log2 e, -m * scale * log2 e)): no multiply per score;
//   * the running sum l is LANE-LOCAL (the four lane groups of a query are added once, in the epilogue), the accumulators are rescaled only in
//     the steps where some lane's maximum moved (wave-uniform branch);
//   * dropout keeps p UNSCALED (the 1 / (1 - p) goes into the epilogue's normalisation / the dV epilogue), selects in fp32 and converts PAIRS
//     (one v_cvt_pk_bf16_f32 per two scores instead of a conversion, a select and a byte permute per score);
#pragma once
#include "common.h"

namespace attn {
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void put2(bf16x8& d, int j, float a, float b) {   // d[j], d[j+1] = bf16(a), bf16(b): one packed conversion
    const f32x2 v = {a, b};
    const bf16x2 t = __builtin_convertvector(v, bf16x2);
    d[j] = t[0]; d[j + 1] = t[1];
}

// ---- forward: lane = one query (lane & 15); the lane group g = lane >> 4 holds keys keybase + {j (j < 4), 16 + j - 4 (j >= 4)} of a 32-key step,
// keybase = (first key of the step) + 4 g.  s = the step's raw scores (S^T accumulators).  State: m = running maximum of the RAW scores (-inf
// before the first unmasked key), l = this LANE's share of the running sum of 2^((s - m) scale2), o = the output accumulators.
// masked (wave-uniform): mw0 / mw1 = the per-key mask bytes of the lane's two key runs (non-zero = padded / beyond Sk), causal: keys > qi.
// Returns the probabilities as the k-operand of the P V product: dropped ones zero, kept ones NOT scaled by 1 / (1 - p).
template <int NC>
__device__ __forceinline__ bf16x8 fwd_step(const f32x4 (&s)[2], float& m, float& l, f32x4 (&o)[NC], float scale2, bool masked, uint32_t mw0, uint32_t mw1,
                                           bool causal, int keybase, int qi, bool drop, uint32_t rowkey, uint32_t thr) {
    float sj[8];
    if (masked) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool mk = (((j < 4 ? mw0 : mw1) >> (8 * (j & 3))) & 0xffu) || (causal && keybase + (j >> 2) * 16 + (j & 3) > qi);
            sj[j] = mk ? -__builtin_inff() : s[j >> 2][j & 3];
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) sj[j] = s[j >> 2][j & 3];
    }
    float mt = fmaxf(fmaxf(fmaxf(sj[0], sj[1]), fmaxf(sj[2], sj[3])), fmaxf(fmaxf(sj[4], sj[5]), fmaxf(sj[6], sj[7])));
    mt = wave::max_x16_x32(mt);
    const float mn = fmaxf(m, mt);
    // while a query has seen only masked keys mn = -inf: 0 is the reference then, and every exponential below is 2^-inf = 0
    const float mref = mn > -__builtin_inff() ? mn : 0.f;
    const float corr = __builtin_amdgcn_exp2f((m - mref) * scale2);
    const float mrs = mref * scale2;
    float p[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) p[j] = __builtin_amdgcn_exp2f(__builtin_fmaf(sj[j], scale2, -mrs));
    l = __builtin_fmaf(l, corr, ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7])));
    m = mn;
    if (__builtin_amdgcn_ballot_w64(corr != 1.f) != 0ull) {   // (matrix-core operands must not sit under DIVERGENT control flow; this branch is uniform)
        asm volatile("" ::: "memory");   // a REAL branch: if-converted, it is the multiplies plus a select per accumulator register in every step
#pragma unroll
        for (int c = 0; c < NC; ++c) o[c] *= corr;
    }
    bf16x8 pf;
    if (drop) {
#pragma unroll
        for (int j = 0; j < 8; j += 2) {   // keys j, j + 1 of this lane are an (even, odd) pair: one hash for both
            const uint32_t h = attn_rng2x16(rowkey, (uint32_t)(keybase + (j >> 2) * 16 + (j & 3)) >> 1);
            put2(pf, j, (h & 0xffffu) >= thr ? p[j] : 0.f, (h >> 16) >= thr ? p[j + 1] : 0.f);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; j += 2) put2(pf, j, p[j], p[j + 1]);
    }
    return pf;
}

// ---- backward, per-key form: lane = one key kj (lane & 15 of the key block); its lane group g holds queries q0 + {j (j < 4), 16 + j - 4 (j >= 4)} of a
// 32-query step, q0 = (first query of the step) + 4 g.  s / dp = the S and dP accumulators, L4 / D4 / R4 = log-sum-exp (log2 domain), delta and
// dropout row key of those queries.  Out: pf = P (dropped ones zero, kept ones UNSCALED: the caller scales dV by 1 / (1 - p) once) and
// dsf = dS = P o (dropout(dP) - delta), both as the k-operands of the dV / dK (and dQ) products.
// MASKS (wave-uniform): this lane's key is masked, queries beyond Sq, causal.  A masked probability is a SELECTED zero, never 0 x exp2(..):
// the exponent of a padded key's score may overflow.
template <bool MASKS>
__device__ __forceinline__ void bwd_step_keys(const f32x4 (&s)[2], const f32x4 (&dp)[2], const f32x4 (&L4)[2], const f32x4 (&D4)[2], const u32x4 (&R4)[2],
                                              float scale2, bool kmasked, bool causal, int kj, int q0, int Sq, bool drop, uint32_t thr, float inv_keep,
                                              bf16x8& pf, bf16x8& dsf) {
    float e[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        e[j] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[j >> 2][j & 3], scale2, -L4[j >> 2][j & 3]));
        if (MASKS) {
            const int qi = q0 + (j >> 2) * 16 + (j & 3);
            e[j] = (kmasked || qi >= Sq || (causal && kj > qi)) ? 0.f : e[j];
        }
    }
    if (drop) {
        const uint32_t pair = (uint32_t)kj >> 1, sh = ((uint32_t)kj & 1u) * 16u;   // the lane's key: one field of its pair's hash
        float a[8], b[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool keep = __builtin_amdgcn_ubfe(attn_rng2x16(R4[j >> 2][j & 3], pair), sh, 16u) >= thr;
            const float t1 = e[j] * D4[j >> 2][j & 3];
            a[j] = keep ? e[j] : 0.f;
            b[j] = keep ? __builtin_fmaf(e[j] * inv_keep, dp[j >> 2][j & 3], -t1) : -t1;
        }
#pragma unroll
        for (int j = 0; j < 8; j += 2) { put2(pf, j, a[j], a[j + 1]); put2(dsf, j, b[j], b[j + 1]); }
    } else {
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            put2(pf, j, e[j], e[j + 1]);
            put2(dsf, j, e[j] * (dp[j >> 2][j & 3] - D4[j >> 2][j & 3]), e[j + 1] * (dp[(j + 1) >> 2][(j + 1) & 3] - D4[(j + 1) >> 2][(j + 1) & 3]));
        }
    }
}

// ---- backward, per-query form (dQ kernel): lane = one query, keys as in fwd_step.  lse2 / delta / rowkey of the lane's query.  Out: dsf only.
template <bool MASKS>
__device__ __forceinline__ bf16x8 bwd_step_queries(const f32x4 (&s)[2], const f32x4 (&dp)[2], float lse2, float delta, float scale2, uint32_t mw0, uint32_t mw1,
                                                   bool causal, int keybase, int qi, bool qok, bool drop, uint32_t rowkey, uint32_t thr, float inv_keep) {
    float e[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        e[j] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[j >> 2][j & 3], scale2, -lse2));
        if (MASKS) {
            const bool mk = (((j < 4 ? mw0 : mw1) >> (8 * (j & 3))) & 0xffu) || (causal && keybase + (j >> 2) * 16 + (j & 3) > qi) || !qok;
            e[j] = mk ? 0.f : e[j];
        }
    }
    bf16x8 dsf;
    if (drop) {
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            const uint32_t h = attn_rng2x16(rowkey, (uint32_t)(keybase + (j >> 2) * 16 + (j & 3)) >> 1);
            const float t0 = e[j] * delta, t1 = e[j + 1] * delta;
            put2(dsf, j, (h & 0xffffu) >= thr ? __builtin_fmaf(e[j] * inv_keep, dp[j >> 2][j & 3], -t0) : -t0,
                 (h >> 16) >= thr ? __builtin_fmaf(e[j + 1] * inv_keep, dp[(j + 1) >> 2][(j + 1) & 3], -t1) : -t1);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; j += 2)
            put2(dsf, j, e[j] * (dp[j >> 2][j & 3] - delta), e[j + 1] * (dp[(j + 1) >> 2][(j + 1) & 3] - delta));
    }
    return dsf;
}
}  // namespace attn
