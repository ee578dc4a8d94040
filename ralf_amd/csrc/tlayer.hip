// One launch per pre-norm transformer layer (forward) for SHORT sequences: the causal decoder layers (S = 5N <= 64 tokens per sample,
// self-attention + cross-attention over the memory + FFN) and the constraint-encoder layers (self-attention + FFN) of
//   image2layout/train/models/common/common.py:25-34,84-135,216-226 (nn.TransformerDecoderLayer / nn.TransformerEncoderLayer, norm_first).
// At B = 64 these layers are chains of 12 (7) launch-latency-bound kernels on 3 200 (256) rows each: ~100 us per decoder layer forward
// for 0.3 us of matrix work.  Here ONE workgroup owns a sample: its <= 64 rows stay in LDS from LayerNorm to the layer's output, the
// weights stream from L2 straight into MFMA operands (each lane loads the 16-byte k-slices of its own weight row, as gemm_skinny
// does), and every tensor the unfused backward needs (LayerNorm outputs and statistics, qkv, attention outputs + lse, the FFN hidden) is
// written out on the way -- the backward pass is unchanged.
//
// Arithmetic is that of the unfused kernels at the same rounding points (bf16 after LayerNorm, after every GEMM epilogue, after the
// attention; fp32 accumulation in k order; the same counter-based dropout masks), so the fused forward reproduces them bit for bit:
//   LayerNorm            = norm.hip ln_fwd_kernel (wave per row, lane owns 4 columns)
//   linear layers        = gemm_impl.h (32x32x16 MFMA chains in ascending k, epilogue order bias -> act -> dropout -> residual)
//   attention            = attention_mfma.hip attn_fwd_mfma (transposed 16x16x32 formulation, 32-key steps, log2-domain softmax)
#include <string.h>

#include "common.h"
#include "attn_core.h"

namespace {
typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

constexpr int TD = 256, TH = 8, TDH = 32, TFF = 1024, TS = 64;
constexpr int LDA = TD + 8;          // [64][264] activation strip: 528-byte rows = 4 banks apart, conflict-free ds_read_b128 over 16 rows
constexpr int LDQ = 3 * TD + 8;      // [64][776] q | k | v strip (1552-byte rows: the same 4-bank shift)
constexpr int NW = 8, NT = 512;
constexpr int BUFA_ELEMS = TS * LDA;                 // 33 792 B
constexpr int BUFB_ELEMS = 3 * TS * LDA;             // 101 376 B: the qkv strip (99 328 B) | Q + K tile + V tile | one FFN hidden chunk
constexpr int LDS_BYTES = (BUFA_ELEMS + BUFB_ELEMS) * 2 + 64 + 16;

// workgroup barrier for LDS hand-overs only: waits for this wave's LDS traffic, NOT for its global loads and stores -- the weight fragments
// requested for the next tile stay in flight across it, and the stores of the tensors kept for the backward pass drain in the background
// (__syncthreads() = a release / acquire fence: s_waitcnt vmcnt(0) at every phase boundary, ~2 k cycles each).  No thread of this kernel
// reads global memory another thread of it wrote.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ float wave_sum(float v) {
    return wave::sum64_desc(v);   // the descending butterfly, bit for bit, without the LDS crossbar (wave_ops.h)
}
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float xor_max(float v) { return wave::max_x16_x32(v); }
__device__ __forceinline__ float xor_sum(float v) { return wave::sum_x16_x32(v); }

// tools/tlayer_probe.hip builds this file with -DRALF_TLAYER_PROBE: s_memtime stamps per workgroup at the phase boundaries
#ifdef RALF_TLAYER_PROBE
__device__ unsigned long long tlayer_probe_buf[16 * 4096];
#define TL_PROBE(i)                                                                                                     \
    do {                                                                                                                \
        if (threadIdx.x == 0 && blockIdx.x < 4096) tlayer_probe_buf[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define TL_PROBE(i)
#endif

// ---- row pass: one wave per row (rows wave, wave + 8, ..), lane l owns columns 4l .. 4l+3 (norm.hip ln_fwd_kernel's assignment) ----
//   value  = STAGE ? bf16(stage[row][..] (fp32, LDS) + res[row][..]) : res[row][..]        (the GEMM epilogue's residual add and rounding)
//   XOUT:    value -> xout (global, 512 contiguous bytes per row)
//   LN:      LayerNorm(value) -> dst (LDS, rows >= S zero) + hout (global) + mean / rstd
// All of a wave's global rows are requested before the first is used (a wave owns 8 rows: one latency, not eight).
constexpr int STG_LD = TD + 4;   // fp32 staging rows of 1040 bytes
// a wave's rows (wave, wave + 8, ..) of a global [S][256] tensor, lane l = columns 4l .. 4l+3: requested early, consumed by row_pass
template <int RB>
__device__ __forceinline__ void load_rows(bf16x4 (&rr)[32 * RB / NW], const bf16* __restrict__ src, int S, int wave, int lane) {
#pragma unroll
    for (int k = 0; k < 32 * RB / NW; ++k) {
        const int row = wave + k * NW;
        rr[k] = *reinterpret_cast<const bf16x4*>(src + (int64_t)(row < S ? row : 0) * TD + lane * 4);
    }
}
// rr: in = the residual rows (load_rows, or the previous pass's output); out (STAGE) = the rows this pass produced (x1 / x2 / out)
template <bool STAGE, bool XOUT, bool LN, int RB = 2, bool HASRES = true>
__device__ __forceinline__ void row_pass(const float* stage, bf16x4 (&rr)[32 * RB / NW], bf16* __restrict__ xout, int S, const float* __restrict__ gamma,
                                         const float* __restrict__ beta, float eps, bf16* dst, bf16* __restrict__ hout, float* __restrict__ mean,
                                         float* __restrict__ rstd, int wave, int lane) {
    float g[4], bb[4];
    if (LN) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { g[i] = gamma[lane * 4 + i]; bb[i] = beta[lane * 4 + i]; }
    }
    constexpr int NR = 32 * RB / NW;   // 8 (4) rows per wave, processed TOGETHER: the two butterfly reductions per row are chains of 6 dependent cross-lane
                                  // reads each; row after row they cost ~10 k cycles per pass (s_memtime), interleaved over the rows one chain's latency
    float v[NR][4];
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        const int row = wave + k * NW;
        if (STAGE) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(stage + row * STG_LD + lane * 4);
            bf16x4 t;
#pragma unroll
            for (int i = 0; i < 4; ++i) t[i] = (bf16)(HASRES ? a[i] + (float)rr[k][i] : a[i]);
            if (XOUT && xout && row < S) *reinterpret_cast<bf16x4*>(xout + (int64_t)row * TD + lane * 4) = t;
            rr[k] = t;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[k][i] = (float)t[i];
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[k][i] = (float)rr[k][i];
        }
    }
    if (!LN) return;
    float s1[NR], s2[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) s1[k] = v[k][0] + v[k][1] + v[k][2] + v[k][3];
#pragma unroll
    for (int k = 0; k < NR; ++k) s1[k] = wave::sum64_desc(s1[k]);   // (NR independent chains of DPP / permlane-swap adds: wave_ops.h)
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        s1[k] *= (1.f / TD);   // mean
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float dd = v[k][i] - s1[k]; q = __fmaf_rn(dd, dd, q); }   // (explicit fma here and in ln_fwd_kernel: same bits)
        s2[k] = q;
    }
#pragma unroll
    for (int k = 0; k < NR; ++k) s2[k] = wave::sum64_desc(s2[k]);
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        const int row = wave + k * NW;
        const float mu = s1[k], rs = rsqrtf(__fmaf_rn(s2[k], 1.f / TD, eps));
        bf16x4 ov;
#pragma unroll
        for (int i = 0; i < 4; ++i) ov[i] = row < S ? (bf16)__fmaf_rn((v[k][i] - mu) * rs, g[i], bb[i]) : (bf16)0.f;
        if (row < S && hout) {   // (inference callers keep neither the normalised rows nor the statistics)
            *reinterpret_cast<bf16x4*>(hout + (int64_t)row * TD + lane * 4) = ov;
            if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
        }
        *reinterpret_cast<bf16x4*>(dst + row * LDA + lane * 4) = ov;
    }
}
// LDS rows [S][WIDTH] (leading dimension ld) -> global rows (leading dimension gld): 16 bytes per lane, row-contiguous
template <int WIDTH>
__device__ __forceinline__ void copy_out(const bf16* src, int ld, bf16* __restrict__ dstg, int64_t gld, int S, int tid) {
    constexpr int VPR = WIDTH / 8;
    for (int e = tid; e < S * VPR; e += NT) {
        const int r = e / VPR, c = e % VPR;
        *reinterpret_cast<uint4*>(dstg + (int64_t)r * gld + c * 8) = *reinterpret_cast<const uint4*>(src + r * ld + c * 8);
    }
}
// global rows [S][256] -> LDS strip [32 RB][LDA] (rows >= S zero)
template <int RB = 2>
__device__ __forceinline__ void copy_in(bf16* dst, const bf16* __restrict__ srcg, int S, int tid) {
    constexpr int VPR = TD / 8;
#pragma unroll
    for (int i = 0; i < 32 * RB * VPR / NT; ++i) {
        const int e = tid + NT * i, r = e / VPR, c = e % VPR;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (r < S) v = *reinterpret_cast<const uint4*>(srcg + (int64_t)r * TD + c * 8);
        *reinterpret_cast<uint4*>(dst + r * LDA + c * 8) = v;
    }
}

// ---- one 64 x 32 output tile of  A[64][256] (LDS, k-contiguous, leading dimension lda) x W[n0 .. n0+31][k0 .. k0+255]^T (global) ----
// The matrix core computes the TRANSPOSED tile (weights as the row operand), so accumulator register r of lane (l31, lh) is
// (row m = l31 [+32], column n0 + (r & 3) + 8 (r >> 2) + 4 lh): four consecutive columns per register group, as in gemm_impl.h.
// Weights arrive PACKED in fragment order (ralf_tlayer_pack): for the 32-row tile t and the 16-wide k-slice i of a [N][K] matrix, the 64 lanes'
// 16-byte operands (lane (l31, lh) = W[32 t + l31][16 i + 8 lh .. + 7]) are 1 KiB of consecutive memory, so one load instruction of a wave
// is one contiguous KiB.  (Each lane reading its own row of the row-major matrix -- 64 different cache lines per instruction, 16 bytes used
// of each -- ran the whole layer at the request rate of the vector memory pipeline: 12-15 k cycles per 32-column tile, s_memtime stamps.)
// The fragments live in registers (16 k-slices = 64 VGPRs); while a tile's matrix work consumes one half, the freed half is refilled with the
// NEXT tile's (`next`: that tile's lane pointer, or null), so the L2 latency of the weight stream hides under the previous tile's work.
struct WFrag { bf16x8 v[16]; };
__device__ __forceinline__ const bf16* w_ptr(const bf16* __restrict__ Wp, int kslices, int tile, int slice0, int lane) {
    return Wp + ((int64_t)tile * kslices + slice0) * 512 + lane * 8;
}
__device__ __forceinline__ void load_w_half(WFrag& w, int half, const bf16* __restrict__ p) {
#pragma unroll
    for (int i = 0; i < 8; ++i) w.v[half * 8 + i] = *reinterpret_cast<const bf16x8*>(p + (half * 8 + i) * 512);
}
__device__ __forceinline__ void load_w(WFrag& w, const bf16* __restrict__ p) { load_w_half(w, 0, p); load_w_half(w, 1, p); }
template <int RB>
__device__ __forceinline__ void tile_mma(f32x16 (&acc)[RB], const bf16* A, int lda, WFrag& w, const bf16* __restrict__ next, int lane) {
    const bf16* a0 = A + (lane & 31) * lda + (lane >> 5) * 8;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int i = half * 8; i < half * 8 + 8; ++i) {
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const bf16x8 x = *reinterpret_cast<const bf16x8*>(a0 + r * 32 * lda + i * 16);
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.v[i], x, acc[r], 0, 0, 0);
            }
        }
        if (next) load_w_half(w, half, next);
    }
}
template <int RB>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[RB]) {
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
}
// The tile's 16 bias values per lane are requested BEFORE its matrix work (load_bias) and added here: a bias load inside the epilogue put one
// L2 round trip (~1.5 k cycles, s_memtime stamps) on every tile's critical path.
struct Bias4 { float4 b[4]; };
__device__ __forceinline__ void load_bias(Bias4& bv, const float* __restrict__ bias, int n0, int lane) {
#pragma unroll
    for (int g = 0; g < 4; ++g) bv.b[g] = *reinterpret_cast<const float4*>(bias + n0 + 8 * g + 4 * (lane >> 5));
}
// epilogue walker: f(row m in 0 .. 32 RB - 1, first column n of 4 consecutive ones, values v[4] = accumulator + bias)
template <int RB, typename F>
__device__ __forceinline__ void tile_epilogue(const f32x16 (&acc)[RB], const Bias4& bv, int n0, int lane, F&& f) {
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v[4] = {acc[i][4 * g] + bv.b[g].x, acc[i][4 * g + 1] + bv.b[g].y, acc[i][4 * g + 2] + bv.b[g].z, acc[i][4 * g + 3] + bv.b[g].w};
            f(i * 32 + (lane & 31), n0 + 8 * g + 4 * (lane >> 5), v);
        }
}
__device__ __forceinline__ bf16x4 to_bf16x4(const float (&v)[4]) {
    bf16x4 t;
#pragma unroll
    for (int q = 0; q < 4; ++q) t[q] = (bf16)v[q];
    return t;
}
// GEMM-epilogue dropout on 4 consecutive elements of a contiguous [rows][N] output (ralf_dropout's mask: common.h); the stream keys of the
// launch's call ids are computed once per kernel
__device__ __forceinline__ void drop4(float (&v)[4], float p, const DropKeys k, uint32_t group) {   // group = (element index) / 4, below 2^32 (checked at launch)
    if (p > 0.f) {
        const uint32_t thr = drop_thr16(p);
        const float inv = 1.f / (1.f - p);
        const uint64_t h = drop_hash4k32(k, group);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = drop_keep(h, q, thr) ? v[q] * inv : 0.f;
    }
}

// ---- attention of ONE head by ONE wave over keys / values staged in LDS (attn_fwd_mfma's arithmetic) ----
// qt: query rows [64][ldq] at the head's columns; kt / vt: key / value rows [nkeys_staged][ldk] at the head's columns; Ms: per staged key
// 1 = masked.  State (m, l, o) per 16-query group lives in the caller (the cross-attention streams several key tiles).
struct AttnState { float m[4], l[4]; f32x4 o[4][2]; };
__device__ __forceinline__ void attn_init(AttnState& st) {
#pragma unroll
    for (int qg = 0; qg < 4; ++qg) {
        st.m[qg] = -__builtin_inff(); st.l[qg] = 0.f;
        st.o[qg][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; st.o[qg][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
}
template <bool CAUSAL>
__device__ __forceinline__ void attn_tile(AttnState& st, const bf16* qt, int ldq, const bf16* kt, const bf16* vt, int ldk, const uint8_t* Ms, bool tile_masked,
                                          int t0, int Sk, int Sq, float scale2, float p_drop, const uint32_t (&rowkey)[4], int lane) {
    const int g = lane >> 4, Ls = lane & 15;
    const uint32_t thr = attn_thr16(p_drop);
#pragma unroll
    for (int qg = 0; qg < 4; ++qg) {
        if (qg * 16 >= Sq) break;                        // (wave-uniform)
        const int qi = qg * 16 + Ls;
        const bf16x8 qf = *reinterpret_cast<const bf16x8*>(qt + (qg * 16 + Ls) * ldq + g * 8);
        float m = st.m[qg], l = st.l[qg];
#pragma unroll
        for (int s0 = 0; s0 < TS; s0 += 32) {
            if (t0 + s0 >= Sk) break;
            if (CAUSAL && t0 + s0 > qg * 16 + 15) break;   // every key of this step lies beyond every query of the group
            f32x4 s[2];
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kt + (s0 + blk * 16 + Ls) * ldk + g * 8);
                s[blk] = mfma16(kf, qf, (f32x4){0.f, 0.f, 0.f, 0.f});
            }
            const bool msk = tile_masked || CAUSAL;
            uint32_t mw0 = 0u, mw1 = 0u;
            if (msk) { mw0 = *reinterpret_cast<const uint32_t*>(Ms + s0 + 4 * g); mw1 = *reinterpret_cast<const uint32_t*>(Ms + s0 + 16 + 4 * g); }
            const bf16x8 pf = attn::fwd_step<2>(s, m, l, st.o[qg], scale2, msk, mw0, mw1, CAUSAL, t0 + s0 + 4 * g, qi, p_drop > 0.f, rowkey[qg], thr);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                // V^T fragment: matrix rows = 16 columns (dims) of the value tile, k-slots = keys {s0+4g+j, s0+16+4g+(j-4)}
                const bf16* q = vt + (s0 + 4 * g + (Ls >> 2)) * ldk + c * 16 + (Ls & 3) * 4;
                const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, q));
                const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, q + 16 * ldk));
                st.o[qg][c] = mfma16(__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7), pf, st.o[qg][c]);
            }
        }
        st.m[qg] = m; st.l[qg] = l;
    }
}
// normalise a head's output into the O strip (LDS, [64][LDA]; rows >= Sq zero) + lse.  st.m = maxima of the RAW scores, st.l = the lanes' shares of
// the sums (attn::fwd_step); kept probabilities went into P V unscaled: the 1 / (1 - p) of the dropout is part of the normalisation here
__device__ __forceinline__ void attn_finish(const AttnState& st, int Sq, int h, bf16* Os, float* __restrict__ lse, int lane, float scale, float p_drop) {
    const int g = lane >> 4, Ls = lane & 15;
    const float keep_scale = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
#pragma unroll
    for (int qg = 0; qg < 4; ++qg) {
        const int qi = qg * 16 + Ls;
        const float lsum = xor_sum(st.l[qg]);
        const float inv = lsum > 0.f ? keep_scale / lsum : 0.f;   // (all keys masked: output 0, lse -inf -- as attn_fwd_mfma)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            bf16x4 t;
#pragma unroll
            for (int r = 0; r < 4; ++r) t[r] = qi < Sq ? (bf16)(st.o[qg][c][r] * inv) : (bf16)0.f;
            *reinterpret_cast<bf16x4*>(Os + qi * LDA + h * TDH + c * 16 + 4 * g) = t;
        }
        if (qi < Sq && g == 0) lse[qi] = lsum > 0.f ? __fmaf_rn(st.m[qg], scale, __logf(lsum)) : -INFINITY;   // (as attn_fwd_mfma)
    }
}

// PART 0: encoder layer (self-attention block + feed-forward block)
// PART 1: decoder layer up to the cross-attention's queries (self-attention block, LayerNorm 2, q projection)
// PART 2: decoder layer from the cross-attention's output (out-projection 2 + residual, LayerNorm 3, feed-forward block)
// PART 4: LayerNorm 1 + the q | k | v projection on 64-row strips of ANY [rows, 256] tensor (the first two launches of a long-sequence layer)
// PART 3: the feed-forward block alone on 64-row strips of ANY [rows, 256] residual stream (LayerNorm 3 -> FFN -> + residual): the second half of
//         the image encoder's 16 384-row layers, whose attention spans 256 tokens and stays with the per-operation kernels
// The cross-attention itself (S x M scores per head over a memory of hundreds of rows) is per-score VALU work that wants the whole chip, not the
// B workgroups of this kernel: it stays ralf_attention_fwd between parts 1 and 2.
template <int PART, int RB = 2>   // RB: 32-row blocks per strip (1 = strips of <= 32 rows: the decode step's batch rows, part 2 only)
__global__ __launch_bounds__(NT) void tlayer_fwd_kernel(const RalfTLayerDesc d) {
    static_assert(RB == 2 || PART == 2 || PART == 3, "32-row strips: parts 2 and 3 only");
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];   // ONE LDS object
    bf16* bufA = reinterpret_cast<bf16*>(lds);              // the A operand of the running product: h1 | o1 | h2 | o2 | h3
    bf16* bufB = bufA + BUFA_ELEMS;                         // q|k|v strip, fp32 epilogue staging, q, one hidden chunk
    float* stage = reinterpret_cast<float*>(bufB);
    uint8_t* Ms = reinterpret_cast<uint8_t*>(bufB + BUFB_ELEMS);
    int* flags = reinterpret_cast<int*>(Ms + 64);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x, S = d.S;
    const int64_t row0 = (int64_t)b * S;
    const uint64_t seed = (d.p_attn > 0.f || d.p_res > 0.f) ? (uint64_t)d.seed[0] : 0;
    const DropKeys k_out1 = drop_keys(seed, d.call_out1), k_out2 = drop_keys(seed, d.call_out2), k_ffn1 = drop_keys(seed, d.call_ffn1), k_ffn2 = drop_keys(seed, d.call_ffn2);
    WFrag w;
    bf16x4 rr[32 * RB / NW];   // this wave's rows of the residual stream (x -> x1 -> x2), in registers from one row pass to the next
    TL_PROBE(0);

    if constexpr (PART == 0 || PART == 1 || PART == 4) {
        // ================= self-attention block: x1 = x + drop(attn(LN1(x)) Wo^T + bo) =================
        const float scale2 = d.scale * 1.4426950408889634f;   // log2 domain
        const bf16* xg = (const bf16*)d.x + row0 * TD;
        load_w(w, w_ptr((const bf16*)d.w_in, 16, wave, 0, lane));
        load_rows<RB>(rr, xg, S, wave, lane);
        row_pass<false, false, true>(nullptr, rr, nullptr, S, d.ln1_g, d.ln1_b, d.eps, bufA, (bf16*)d.h1 + row0 * TD, d.mean1 + row0, d.rstd1 + row0, wave, lane);
        if (PART != 4 && tid < TS) {
            const bool mk = tid >= S || (d.kpm && d.kpm[(int64_t)b * d.kpm_bs + tid]);
            Ms[tid] = mk ? 1 : 0;
            const unsigned long long any = __ballot(mk);
            if (tid == 0) flags[0] = any != 0ull ? 1 : 0;
        }
        lds_barrier();
        TL_PROBE(1);
#pragma unroll
        for (int t = 0; t < 3; ++t) {   // qkv = h1 Win^T + bin  -> LDS strip [64][LDQ]: wave h writes exactly head h's q, k and v columns
            const int n0 = (t * NW + wave) * 32;
            const bf16* next = t + 1 < 3 ? w_ptr((const bf16*)d.w_in, 16, (t + 1) * NW + wave, 0, lane) : (PART == 4 ? nullptr : w_ptr((const bf16*)d.w_o, 16, wave, 0, lane));
            f32x16 acc[2];
            zero_acc(acc);
            Bias4 bv;
            load_bias(bv, d.b_in, n0, lane);
            tile_mma(acc, bufA, LDA, w, next, lane);
            tile_epilogue(acc, bv, n0, lane, [&](int m, int n, float (&v)[4]) {
                *reinterpret_cast<bf16x4*>(bufB + m * LDQ + n) = to_bf16x4(v);
            });
        }
        lds_barrier();
        TL_PROBE(2);
        copy_out<3 * TD>(bufB, LDQ, (bf16*)d.qkv + row0 * (3 * TD), 3 * TD, S, tid);
        if constexpr (PART == 4) return;
        {   // wave = head: softmax(q k^T / sqrt(32) + masks) v, keys = the sample's own rows
            const int h = wave;
            AttnState st;
            attn_init(st);
            uint32_t rowkey[4];
#pragma unroll
            for (int qg = 0; qg < 4; ++qg)
                rowkey[qg] = d.p_attn > 0.f ? attn_rowkey(seed, d.call_attn1, ((uint64_t)b * TH + h) * S + qg * 16 + (lane & 15)) : 0u;
            const bool tm = flags[0] != 0;
            if (d.causal) attn_tile<true>(st, bufB + h * TDH, LDQ, bufB + TD + h * TDH, bufB + 2 * TD + h * TDH, LDQ, Ms, tm, 0, S, S, scale2, d.p_attn, rowkey, lane);
            else attn_tile<false>(st, bufB + h * TDH, LDQ, bufB + TD + h * TDH, bufB + 2 * TD + h * TDH, LDQ, Ms, tm, 0, S, S, scale2, d.p_attn, rowkey, lane);
            attn_finish(st, S, h, bufA, d.lse1 + ((int64_t)b * TH + h) * S, lane, d.scale, d.p_attn);
        }
        lds_barrier();
        TL_PROBE(3);
        copy_out<TD>(bufA, LDA, (bf16*)d.o1 + row0 * TD, TD, S, tid);
        {   // drop(o1 Wo^T + bo) -> fp32 staging (the strip is dead)
            f32x16 acc[2];
            zero_acc(acc);
            Bias4 bv;
            load_bias(bv, d.b_o, wave * 32, lane);
            load_rows<RB>(rr, xg, S, wave, lane);   // the residual rows of the pass behind the next barrier
            tile_mma(acc, bufA, LDA, w, PART == 1 ? w_ptr((const bf16*)d.w_q, 16, wave, 0, lane) : w_ptr((const bf16*)d.w1, 16, wave, 0, lane), lane);
            tile_epilogue(acc, bv, wave * 32, lane, [&](int m, int n, float (&v)[4]) {
                drop4(v, d.p_res, k_out1, ((uint32_t)(row0 + m) * TD + n) >> 2);
                *reinterpret_cast<f32x4*>(stage + m * STG_LD + n) = (f32x4){v[0], v[1], v[2], v[3]};
            });
        }
        lds_barrier();
        TL_PROBE(4);
        // x1 = x + staged rows; the next LayerNorm of the residual stream on the way
        if constexpr (PART == 1)
            row_pass<true, true, true>(stage, rr, (bf16*)d.x1 + row0 * TD, S, d.ln2_g, d.ln2_b, d.eps, bufA, (bf16*)d.h2 + row0 * TD, d.mean2 + row0, d.rstd2 + row0, wave, lane);
        else
            row_pass<true, true, true>(stage, rr, (bf16*)d.x1 + row0 * TD, S, d.ln3_g, d.ln3_b, d.eps, bufA, (bf16*)d.h3 + row0 * TD, d.mean3 + row0, d.rstd3 + row0, wave, lane);
        lds_barrier();
        TL_PROBE(5);
    }
    if constexpr (PART == 1) {   // q = h2 Wq^T + bq
        f32x16 acc[2];
        zero_acc(acc);
        Bias4 bv;
        load_bias(bv, d.b_q, wave * 32, lane);
        tile_mma(acc, bufA, LDA, w, nullptr, lane);
        tile_epilogue(acc, bv, wave * 32, lane, [&](int m, int n, float (&v)[4]) {
            *reinterpret_cast<bf16x4*>(bufB + m * LDA + n) = to_bf16x4(v);
        });
        lds_barrier();
        copy_out<TD>(bufB, LDA, (bf16*)d.q + row0 * TD, TD, S, tid);
        TL_PROBE(6);
        return;
    }
    if constexpr (PART == 2) {
        // ================= cross-attention block, second half: x2 = x1 + drop(o2 Wo2^T + bo2) =================
        const bf16* x1g = (const bf16*)d.x1 + row0 * TD;
        load_w(w, w_ptr((const bf16*)d.w_o2, 16, wave, 0, lane));
        load_rows<RB>(rr, x1g, S, wave, lane);
        copy_in<RB>(bufA, (const bf16*)d.o2 + row0 * TD, S, tid);
        lds_barrier();
        TL_PROBE(7);
        {
            f32x16 acc[RB];
            zero_acc(acc);
            Bias4 bv;
            load_bias(bv, d.b_o2, wave * 32, lane);
            tile_mma(acc, bufA, LDA, w, w_ptr((const bf16*)d.w1, 16, wave, 0, lane), lane);
            tile_epilogue(acc, bv, wave * 32, lane, [&](int m, int n, float (&v)[4]) {
                drop4(v, d.p_res, k_out2, ((uint32_t)(row0 + m) * TD + n) >> 2);
                *reinterpret_cast<f32x4*>(stage + m * STG_LD + n) = (f32x4){v[0], v[1], v[2], v[3]};
            });
        }
        lds_barrier();
        TL_PROBE(8);
        // (inference callers pass NULL for h3 / mean3 / rstd3 / hid: nothing is kept for a backward pass; x2 is always written, it is read back below)
        row_pass<true, true, true, RB>(stage, rr, (bf16*)d.x2 + row0 * TD, S, d.ln3_g, d.ln3_b, d.eps, bufA, d.h3 ? (bf16*)d.h3 + row0 * TD : nullptr,
                                       d.mean3 + row0, d.rstd3 + row0, wave, lane);
        lds_barrier();
    }
    if constexpr (PART == 3) {   // the strip's rows of the residual stream: kept in registers for the last add, normalised into LDS
        const bf16* xg = (const bf16*)d.x + row0 * TD;
        load_w(w, w_ptr((const bf16*)d.w1, 16, wave, 0, lane));
        load_rows<RB>(rr, xg, S, wave, lane);
        row_pass<false, false, true, RB>(nullptr, rr, nullptr, S, d.ln3_g, d.ln3_b, d.eps, bufA, (bf16*)d.h3 + row0 * TD, d.mean3 + row0, d.rstd3 + row0, wave, lane);
        lds_barrier();
    }
    TL_PROBE(9);

    // ================= feed-forward block: out = r + drop(W2 drop(relu(W1 LN3(r) + b1)) + b2) =================
    f32x16 yacc[RB];
    zero_acc(yacc);
    Bias4 bv2;
    load_bias(bv2, d.b2, wave * 32, lane);
    bf16* Hc = bufB;   // one 256-wide chunk of the hidden activation, [64][LDA]
    bf16* Zc = bufB + BUFA_ELEMS;   // ... and of its pre-activation (GELU only)
#pragma unroll 1
    for (int c = 0; c < TFF / TD; ++c) {
        {   // hidden columns c*256 + wave*32 ..: tile c*8 + wave of W1; next in the weight stream: W2[wave*32 ..][c*256 ..]
            f32x16 acc[RB];
            zero_acc(acc);
            if (c == 1) TL_PROBE(12);
            Bias4 bv;
            load_bias(bv, d.b1, c * TD + wave * 32, lane);
            tile_mma(acc, bufA, LDA, w, w_ptr((const bf16*)d.w2, TFF / 16, wave, c * 16, lane), lane);
            if (c == 1) TL_PROBE(13);
            const bool gelu = PART == 3 && d.act == RALF_ACT_GELU;   // (uniform)
            tile_epilogue(acc, bv, wave * 32, lane, [&](int m, int n, float (&v)[4]) {
                const int col = c * TD + n;
                if (gelu) {
                    if (d.z) *reinterpret_cast<bf16x4*>(Zc + m * LDA + n) = to_bf16x4(v);   // the pre-activation, as RalfGemmDesc.C2 keeps it
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = gelu_f(v[q]);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
                }
                drop4(v, d.p_res, k_ffn1, (uint32_t)(row0 + m) * (TFF / 4) + (col >> 2));
                *reinterpret_cast<bf16x4*>(Hc + m * LDA + n) = to_bf16x4(v);
            });
        }
        if (c == 1) TL_PROBE(14);
        lds_barrier();
        if (c == 1) TL_PROBE(15);
        if (d.hid) copy_out<TD>(Hc, LDA, (bf16*)d.hid + row0 * TFF + c * TD, TFF, S, tid);
        if (PART == 3 && d.act == RALF_ACT_GELU && d.z) copy_out<TD>(Zc, LDA, (bf16*)d.z + row0 * TFF + c * TD, TFF, S, tid);
        tile_mma(yacc, Hc, LDA, w, c + 1 < TFF / TD ? w_ptr((const bf16*)d.w1, 16, (c + 1) * NW + wave, 0, lane) : nullptr, lane);
        lds_barrier();
    }
    TL_PROBE(10);
    tile_epilogue(yacc, bv2, wave * 32, lane, [&](int m, int n, float (&v)[4]) {
        drop4(v, d.p_res, k_ffn2, ((uint32_t)(row0 + m) * TD + n) >> 2);
        *reinterpret_cast<f32x4*>(stage + m * STG_LD + n) = (f32x4){v[0], v[1], v[2], v[3]};
    });
    lds_barrier();
    if (PART == 3 && d.no_res) row_pass<true, true, false, RB, false>(stage, rr, (bf16*)d.out + row0 * TD, S, nullptr, nullptr, 0.f, nullptr, nullptr, nullptr, nullptr, wave, lane);
    else row_pass<true, true, false, RB>(stage, rr, (bf16*)d.out + row0 * TD, S, nullptr, nullptr, 0.f, nullptr, nullptr, nullptr, nullptr, wave, lane);
    TL_PROBE(11);
}

// ---- backward of the strip-wise tail (see RalfTLayerBwdDesc): the forward feed-forward loop with the roles turned -- the strip of dy_m is the
// LDS operand, W2^T streams where W1 did (dz chunk = 256 hidden columns, masked by the forward hidden), W1^T where W2 did (dh accumulates over
// the four chunks) -- then the LayerNorm backward as a row pass and the out-projection's data gradient ----
template <int RB>   // 32-row blocks per strip (1: strips of <= 32 rows, for row counts that would leave most CUs empty at 64)
__global__ __launch_bounds__(NT) void tlayer_bwd_kernel(const RalfTLayerBwdDesc d) {
    constexpr int RS = 32 * RB;   // rows of the strip's MFMA tiles
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
    bf16* bufA = reinterpret_cast<bf16*>(lds);          // dy_m strip, later g_m
    bf16* bufB = bufA + BUFA_ELEMS;
    bf16* Mc = bufB;                                     // forward hidden chunk (the ReLU / dropout mask source)
    bf16* Dc = bufB + BUFA_ELEMS;                        // dz chunk
    float* stage = reinterpret_cast<float*>(bufB);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x, S = d.S;
    const int64_t row0 = (int64_t)b * S;
    const float inv_keep = 1.f / (1.f - d.p);
    WFrag w;
    f32x16 yacc[RB];
    zero_acc(yacc);
    Bias4 zero4;
#pragma unroll
    for (int g = 0; g < 4; ++g) zero4.b[g] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (d.stage == 4) {
        // LayerNorm 1 + q | k | v projection backward: dh = dqkv Win (the strip of dqkv [64][768] is the LDS operand, Win^T [256][768] streams in
        // three 256-wide k chunks), then the LayerNorm backward below; no product behind it
        const int nk = d.nk;   // 256-wide chunks of the incoming gradient: 3 (dqkv, the packed in-projection) or 1 (a 256 -> 256 projection)
        load_w(w, w_ptr((const bf16*)d.w1t, nk * 16, wave, 0, lane));
        {
            const int vpr = nk * (TD / 8);
            for (int e = tid; e < RS * vpr; e += NT) {
                const int r = e / vpr, cc = e - r * vpr;
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (r < S) v = *reinterpret_cast<const uint4*>((const bf16*)d.dy_m + (row0 + r) * (int64_t)(nk * TD) + cc * 8);
                *reinterpret_cast<uint4*>(bufB + r * LDQ + cc * 8) = v;
            }
        }
        lds_barrier();
#pragma unroll 1
        for (int c = 0; c < nk; ++c)
            tile_mma(yacc, bufB + c * TD, LDQ, w, c + 1 < nk ? w_ptr((const bf16*)d.w1t, nk * 16, wave, (c + 1) * 16, lane) : (d.wot ? w_ptr((const bf16*)d.wot, 16, wave, 0, lane) : nullptr), lane);
    } else {
    load_w(w, w_ptr((const bf16*)d.w2t, 16, wave, 0, lane));
    copy_in<RB>(bufA, (const bf16*)d.dy_m + row0 * TD, S, tid);
    // the forward hidden chunk of the next iteration travels in registers (4 x 16 bytes per thread) while the current one is used
    uint4 hreg[2 * RB];
    auto hid_load = [&](int c) {
#pragma unroll
        for (int i = 0; i < 2 * RB; ++i) {
            const int e = tid + NT * i, r = e >> 5, cc = e & 31;
            hreg[i] = make_uint4(0u, 0u, 0u, 0u);
            if (r < S) hreg[i] = *reinterpret_cast<const uint4*>((const bf16*)d.hid + (row0 + r) * TFF + c * TD + cc * 8);
        }
    };
    auto hid_store = [&]() {
#pragma unroll
        for (int i = 0; i < 2 * RB; ++i) {
            const int e = tid + NT * i, r = e >> 5, cc = e & 31;
            *reinterpret_cast<uint4*>(Mc + r * LDA + cc * 8) = hreg[i];
        }
    };
    hid_load(0);
#pragma unroll 1
    for (int c = 0; c < TFF / TD; ++c) {
        hid_store();
        if (c + 1 < TFF / TD) hid_load(c + 1);
        lds_barrier();   // dy_m strip (first iteration) and the mask chunk are in place; the previous chunk's readers of Dc are done
        {   // dz columns c*256 + wave*32 ..: tile c*8 + wave of W2^T [1024][256]; next in the weight stream: W1^T[wave*32 ..][c*256 ..]
            f32x16 acc[RB];
            zero_acc(acc);
            tile_mma(acc, bufA, LDA, w, w_ptr((const bf16*)d.w1t, TFF / 16, wave, c * 16, lane), lane);
            const bool gelu = d.stage == 5;   // (uniform) hid = the GELU's pre-activation z
            tile_epilogue(acc, zero4, wave * 32, lane, [&](int m, int n, float (&v)[4]) {
                const bf16x4 hm = *reinterpret_cast<const bf16x4*>(Mc + m * LDA + n);
                if (gelu) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] *= gelu_grad((float)hm[q]);   // (RALF_AUX_GELU_GRAD)
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = (float)hm[q] > 0.f ? v[q] * inv_keep : 0.f;   // (RALF_AUX_RELU_MASK with aux_scale = 1 / (1 - p))
                }
                *reinterpret_cast<bf16x4*>(Dc + m * LDA + n) = to_bf16x4(v);
            });
        }
        lds_barrier();
        copy_out<TD>(Dc, LDA, (bf16*)d.dz + row0 * TFF + c * TD, TFF, S, tid);
        tile_mma(yacc, Dc, LDA, w, c + 1 < TFF / TD ? w_ptr((const bf16*)d.w2t, 16, (c + 1) * NW + wave, 0, lane) : (d.stage == 3 ? w_ptr((const bf16*)d.wot, 16, wave, 0, lane) : nullptr), lane);
    }
    }
    lds_barrier();   // every wave is done with Mc / Dc (the dqkv strip): the staging tile takes their place
    tile_epilogue(yacc, zero4, wave * 32, lane, [&](int m, int n, float (&v)[4]) {
        *reinterpret_cast<f32x4*>(stage + m * STG_LD + n) = (f32x4){v[0], v[1], v[2], v[3]};
    });
    lds_barrier();
    if (d.stage <= 1) {   // dh rows, as ralf_gemm would have written them
        bf16x4 none[RS / NW];
        row_pass<true, true, false, RB, false>(stage, none, (bf16*)d.g + row0 * TD, S, nullptr, nullptr, 0.f, nullptr, nullptr, nullptr, nullptr, wave, lane);
        return;
    }
    // ---- LayerNorm backward (norm.hip ln_bwd_kernel's arithmetic: wave per row, lane l owns columns 4l .. 4l+3), all 8 rows of a wave together ----
    constexpr int NR = RS / NW;
    float gm[4], ag[4] = {0.f, 0.f, 0.f, 0.f}, ab[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) gm[i] = d.ln3_g[lane * 4 + i];
    bf16x4 xr[NR], sk[NR];
    float mu[NR], rs[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        const int row = wave + k * NW, rc = row < S ? row : 0;
        xr[k] = *reinterpret_cast<const bf16x4*>((const bf16*)d.x2 + (row0 + rc) * TD + lane * 4);
        if (d.dy) sk[k] = *reinterpret_cast<const bf16x4*>((const bf16*)d.dy + (row0 + rc) * TD + lane * 4);
        mu[k] = d.mean3[row0 + rc]; rs[k] = d.rstd3[row0 + rc];
    }
    float dv[NR][4], xh[NR][4], s1[NR], s2[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        const int row = wave + k * NW;
        const f32x4 a = *reinterpret_cast<const f32x4*>(stage + row * STG_LD + lane * 4);
        const float wgt = row < S ? 1.f : 0.f;
        s1[k] = s2[k] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dv[k][i] = (float)(bf16)a[i];   // dh as the bf16 tensor ralf_gemm would have written
            xh[k][i] = ((float)xr[k][i] - mu[k]) * rs[k];
            ag[i] += wgt * dv[k][i] * xh[k][i];
            ab[i] += wgt * dv[k][i];
            const float dg = dv[k][i] * gm[i];
            s1[k] = __fmaf_rn(dv[k][i], gm[i], s1[k]); s2[k] = __fmaf_rn(dg, xh[k][i], s2[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < NR; ++k) { s1[k] = wave::sum64_desc(s1[k]); s2[k] = wave::sum64_desc(s2[k]); }
    lds_barrier();   // every wave has read its staged rows: bufB is free for the column sums, bufA for g_m
    const DropKeys k_out = drop_keys(d.p > 0.f ? (uint64_t)d.seed[0] : 0, d.call_out);
    const uint32_t thr = drop_thr16(d.p);
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        const int row = wave + k * NW;
        const float m1 = s1[k] * (1.f / TD), m2 = s2[k] * (1.f / TD);
        bf16x4 go, gmk;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float t = __fmaf_rn(-xh[k][i], m2, __fmaf_rn(dv[k][i], gm[i], -m1));
            go[i] = (bf16)(d.dy ? __fmaf_rn(rs[k], t, (float)sk[k][i]) : rs[k] * t);
        }
        gmk = go;
        if (d.p > 0.f) {
            const uint64_t hh = drop_hash4k(k_out, (uint64_t)((row0 + row) * TD + lane * 4) >> 2);
#pragma unroll
            for (int i = 0; i < 4; ++i) gmk[i] = (bf16)(drop_keep(hh, i, thr) ? (float)go[i] * inv_keep : 0.f);
        }
        if (row >= S) {
#pragma unroll
            for (int i = 0; i < 4; ++i) gmk[i] = (bf16)0.f;
        } else {
            *reinterpret_cast<bf16x4*>((bf16*)d.g + (row0 + row) * TD + lane * 4) = go;
            if (d.p > 0.f) *reinterpret_cast<bf16x4*>((bf16*)d.g_m + (row0 + row) * TD + lane * 4) = gmk;
        }
        *reinterpret_cast<bf16x4*>(bufA + row * LDA + lane * 4) = gmk;
    }
    if (d.dgamma) {   // column sums of this strip: wave partials -> LDS -> one atomic per column (as ln_bwd_kernel)
        float* red = stage;   // [2][8][256]
#pragma unroll
        for (int i = 0; i < 4; ++i) { red[(0 * NW + wave) * TD + lane * 4 + i] = ag[i]; red[(1 * NW + wave) * TD + lane * 4 + i] = ab[i]; }
    }
    lds_barrier();
    if (d.dgamma) {
        const float* red = stage;
        const int which = tid >> 8, cc = tid & 255;
        float t = 0.f;
#pragma unroll
        for (int wv = 0; wv < NW; ++wv) t += red[(which * NW + wv) * TD + cc];
        atomicAdd((which ? d.dbeta : d.dgamma) + cc, t);
    }
    if (!(d.stage == 3 || (d.stage == 4 && d.wot))) return;
    {   // d_o = g_m Wo: the out-projection's data gradient (W = Wo^T in fragment order, requested during the last dh tile)
        f32x16 acc[RB];
        zero_acc(acc);
        tile_mma(acc, bufA, LDA, w, nullptr, lane);
        lds_barrier();   // the column sums have been read: the strip's d_o rows go through the same LDS
        bf16* Oc = bufB;
        tile_epilogue(acc, zero4, wave * 32, lane, [&](int m, int n, float (&v)[4]) { *reinterpret_cast<bf16x4*>(Oc + m * LDA + n) = to_bf16x4(v); });
        lds_barrier();
        copy_out<TD>(Oc, LDA, (bf16*)d.d_o + row0 * TD, TD, S, tid);
    }
}

// ---- weights -> fragment order (see WFrag).  One 16-byte chunk per thread: chunk c of job j = lane (c & 63) of (tile, k-slice) c >> 6 ----
constexpr int PACK_MAX_JOBS = 96;
struct PackJobs { RalfPackJob j[PACK_MAX_JOBS]; };
__global__ __launch_bounds__(256) void tlayer_pack_kernel(const PackJobs jobs) {
    const RalfPackJob jb = jobs.j[blockIdx.y];
    const int kslices = jb.K / 16;
    const int64_t nchunks = (int64_t)(jb.N / 32) * kslices * 64;
    for (int64_t c = blockIdx.x * 256 + threadIdx.x; c < nchunks; c += (int64_t)gridDim.x * 256) {
        const int lane = (int)(c & 63);
        const int64_t ts = c >> 6;
        const int tile = (int)(ts / kslices), slice = (int)(ts % kslices);
        const int n = tile * 32 + (lane & 31), k0 = slice * 16 + (lane >> 5) * 8;
        if (!jb.transpose) {
            *reinterpret_cast<uint4*>((bf16*)jb.dst + c * 8) = *reinterpret_cast<const uint4*>((const bf16*)jb.src + (int64_t)n * jb.ld + k0);
        } else {   // element [n][k] = src[k][n]
            bf16x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = ((const bf16*)jb.src)[(int64_t)(k0 + e) * jb.ld + n];
            *reinterpret_cast<bf16x8*>((bf16*)jb.dst + c * 8) = v;
        }
    }
}
}  // namespace

extern "C" int ralf_tlayer_pack(const RalfPackJob* jobs, int njobs, void* stream) {
    RALF_REQUIRE(jobs && njobs > 0 && njobs <= PACK_MAX_JOBS, "tlayer_pack: 1 .. %d jobs per call (got %d)", PACK_MAX_JOBS, njobs);
    PackJobs pj;
    memset(&pj, 0, sizeof(pj));
    int64_t most = 0;
    for (int i = 0; i < njobs; ++i) {
        const RalfPackJob& j = jobs[i];
        RALF_REQUIRE(j.src && j.dst && j.N > 0 && j.K > 0 && j.N % 32 == 0 && j.K % 16 == 0 && j.ld >= (j.transpose ? j.N : j.K) && (j.transpose || j.ld % 8 == 0),
                     "tlayer_pack: job %d: [N %% 32 == 0][K %% 16 == 0] bf16, ld %% 8 == 0", i);
        RALF_REQUIRE((((uintptr_t)j.src | (uintptr_t)j.dst) & 15) == 0, "tlayer_pack: job %d: src / dst must be 16-byte aligned (16-byte vector accesses)", i);
        pj.j[i] = j;
        most = most > (int64_t)j.N * j.K / 8 ? most : (int64_t)j.N * j.K / 8;
    }
    hipLaunchKernelGGL(tlayer_pack_kernel, dim3((unsigned)((most + 1023) / 1024), njobs), dim3(256), 0, (hipStream_t)stream, pj);
    return ralf::check_launch("tlayer_pack");
}

extern "C" int ralf_tlayer_fwd(const RalfTLayerDesc* dp, void* stream) {
    RALF_REQUIRE(dp, "tlayer_fwd: null descriptor");
    const RalfTLayerDesc& d = *dp;
    RALF_REQUIRE(d.part >= 0 && d.part <= 4, "tlayer_fwd: part 0 (encoder layer), 1 or 2 (decoder layer before / after its cross-attention), 3 (feed-forward block), 4 (LayerNorm + qkv)");
    RALF_REQUIRE(d.B > 0 && d.S > 0 && d.S <= TS, "tlayer_fwd: needs 1 <= S <= %d rows per sample (got %d)", TS, d.S);
    RALF_REQUIRE((int64_t)d.B * d.S < (1 << 22), "tlayer_fwd: at most 2^22 rows per launch (32-bit dropout element indices)");
    RALF_REQUIRE(d.p_attn >= 0.f && d.p_attn < 1.f && d.p_res >= 0.f && d.p_res < 1.f && ((d.p_attn == 0.f && d.p_res == 0.f) || d.seed), "tlayer_fwd: dropout needs a seed");
    if (d.part == 4) RALF_REQUIRE(d.x && d.ln1_g && d.ln1_b && d.w_in && d.b_in && d.h1 && d.mean1 && d.rstd1 && d.qkv, "tlayer_fwd: part 4 needs x, LayerNorm 1, in_proj and h1 / mean1 / rstd1 / qkv");
    if (d.part == 0 || d.part == 1) {
        RALF_REQUIRE(d.x && d.ln1_g && d.ln1_b && d.w_in && d.b_in && d.w_o && d.b_o, "tlayer_fwd: self-attention block: null input / weight pointer");
        RALF_REQUIRE(d.h1 && d.mean1 && d.rstd1 && d.qkv && d.o1 && d.lse1 && d.x1, "tlayer_fwd: self-attention block: null output pointer");
        RALF_REQUIRE(!d.kpm || d.kpm_bs >= d.S, "tlayer_fwd: kpm row stride");
    }
    if (d.part == 1) RALF_REQUIRE(d.ln2_g && d.ln2_b && d.w_q && d.b_q && d.h2 && d.mean2 && d.rstd2 && d.q, "tlayer_fwd: part 1 needs LayerNorm 2, the q projection and their outputs");
    if (d.part == 2) RALF_REQUIRE(d.x1 && d.o2 && d.w_o2 && d.b_o2 && d.x2, "tlayer_fwd: part 2 needs x1, o2, the second out-projection and x2");
    if (d.part == 3) RALF_REQUIRE(d.x && (d.act == RALF_ACT_RELU || d.act == RALF_ACT_NONE || d.act == RALF_ACT_GELU), "tlayer_fwd: part 3 needs x; act = ReLU (default) or GELU");
    RALF_REQUIRE(d.part == 3 || (!d.z && !d.no_res && d.act != RALF_ACT_GELU), "tlayer_fwd: act / z / no_res belong to part 3");
    if (d.part != 1 && d.part != 4) {
        RALF_REQUIRE(d.ln3_g && d.ln3_b && d.w1 && d.b1 && d.w2 && d.b2, "tlayer_fwd: feed-forward block: null weight pointer");
        RALF_REQUIRE(d.out && ((d.h3 && d.mean3 && d.rstd3 && d.hid) || (d.part == 2 && !d.h3 && !d.mean3 && !d.rstd3 && !d.hid)),
                     "tlayer_fwd: feed-forward block: null output pointer (part 2 alone may run without h3 / mean3 / rstd3 / hid: inference)");
    }
    {   // every activation / packed-weight pointer is read or written in 16-byte vectors (bf16x8 / uint4)
        const void* al[] = {d.x, d.x1, d.x2, d.o1, d.o2, d.q, d.qkv, d.h1, d.h2, d.h3, d.hid, d.z, d.out, d.w_in, d.w_o, d.w_q, d.w_o2, d.w1, d.w2};
        uintptr_t bits = 0;
        for (const void* p : al) bits |= (uintptr_t)p;
        RALF_REQUIRE((bits & 15) == 0, "tlayer_fwd: activation and packed-weight pointers must be 16-byte aligned");
    }
    hipStream_t st = (hipStream_t)stream;
    if (d.part == 0) hipLaunchKernelGGL((tlayer_fwd_kernel<0>), dim3(d.B), dim3(NT), 0, st, d);
    else if (d.part == 1) hipLaunchKernelGGL((tlayer_fwd_kernel<1>), dim3(d.B), dim3(NT), 0, st, d);
    else if (d.part == 3 && d.S <= 32) hipLaunchKernelGGL((tlayer_fwd_kernel<3, 1>), dim3(d.B), dim3(NT), 0, st, d);
    else if (d.part == 3) hipLaunchKernelGGL((tlayer_fwd_kernel<3>), dim3(d.B), dim3(NT), 0, st, d);
    else if (d.part == 4) hipLaunchKernelGGL((tlayer_fwd_kernel<4>), dim3(d.B), dim3(NT), 0, st, d);
    else if (d.S <= 32) hipLaunchKernelGGL((tlayer_fwd_kernel<2, 1>), dim3(d.B), dim3(NT), 0, st, d);   // strips of one 32-row block
    else hipLaunchKernelGGL((tlayer_fwd_kernel<2>), dim3(d.B), dim3(NT), 0, st, d);
    return ralf::check_launch("tlayer_fwd");
}

extern "C" int ralf_tlayer_bwd(const RalfTLayerBwdDesc* dp, void* stream) {
    RALF_REQUIRE(dp, "tlayer_bwd: null descriptor");
    const RalfTLayerBwdDesc& d = *dp;
    RALF_REQUIRE(d.B > 0 && d.S > 0 && d.S <= TS && d.p >= 0.f && d.p < 1.f, "tlayer_bwd: needs 1 <= S <= %d rows per strip, 0 <= p < 1", TS);
    RALF_REQUIRE(d.stage == 1 || (d.stage >= 3 && d.stage <= 5), "tlayer_bwd: stage 1 (dz, dh), 3 (the whole tail), 4 (LayerNorm 1 + q | k | v projection) or 5 (GELU feed-forward)");
    RALF_REQUIRE(d.dy_m && d.w1t && d.g && (d.stage == 4 ? (d.nk == 1 || d.nk == 3) && (!d.wot || d.d_o) : (d.hid && d.w2t && d.dz)), "tlayer_bwd: null pointer (stage 4: nk = 1 or 3)");
    RALF_REQUIRE(d.stage == 1 || (d.x2 && d.mean3 && d.rstd3 && d.ln3_g && d.g_m && (d.p == 0.f || d.seed) && (d.stage != 3 || (d.dy && d.wot && d.d_o))),
                 "tlayer_bwd: stages 3 / 4 need the LayerNorm operands (3: and the skip gradient and the out-projection)");
    RALF_REQUIRE((int64_t)d.B * d.S < (1 << 22), "tlayer_bwd: at most 2^22 rows per launch (32-bit dropout element indices, like the forward)");
    {
        const void* al[] = {d.dy_m, d.dy, d.hid, d.dz, d.g, d.g_m, d.x2, d.d_o, d.w1t, d.w2t, d.wot};
        uintptr_t bits = 0;
        for (const void* p : al) bits |= (uintptr_t)p;
        RALF_REQUIRE((bits & 15) == 0, "tlayer_bwd: activation and packed-weight pointers must be 16-byte aligned");
    }
    if (d.S <= 32) hipLaunchKernelGGL(tlayer_bwd_kernel<1>, dim3(d.B), dim3(NT), 0, (hipStream_t)stream, d);
    else hipLaunchKernelGGL(tlayer_bwd_kernel<2>, dim3(d.B), dim3(NT), 0, (hipStream_t)stream, d);
    return ralf::check_launch("tlayer_bwd");
}
