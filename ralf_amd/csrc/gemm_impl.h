// MFMA GEMM for gfx950 with fused epilogues; the one contraction kernel behind every Linear,
// attention projection, FFN, vocab head and (through the implicit-im2col gather) every Conv2d
// of the RALF train step.
//
//   C[z][m][n] = epi( alpha * sum_k A[z][m][k] * B[z][k][n] )
//
// Replaces torch.nn.functional.linear / conv2d and their autograd backward as called from
//   nn.TransformerEncoderLayer / DecoderLayer / MultiheadAttention
//       (image2layout/train/models/retrieval_augmented_autoreg.py:116-126, common/common.py:25-34)
//   FeedForward / Attention (common/attention.py:15-71), BaseDecoder.head (common/common.py:38-40)
//   ResnetBackbone convolutions (common/image.py:80-111)
//
// Design:
//   * 64x64 output tile per 4-wave workgroup (2x2 waves of 32x32) or 128x128 per 8-wave workgroup (2x4 waves of 64x32),
//     see launch_cfg; fragments of 32x32 (v_mfma_f32_32x32x16_bf16 or the exact-fp32
//     v_mfma_f32_32x32x2_f32); fp32 accumulate always.  The matrix core computes the TRANSPOSED tile
//     (weights as the row operand) so every lane ends up with 4 consecutive output columns: 8/16-byte
//     epilogue loads and stores instead of 2-byte ones.
//   * each operand is either "k-contiguous" (row-major [rows][K]) or "row-contiguous" ([K][rows]);
//     tiles are staged global -> registers -> LDS in their MEMORY order (coalesced 16-B loads) and
//     the row-contiguous case is fed to the matrix core with ds_read_b64_tr_b16 (bf16) / plain
//     ds_read_b32 (fp32), so NN / NT / TN products need no transposed copies in HBM.
//   * the "pixel x (kh,kw,c)" operand of a convolution is gathered on the fly from NHWC (implicit
//     im2col; forward / data-gradient / weight-gradient all use the same gather).
//   * split-K for the weight-gradient shapes (tiny output, huge reduction), deterministic:
//     partial slabs + a reduce kernel that also runs the epilogue.
//   * epilogue: alpha, bias, ReLU/GELU(erf), activation-gradient masks, residual add, second
//     (pre-activation) output, fp32 or bf16 stores.
#pragma once
#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace {

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));   // 16-byte staging vector (a native vector: HIP's uint4 struct is copied with memcpy and can pin the staging arrays in scratch)
#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))


// division by a launch-time constant as multiply-high + shift (n < 2^31): the im2col gather decomposes a row
// into (image, y, x) and a column into (kh, kw, c) for every staged vector of every k-tile -- with hardware-less
// integer division that was ~190 VALU instructions per k-tile per wave (PMC: 48 VALU per MFMA on the 3x3 kernels).
struct FastDiv {
    uint32_t mul, shr, den;
    __host__ void set(uint32_t d) {
        den = d ? d : 1;
        if (den == 1) { mul = 0; shr = 0; return; }
        uint32_t lg = 0;
        while ((1u << lg) < den) ++lg;
        const uint32_t p = 31 + lg;
        mul = (uint32_t)((((uint64_t)1 << p) + den - 1) / den);
        shr = p - 32;
    }
    __device__ __forceinline__ uint32_t div(uint32_t n) const { const uint32_t t = __umulhi(n, mul) >> shr; return den == 1 ? n : t; }   // (a select, not a branch)
    __device__ __forceinline__ void divmod(uint32_t n, int& q, int& r) const { const uint32_t t = div(n); q = (int)t; r = (int)(n - t * den); }
};

// tools/gemm_probe.hip builds ONE instantiation with -DRALF_GEMM_PROBE: per-workgroup phase time stamps (s_memtime) + placement
#ifdef RALF_GEMM_PROBE
__device__ unsigned long long ralf_probe_buf[8 * 65536];
#define RALF_PROBE(i)                                                                                                          \
    do {                                                                                                                       \
        if (threadIdx.x == 0 && blockIdx.x < 65536) ralf_probe_buf[(size_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define RALF_PROBE(i)
#endif

// (experiment switch, off: persistent workgroups measured -4...+6 % in round 1; NOT maintained since the gather loaders keep per-tile state in
//  registers -- a build with it on runs the train step at 89 ms)
#ifndef RALF_GEMM_PERSISTENT
#define RALF_GEMM_PERSISTENT 0
#endif

struct KParams {
    RalfGemmDesc d;
    FastDiv fd_hw, fd_rw, fd_sc, fd_kw, fd_st;   // RH*RW, RW, SC, KW, stride of d.g
    FastDiv fd_tap;                              // stride in mode 1 (data gradient), 1 in mode 0: the divisor of tap_offset
    int tiles_m, tiles_n, nwg;
    int mfast;        // tile order: 0 = column tiles fastest (consecutive workgroups of an XCD share an A tile: B small / L2-resident), 1 = row tiles fastest
    int kchunk;       // K range handled by one split (multiple of BK)
    float* partial;   // [split][batch][M][N] fp32 (splitk > 1)
    float* colsum;    // grouped weight gradients (gemm_body<.., CS = true>): fp32 [M], += the column sums of the k-major A operand (the bias gradient
    float* colsum_partial;   // that goes with dW = dy^T x), or NULL; splitk > 1: slabs [split][M] instead, summed by the grouped reduce kernel
    int vec_epi;      // leading dims / bases allow 4-wide epilogue accesses
    int fast;         // interior fast path: aligned operands, K range a multiple of BK (no per-tile bounds math)
    int tapuni;       // gather = 1 and channels % BK == 0: every k-tile lies inside one (kh, kw) tap
    // lean gather loaders (32-bit element offsets, no per-vector division, see gemm_body):
    int ln_sgn, ln_sh, ln_pm;       // gather 1: tap sign (+1 forward, -1 data gradient), log2 / mask of the tap stride (data gradient: the conv stride)
    int img, swsc;                  // elements of one source image, of one source row
    int inc_b, inc_y, inc_x;        // gather 2: BK rows of the pixel grid = inc_b ELEMENTS of whole images + inc_y rows + inc_x pixels
    // data gradient of a STRIDE-2 convolution by parity classes (GATHER 14): rows are ordered (class, image, y / 2, x / 2), class = (y & 1, x & 1)
    FastDiv fd_q, fd_hw2, fd_rw2;   // rows per class (M / 4), (RH / 2) * (RW / 2), RW / 2
    int par;                        // 1: take the parity form (set by ralf_gemm)
    // 3 x 3 / stride-1 convolutions with the tile's input PATCH resident in LDS (GATHER 15): a tile = (tile rows / SW) whole image rows
    int patch;                      // take the patch form (set by ralf_gemm): 1 = 128 x 128 tiles, 2 = 256 x 128, 3 = 256 x 64
    int p_pw, p_str;                // patch width in pixels (SW + 2), bytes per patch pixel (2 SC + 16)
    int p_swsh, p_c8sh;             // log2(SW), log2(SC / 8)
    FastDiv fd_pw;                  // p_pw
};

// virtual row (class-major order) -> class, image, pixel of the output grid
__device__ __forceinline__ void par_decompose(const KParams& P, uint32_t row, int& cls, int& b, int& ry, int& rx) {
    int rem, rem2, y2, x2;
    P.fd_q.divmod(row, cls, rem);
    P.fd_hw2.divmod((uint32_t)rem, b, rem2);
    P.fd_rw2.divmod((uint32_t)rem2, y2, x2);
    ry = 2 * y2 + (cls >> 1);
    rx = 2 * x2 + (cls & 1);
}
__device__ __forceinline__ int par_real_row(const KParams& P, int row) {
    int cls, b, ry, rx;
    par_decompose(P, (uint32_t)row, cls, b, ry, rx);
    return (b * P.d.g.RH + ry) * P.d.g.RW + rx;
}

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

template <typename T> struct TT;
template <> struct TT<float> {
    static constexpr int VEC = 4, BK = 32, KSTEP = 2;
    static constexpr int LDK = 32;    // k-contiguous tile: [rows][32], rotated
    static constexpr int RPAD = 0;    // row-contiguous tile: [32][rows]
};
template <> struct TT<bf16> {
    static constexpr int VEC = 8, BK = 64, KSTEP = 16;
    static constexpr int LDK = 72;    // [rows][64 + 8 pad]
    static constexpr int RPAD = 8;    // [64][rows + 8 pad]
};

// (the epilogue dropout mask: drop_hash4 / drop_apply, common.h -- shared with ralf_dropout, the LayerNorm backward and tlayer.hip)

// (gelu_f / gelu_grad: common.h)
template <typename T> __device__ __forceinline__ float ldf(const void* p, int64_t i);
template <> __device__ __forceinline__ float ldf<float>(const void* p, int64_t i) { return ((const float*)p)[i]; }
template <> __device__ __forceinline__ float ldf<bf16>(const void* p, int64_t i) { return (float)((const bf16*)p)[i]; }
template <typename T> __device__ __forceinline__ void stf(void* p, int64_t i, float v);
template <> __device__ __forceinline__ void stf<float>(void* p, int64_t i, float v) { ((float*)p)[i] = v; }
template <> __device__ __forceinline__ void stf<bf16>(void* p, int64_t i, float v) { ((bf16*)p)[i] = (bf16)v; }

// epilogue on one element (tails / unaligned outputs and the scalar split-K reducer); EPI as in epilogue_store4.
// (A non-inlined version was tried to shrink the code: device function calls cost far more -- 3x slower kernels.)
template <typename T, int EPI>
__device__ __forceinline__ void epilogue_store(const RalfGemmDesc& d, int z0, int z1, int m, int n, float v) {
    v *= d.alpha;
    if (d.colscale) v *= d.colscale[n];
    if (d.bias) v += d.bias[z0 * d.sBias0 + n];
    const int64_t coff = z0 * d.sC0 + z1 * d.sC1 + (int64_t)m * d.ldc + n;
    if (EPI >= 2 && d.C2) {  // pre-activation copy (needed by the activation gradient)
        if (d.out_f32) stf<float>(d.C2, coff, v); else stf<T>(d.C2, coff, v);
    }
    if (d.act == RALF_ACT_RELU) v = v > 0.f ? v : 0.f;
    else if (EPI >= 2 && d.act == RALF_ACT_GELU) v = gelu_f(v);
    if (EPI >= 1 && d.drop_p > 0.f) {  // element index = m*N + n of the contiguous [M,N] output (single batch)
        v = drop_keep1((uint64_t)d.seed[0], d.call_id, (uint64_t)m * d.N + n, drop_thr16(d.drop_p)) ? v * (1.f / (1.f - d.drop_p)) : 0.f;
    }
    if (EPI >= 1 && d.aux) {
        const float a = ldf<T>(d.aux, coff);
        if (d.aux_mode == RALF_AUX_RELU_MASK) v = a > 0.f ? v * d.aux_scale : 0.f;
        else if (EPI >= 2 && d.aux_mode == RALF_AUX_GELU_GRAD) v *= gelu_grad(a);
    }
    if (d.res) v += ldf<T>(d.res, z0 * d.sR0 + z1 * d.sR1 + (int64_t)m * d.ldr + n);
    if (d.act == RALF_ACT_RELU_POST) v = v > 0.f ? v : 0.f;
    if (EPI >= 2 && d.atomic_out) {   // (no early `return` in these helpers: it defeats unrolling of the caller's accumulator loops)
        atomicAdd((float*)d.C + coff, v);
    } else {
        if (d.accumulate) v += d.out_f32 ? ldf<float>(d.C, coff) : ldf<T>(d.C, coff);
        if (d.out_f32) stf<float>(d.C, coff, v); else stf<T>(d.C, coff, v);
    }
}

// W (4 or 8) consecutive elements <-> fp32 registers; 8 x bf16 is one 16-byte access, 8 x fp32 two
template <typename T, int W> struct VIO;
template <int W> struct VIO<float, W> {
    static __device__ __forceinline__ void ld(const void* p, int64_t i, float (&v)[W]) {
#pragma unroll
        for (int h = 0; h < W / 4; ++h) {
            const float4 t = *reinterpret_cast<const float4*>((const float*)p + i + 4 * h);
            v[4 * h] = t.x; v[4 * h + 1] = t.y; v[4 * h + 2] = t.z; v[4 * h + 3] = t.w;
        }
    }
    static __device__ __forceinline__ void st(void* p, int64_t i, const float (&v)[W]) {
#pragma unroll
        for (int h = 0; h < W / 4; ++h)
            *reinterpret_cast<float4*>((float*)p + i + 4 * h) = make_float4(v[4 * h], v[4 * h + 1], v[4 * h + 2], v[4 * h + 3]);
    }
};
template <> struct VIO<bf16, 4> {
    static __device__ __forceinline__ void ld(const void* p, int64_t i, float (&v)[4]) {
        const bf16x4 t = *reinterpret_cast<const bf16x4*>((const bf16*)p + i);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = (float)t[q];
    }
    static __device__ __forceinline__ void st(void* p, int64_t i, const float (&v)[4]) {
        bf16x4 t;
#pragma unroll
        for (int q = 0; q < 4; ++q) t[q] = (bf16)v[q];
        *reinterpret_cast<bf16x4*>((bf16*)p + i) = t;
    }
};
template <> struct VIO<bf16, 8> {
    static __device__ __forceinline__ void ld(const void* p, int64_t i, float (&v)[8]) {
        const bf16x8 t = *reinterpret_cast<const bf16x8*>((const bf16*)p + i);
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = (float)t[q];
    }
    static __device__ __forceinline__ void st(void* p, int64_t i, const float (&v)[8]) {
        bf16x8 t;
#pragma unroll
        for (int q = 0; q < 8; ++q) t[q] = (bf16)v[q];
        *reinterpret_cast<bf16x8*>((bf16*)p + i) = t;
    }
};

// epilogue on W consecutive columns n..n+W-1 of row m (same semantics as epilogue_store).
// EPI selects how much of the epilogue is compiled in (code size = issue slots and I-cache):
//   0: alpha, bias, ReLU, residual, accumulate      1: + fused dropout, ReLU-gradient mask
//   2: everything (GELU, GELU gradient, pre-activation copy, atomics)
template <typename T, int EPI, int W>
__device__ __forceinline__ void epilogue_storev(const RalfGemmDesc& d, int z0, int z1, int m, int n, float (&v)[W]) {
    const int64_t coff = z0 * d.sC0 + z1 * d.sC1 + (int64_t)m * d.ldc + n;
#pragma unroll
    for (int q = 0; q < W; ++q) v[q] *= d.alpha;
    if (d.colscale) {
        float c[W];
        VIO<float, W>::ld(d.colscale, n, c);
#pragma unroll
        for (int q = 0; q < W; ++q) v[q] *= c[q];
    }
    if (d.bias) {
        float b[W];
        VIO<float, W>::ld(d.bias, z0 * d.sBias0 + n, b);
#pragma unroll
        for (int q = 0; q < W; ++q) v[q] += b[q];
    }
    if (EPI >= 2 && d.C2) { if (d.out_f32) VIO<float, W>::st(d.C2, coff, v); else VIO<T, W>::st(d.C2, coff, v); }
    if (d.act == RALF_ACT_RELU) {
#pragma unroll
        for (int q = 0; q < W; ++q) v[q] = fmaxf(v[q], 0.f);
    } else if (EPI >= 2 && d.act == RALF_ACT_GELU) {
#pragma unroll
        for (int q = 0; q < W; ++q) v[q] = gelu_f(v[q]);
    }
    if (EPI >= 1 && d.drop_p > 0.f) {
        drop_apply<W>(v, (uint64_t)d.seed[0], d.call_id, (uint64_t)m * d.N + n, drop_thr16(d.drop_p), 1.f / (1.f - d.drop_p));
    }
    if (EPI >= 1 && d.aux) {
        float a[W];
        VIO<T, W>::ld(d.aux, coff, a);
        if (d.aux_mode == RALF_AUX_RELU_MASK) {
#pragma unroll
            for (int q = 0; q < W; ++q) v[q] = a[q] > 0.f ? v[q] * d.aux_scale : 0.f;
        } else if (EPI >= 2 && d.aux_mode == RALF_AUX_GELU_GRAD) {
#pragma unroll
            for (int q = 0; q < W; ++q) v[q] *= gelu_grad(a[q]);
        }
    }
    if (d.res) {
        float r[W];
        VIO<T, W>::ld(d.res, z0 * d.sR0 + z1 * d.sR1 + (int64_t)m * d.ldr + n, r);
#pragma unroll
        for (int q = 0; q < W; ++q) v[q] += r[q];
    }
    if (d.act == RALF_ACT_RELU_POST) {
#pragma unroll
        for (int q = 0; q < W; ++q) v[q] = fmaxf(v[q], 0.f);
    }
    if (EPI >= 2 && d.atomic_out) {
#pragma unroll
        for (int q = 0; q < W; ++q) atomicAdd((float*)d.C + coff + q, v[q]);
    } else {
        if (d.accumulate) {
            float c[W];
            if (d.out_f32) VIO<float, W>::ld(d.C, coff, c); else VIO<T, W>::ld(d.C, coff, c);
#pragma unroll
            for (int q = 0; q < W; ++q) v[q] += c[q];
        }
        if (d.out_f32) VIO<float, W>::st(d.C, coff, v); else VIO<T, W>::st(d.C, coff, v);
    }
}
// EPI 3 (RalfGemmDesc.bnb_*): the output is the gradient dz of z = relu(BN(x) (+ res)); 8 consecutive columns of row m.  After alpha / bias / res
// the ReLU mask bits zero the inactive elements, the value is rounded to T (what the BatchNorm backward will read) and the thread's
// running column sums s1 += dz, s2 += dz * (x - mean) are updated; the caller reduces them over the 64-row block.
template <typename T>
__device__ __forceinline__ void epilogue_storev_bnb(const RalfGemmDesc& d, int m, int n, float (&v)[8], const float (&mu)[8], float (&s1)[8], float (&s2)[8]) {
    const int64_t e = (int64_t)m * d.N + n;   // contiguous [M][N]: output, x and the mask bits share the element index
    float xv[8];
    VIO<T, 8>::ld(d.bnb_x, e, xv);
    const uint32_t mb = d.bnb_mask ? (uint32_t)d.bnb_mask[e >> 3] : 0xffu;
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] *= d.alpha;
    if (d.bias) {
        float b[8];
        VIO<float, 8>::ld(d.bias, n, b);
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] += b[q];
    }
    if (d.res) {
        float r[8];
        VIO<T, 8>::ld(d.res, (int64_t)m * d.ldr + n, r);
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] += r[q];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        v[q] = (float)(T)(((mb >> q) & 1u) ? v[q] : 0.f);
        s1[q] += v[q];
        s2[q] += v[q] * (xv[q] - mu[q]);
    }
    VIO<T, 8>::st(d.C, e, v);
}
// EPI 4 (RalfGemmDesc.flt_*): no output matrix -- the values of row m that reach the row's threshold go to the slots this COLUMN TILE owns in the
// row's candidate list: flt_list[(m * tiles_n + tn) * flt_cap + p], p counted by an LDS counter per tile row (no global atomics: device-scope
// returning atomics on ~1000 addresses ran the product at 850 us instead of 390), flt_count[m * tiles_n + tn] = hits of the tile (every
// (row, tile) pair is written exactly once: no zero-fill)
template <int W>
__device__ __forceinline__ void epilogue_filter(const RalfGemmDesc& d, int m, int n, const float (&v)[W], int ncols, float th, unsigned* lcnt, int2* slots) {
#pragma unroll
    for (int q = 0; q < W; ++q) {
        const float s = v[q] * d.alpha;
        if (s >= th && n + q < ncols) {   // (rare: the threshold passes ~1-2 % of the scores)
            const unsigned pos = atomicAdd(lcnt, 1u);
            if (pos < (unsigned)d.flt_cap) slots[pos] = make_int2(n + q, __float_as_int(s));
        }
    }
}
template <typename T, int EPI>
__device__ __forceinline__ void epilogue_store4(const RalfGemmDesc& d, int z0, int z1, int m, int n, float (&v)[4]) {
    epilogue_storev<T, EPI, 4>(d, z0, z1, m, n, v);
}

// ---- operand loaders -------------------------------------------------------------------------
// A "source matrix" is row-major with contiguous columns: plain (ptr + row*ld + col) or the
// implicit im2col matrix rows = pixels of a (RH x RW) grid per image, cols = (kh, kw, c).
struct RowInfo { int64_t base; int y0, x0; bool ok; };

template <bool GATHER>
__device__ __forceinline__ RowInfo row_info(const KParams& P, int64_t row, int64_t nrows, int64_t ld) {
    const RalfConvGeom& g = P.d.g;
    RowInfo r;
    r.ok = row < nrows;
    if (!GATHER) { r.base = row * ld; r.y0 = r.x0 = 0; return r; }
    int b, rem, ry, rx;
    P.fd_hw.divmod((uint32_t)row, b, rem);
    P.fd_rw.divmod((uint32_t)rem, ry, rx);
    r.base = (int64_t)b * g.SH * g.SW * g.SC;
    if (g.mode == 0) { r.y0 = ry * g.stride - g.pad; r.x0 = rx * g.stride - g.pad; }
    else             { r.y0 = ry + g.pad;            r.x0 = rx + g.pad; }
    return r;
}

// ---- loaders.  The gather (and the operands paired with a gather) load UNCONDITIONALLY from a clamped address and hand back a
// validity flag; the zeroing of invalid vectors happens when the registers are staged to LDS (a select right behind the load
// made the compiler wait for that load before it computed the next vector's address): a load behind `if (!ok) return zero` is compiled as a branch whose join waits with s_waitcnt vmcnt(0), i.e. the
// 4-8 staging vectors of a k-tile went out ONE AT A TIME (ISA of the 3x3 convolution kernels: load, vmcnt(0), load, vmcnt(0), ...).
__device__ __forceinline__ u32x4 sel_vec(bool ok, const u32x4 v) {
    u32x4 o;
    o.x = ok ? v.x : 0u; o.y = ok ? v.y : 0u; o.z = ok ? v.z : 0u; o.w = ok ? v.w : 0u;
    return o;
}
// source pixel of (output row r, tap kh, kw): offset in elements, validity folded into ok
__device__ __forceinline__ int64_t tap_offset(const RalfConvGeom& g, const KParams& P, const RowInfo& r, int kh, int kw, bool& ok) {
    // mode 0 (forward / weight gradient): source = (y0 + kh, x0 + kw);  mode 1 (data gradient): source = ((y0 - kh) / stride, ..) where
    // that division is exact.  ONE branch-free form: P.fd_tap divides by 1 in mode 0 (set at launch), sgn = +-1.
    const int sgn = g.mode ? -1 : 1;
    const int ty = r.y0 + sgn * kh, tx = r.x0 + sgn * kw;
    ok = ok && ty >= 0 && tx >= 0;
    int sy, sx, ry, rx;
    P.fd_tap.divmod((uint32_t)(ty < 0 ? 0 : ty), sy, ry);
    P.fd_tap.divmod((uint32_t)(tx < 0 ? 0 : tx), sx, rx);
    ok = ok && (ry | rx) == 0 && sy < g.SH && sx < g.SW;
    return r.base + ((int64_t)sy * g.SW + sx) * g.SC;
}
// 16-byte vector of VEC elements at (row, col..col+VEC-1); zero outside the matrix / the padding
template <typename T, bool GATHER>
__device__ __forceinline__ u32x4 load_vec(const T* __restrict__ p, const KParams& P, const RowInfo& r, int col, int ncols, bool aligned, bool& okout) {
    constexpr int VEC = TT<T>::VEC;
    const RalfConvGeom& g = P.d.g;
    const u32x4 z = {0u, 0u, 0u, 0u};
    if (GATHER) {
        bool ok = r.ok && col < ncols;
        int c, t, kw, kh;
        P.fd_sc.divmod((uint32_t)col, t, c);  // SC % VEC == 0: a vector never straddles a tap
        P.fd_kw.divmod((uint32_t)t, kh, kw);
        const int64_t off = tap_offset(g, P, r, kh, kw, ok) + c;
        okout = ok;
        return *reinterpret_cast<const u32x4*>(p + (off & -(int64_t)ok));   // (mask, not a select: no control flow)
    }
    okout = true;
    if (!r.ok || col >= ncols) return z;
    const T* q = p + r.base + col;
    if (aligned && col + VEC <= ncols) return *reinterpret_cast<const u32x4*>(q);
    T tmp[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) tmp[i] = (col + i < ncols) ? q[i] : (T)0.f;
    return *reinterpret_cast<u32x4*>(tmp);
}
// plain operand of a convolution GEMM (weights / dy): whole, 16-byte aligned vectors by construction (channel counts are
// multiples of VEC, checked at launch), so no element-wise fallback and no branch
template <typename T>
__device__ __forceinline__ u32x4 load_vec_al(const T* __restrict__ p, const RowInfo& r, int col, int ncols, bool& okout) {
    const bool ok = r.ok && col < ncols;
    okout = ok;
    return *reinterpret_cast<const u32x4*>(p + ((r.base + col) & -(int64_t)ok));
}

// im2col column (kh, kw, c) of a staged vector: loop-invariant for the weight-gradient gather (GATHER == 2), where
// the gathered operand's COLUMNS are fixed per thread and its rows (pixels) advance with k
struct ColInfo { int c, kh, kw; bool ok; };
__device__ __forceinline__ ColInfo col_info(const KParams& P, int col, int ncols) {
    ColInfo ci;
    int t;
    P.fd_sc.divmod((uint32_t)col, t, ci.c);
    P.fd_kw.divmod((uint32_t)t, ci.kh, ci.kw);
    ci.ok = col < ncols;
    return ci;
}
// staging registers -> LDS tile.  KC: [rows][LDK] (k-contiguous source), else [BK][LDR] (row-contiguous source)
template <typename T, bool KC, int NV, int RV, int LDR, int NT>
__device__ __forceinline__ void lds_stage(T* l, const u32x4 (&regs)[NV], int tid) {
    using X = TT<T>;
    constexpr int KV = X::BK / X::VEC;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = tid + NT * i;
        if constexpr (KC) {
            const int r = v / KV, kv = v % KV;
            if constexpr (sizeof(T) == 4) {  // fp32: rotate each row so 32 rows x same k hit 32 banks
                float* base = (float*)l + r * 32;
                const float* s = reinterpret_cast<const float*>(&regs[i]);
                const int rot = kv * 4 + r;
                base[(rot + 0) & 31] = s[0]; base[(rot + 1) & 31] = s[1];
                base[(rot + 2) & 31] = s[2]; base[(rot + 3) & 31] = s[3];
            } else {
                *reinterpret_cast<u32x4*>(l + r * X::LDK + kv * X::VEC) = regs[i];
            }
        } else {
            const int kr = v / RV, c = v % RV;
            *reinterpret_cast<u32x4*>(l + kr * LDR + c * X::VEC) = regs[i];
        }
    }
}

// AK: A is k-contiguous ([M][K]); else stored [K][M].   BKC: B is k-contiguous ([N][K]); else [K][N].
// GATHER: 0 none, 1 = A (k-contiguous) is an im2col matrix whose k-tiles lie inside one tap (channels % BK == 0), 4 = the same with any
// channel count (per-vector tap decomposition), 2 = B (row-contiguous) is an im2col matrix, 3 = no gather, interior fast path.
// FM, FN: 32x32 fragments per wave along m / n  ->  workgroup tile (64*FM) x (64*FN).
// NW: waves per workgroup, 4 (2 x 2 waves, FM x FN fragments each) or 8 (2 x 4 waves, FM x FN/2 fragments each; FN = 2 only)
// LDS bytes of one workgroup: two buffers of operand tiles, re-used as the fp32 C staging tile [64][BN + 4] of the epilogue
// GATHER 7 / 9 / 8 = the interior fast path (3) with an A-OPERAND TRANSFORM (RalfGemmDesc.at_*): 7 = mode 1 with a second operand (BatchNorm apply +
// residual), 9 = mode 1 without, 8 = mode 2 (BatchNorm backward apply).  Register-staged only: the transform happens between the global load
// and the LDS write; its per-channel coefficients sit in LDS behind the operand buffers (read with ds_read: the vector-memory queue keeps
// nothing but the prefetched tiles, so the prefetch distance survives).
constexpr int gemm_at_mode(int GATHER) { return (GATHER == 7 || GATHER == 9) ? 1 : GATHER == 8 ? 2 : 0; }
constexpr int AT_KMAX = 512;                       // channels of the transformed operand (K of the product)
constexpr int AT_LDS_BYTES = 3 * AT_KMAX * 4;      // c1 | c2 | c3
// GATHER 10 / 12 / 13 (round 6): the PIPELINED direct-to-LDS ring -- 3 / 4 / 2 stages, operand fragments double-buffered in registers, the k-tile's
// one barrier in front of its LAST k-slice (gemm_body: "pipelined ring"); 11 = 10 with the tap-uniform im2col gather of A through the ring
constexpr bool gemm_is_mt(int GATHER) { return GATHER >= 10 && GATHER <= 13; }
// GATHER 15 (round 6): 3 x 3 / stride-1 convolution, forward or data gradient, with the tile's INPUT PATCH resident in LDS -- see gemm_body
// LDS of a variant: the weights' ring (NST stages of [64 FN][64] k-tiles) in front of the patch.  64-column tiles (layer1: 64 channels) keep two
// stages and a 6 x 66-pixel patch so that TWO workgroups fit a CU (a tile's reduction is nine k-tiles: prologue and epilogue want company).
constexpr int gemm_patch_nst(int FN) { return FN == 1 ? 2 : 3; }
constexpr int gemm_patch_ring(int FN) { return gemm_patch_nst(FN) * 64 * FN * 64 * 2; }
constexpr int gemm_patch_bytes(int FN) { return FN == 1 ? 6 * 66 * 144 : 96 * 1024; }   // (FN >= 2: 10 x 18 pixels x 256 channels + 16 bytes per pixel = 95 040)
template <int GATHER, int FM>
constexpr int gemm_nbuf() { return (GATHER == 6 || GATHER == 10 || GATHER == 11) ? 3 : GATHER == 12 ? 4 : ((GATHER == 1 || GATHER == 4) && FM == 1) ? 1 : 2; }   // the prefetch-distance-1 kernels keep one buffer (GATHER 14: 128 x 128 only)
constexpr bool gemm_is_glds(int GATHER) { return GATHER == 5 || GATHER == 6 || gemm_is_mt(GATHER); }
// GLDS (GATHER 5 / 6): operand tiles go global -> LDS directly (global_load_lds_dwordx4), unpadded images whose 16-byte slots are
// XOR-swizzled through the SOURCE address; a ring of 2 / 3 stages
template <typename T, bool AK, bool BKC, int FM, int FN, int NBUF = 2, bool GLDS = false>
constexpr int gemm_lds_bytes() {
    using X = TT<T>;
    constexpr int BM = 64 * FM, BN = 64 * FN, BK = X::BK;
    constexpr int A_ELEMS = GLDS ? BM * BK : AK ? BM * X::LDK : BK * (BM + X::RPAD);
    constexpr int B_ELEMS = GLDS ? BN * BK : BKC ? BN * X::LDK : BK * (BN + X::RPAD);
    constexpr int ops = NBUF * (A_ELEMS + B_ELEMS) * (int)sizeof(T), cst = 64 * (BN + 4) * 4;   // NBUF operand buffers (see the k-loop)
    return ops > cst ? ops : cst;
}

// counted wait for this wave's oldest vector-memory operations (all but the N youngest) + its LDS reads; compiler memory fence
template <int N>
__device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory"); }
// workgroup barrier that does NOT drain the vector-memory queue (LDS-DMA loads stay in flight across it)
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// the body of one workgroup: output tiles bid0, bid0 + grid_x, ... of batch entry z (grid_x matters only for persistent
// launches).  Called by gemm_kernel (one problem per launch) and gemm_grouped_kernel (many problems per launch).
// WGM: waves along m (2 everywhere except the round-6 256-row tiles: 4 x 2 waves of 64 x 64)
template <typename T, bool AK, bool BKC, int GATHER, int FM, int FN, int EPI, int NW, bool CS = false, int WGM = 2>
__device__ __forceinline__ void gemm_body(const KParams& P, const int bid0, const int grid_x, const int z, const int nbatch, unsigned char* lds_raw) {
    using X = TT<T>;
    static_assert(!CS || (!AK && (GATHER == 5) && sizeof(T) == 2), "column sums of A: k-major bf16 A through the direct-to-LDS ring (the grouped weight gradients)");
    constexpr int VEC = X::VEC, BK = X::BK;
    constexpr int BM = 64 * FM, BN = 64 * FN;
    constexpr int LDRA = BM + X::RPAD, LDRB = BN + X::RPAD;
    constexpr int A_ELEMS = AK ? BM * X::LDK : BK * LDRA;
    constexpr int KV = BK / VEC;                                   // vectors along k (k-contiguous tile)
    constexpr int RVA = BM / VEC, RVB = BN / VEC;                  // vectors along rows (row-contiguous tile)
    constexpr int NT = 64 * NW, WGN = NW / WGM;                     // threads; waves along n (WGM along m)
    constexpr int WFM = FM * 2 / WGM, WFN = FN * 2 / WGN;          // 32x32 fragments per wave
    static_assert(NW == 4 || (NW == 8 && (FN == 2 || FN == 4 || GATHER == 15)), "8 waves: 2 x 4 over a 128- or 256-wide tile (or 4 x 2 over 256 x 128)");
    static_assert(WFM >= 1 && WFN >= 1 && WFM * WGM == 2 * FM && WFN * WGN == 2 * FN, "the wave grid tiles the workgroup tile");
    static_assert(WGM == 2 || (WFM == 2 && ((EPI != 3 && EPI != 4) || GATHER == 15)), "wave grids other than 2 x n: 64-row wave tiles, plain epilogues");
    constexpr int NVA = BM * BK / VEC / NT, NVB = BN * BK / VEC / NT;  // 16-byte vectors per thread per k-tile
    constexpr int CP = BN + 4;                                     // fp32 C staging tile [64][CP] (epilogue)
    constexpr int B_ELEMS = BKC ? BN * X::LDK : BK * LDRB;
    T* const la0 = reinterpret_cast<T*>(lds_raw);
    T* const lb0 = la0 + A_ELEMS;
    T* const la1 = lb0 + B_ELEMS;
    T* const lb1 = la1 + A_ELEMS;
    const RalfGemmDesc& d = P.d;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave / WGN, wn = wave % WGN;
    RALF_PROBE(0);
#ifdef RALF_GEMM_PROBE
    if (tid == 0 && blockIdx.x < 65536) {
        ralf_probe_buf[(size_t)blockIdx.x * 8 + 6] = __builtin_amdgcn_s_memrealtime();
        ralf_probe_buf[(size_t)blockIdx.x * 8 + 7] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | 4);
    }
#endif
    // PERSISTENT workgroups: the grid is at most (CUs x resident workgroups per CU); each workgroup walks over output tiles
    // bid = blockIdx.x, blockIdx.x + gridDim.x, ... and issues the first operand loads of its NEXT tile before the epilogue of
    // the current one.  (s_memtime stamps, tools/gemm_probe.hip: of a 64x64x256 tile's 9.8 k cycles, 1.3 k were index setup and
    // 2.4 k the wait for the first operand tile -- per tile, with nothing else of that workgroup to overlap it.)
    //
    // One linear index over (split, tile): all output tiles of ONE k-split get consecutive virtual ids, i.e. run together on
    // one XCD, so the operand strips they share (the same rows of dy and x in a weight gradient) are fetched from HBM once
    // and re-read from that XCD's L2.  (With the split on blockIdx.y the tiles of a split were scattered over all 8 XCDs:
    // PMC FETCH_SIZE showed 11 GB per step fetched by the weight-gradient GEMMs for 4.5 GB of operands.)  gridDim.x is a
    // multiple of 8 whenever a workgroup owns more than one tile, so all tiles of a workgroup map to its own XCD's range.
    const int total = P.nwg * d.splitk;
    const int z0 = z % d.nb0, z1 = z / d.nb0;
    const T* Ap = (const T*)d.A + z0 * d.sA0 + z1 * d.sA1;
    const T* Bp = (const T*)d.B + z0 * d.sB0 + z1 * d.sB1;
    const bool a_al = (d.lda % VEC == 0) && (((uintptr_t)Ap & 15) == 0);
    const bool b_al = (d.ldb % VEC == 0) && (((uintptr_t)Bp & 15) == 0);

    u32x4 ra0[NVA], rb0[NVB], ra1[NVA], rb1[NVB];   // two staging register sets: prefetch distance 2 k-tiles
    u32x4 rc0[NVA], rc1[NVA];                       // (operand transform: the second operand of A's vectors)
    int at_oo[NVA];                                 // (operand transform: element offset of each staged vector in A / at_a2 / at_out at k = 0)
    uint32_t at_rowok = 0;                          // bit i: vector i's row lies inside M (write-through)
    int64_t at_a2d = 0;                             // at_a2 - A in elements
    RowInfo ia[NVA], ib[NVB];
    // PMC (SQ_ACTIVE_INST_ANY ~ 84 % of the kernel, MFMA busy 10 %) showed this kernel is instruction-issue bound:
    // the interior fast path keeps one pointer per staging vector and advances it by a constant per k-tile.
    // Rows beyond M/N are clamped to the last valid row (their results are never stored).
    // GATHER 14 = GATHER 1 (tap-uniform im2col gather of A) for the DATA GRADIENT OF A STRIDE-2 CONVOLUTION, by parity classes (round 6).  Output
    // pixel (y, x) of the input grid receives tap (kh, kw) only when y + pad - kh and x + pad - kw are even: a quarter of the taps on average
    // (3 x 3: 1, 2, 2 or 4 of 9; 1 x 1: the even-even pixels only).  The plain gather multiplied every tap for every pixel and zeroed three
    // quarters of the operand vectors.  Here the rows are visited class by class -- virtual row = (class, image, y / 2, x / 2), a tile never
    // straddles classes -- and a tile walks ONLY the k-tiles of its class's taps (none at all: the zero rows of a 1 x 1 stride-2 gradient,
    // epilogue only).  Valid taps in the same order, the skipped products were exact zeros: same results.  Epilogues address memory by the
    // REAL row (par_real_row).
    constexpr bool PAR = GATHER == 14;
    constexpr bool G1 = GATHER == 1 || PAR;
    static_assert(!PAR || (AK && BKC && FM == 2 && FN == 2 && NW == 8 && (EPI == 0 || EPI == 3)), "parity form: 128 x 128 tiles, plain / BatchNorm-backward epilogues");
    // GATHER 15 = the tap gather of a 3 x 3 / stride-1 / pad-1 convolution (forward or data gradient) with the tile's INPUT PATCH RESIDENT IN LDS (round 6).
    // The tile GEMM is bound by the bytes a CU takes in per flop, and the tap gather loads a tile's A operand nine times -- once
    // per tap, shifted by a pixel.  Here a tile is BM / SW whole image rows; their (rows + 2) x (SW + 2) halo patch, all channels, is loaded ONCE
    // ([pixel][SC + 8] bf16: the odd 16-byte stride keeps the pixels of a fragment read on distinct banks; pixels outside the image are zeros), and
    // the A fragment of k-tile (tap, channel chunk) is read at a per-tap BYTE OFFSET from the same image.  Only the weights stream (direct-to-LDS
    // ring): 92 + 590 KB per 128 x 128 tile of layer3 instead of 590 + 590.  Same k-tiles in the same order, same MFMA chain: bit-identical to
    // GATHER 1.  Tiles (ralf_gemm picks, P.patch): 128 x 128 on 8 waves of 64 x 32 (layer3; layer2 at small batch), 256 x 128 on 4 x 2 waves of
    // 64 x 64 with the fragments double-buffered in registers (layer2), 256 x 64 on 4 x 2 waves of 64 x 32 with TWO workgroups per CU (layer1: nine
    // k-tiles per tile -- prologue and epilogue want company).
    // MEASURED (tools/lab/patch_lab.hip, profiles/r06_patch_lab.txt; B = 64): layer1 42.7 -> 31.5 us, layer2 30.4 -> 25.8, layer3 33.0 -> 28.5.  What
    // bounds it now (cycle stamps of layer3, 56 k cycles per tile): the patch prologue -- 95 KB per CU, every CU at once, ~10 k cycles (the chip's
    // ~11 B/clk/CU prologue burst) --, 36 k-tiles at ~1 100 cycles each against 540 of bare MFMAs, the epilogue 4-6 k.  The k-tile's cost is additive:
    // + 240 cycles for the 16 fragment reads (the LDS array's own 4 cycles per ds_read_b128 and wave), + 240 for the wait / barrier / four LDS-DMA
    // issues, whether the loop is the plain one, fenced phase by phase or woven [MFMA, read] by sched_group_barrier (all within 3 %); a deeper
    // ring (4 stages) changes nothing (not load latency), 128 x 128 on 4 waves of 64 x 64 is no faster than 8 waves of 64 x 32.
    constexpr bool PATCH = GATHER == 15;
    static_assert(!PATCH || (AK && BKC && sizeof(T) == 2 && (EPI == 0 || EPI == 3) && !RALF_GEMM_PERSISTENT), "patch form: bf16, plain / BatchNorm-backward epilogues");
    uint32_t par_taps = 0;                              // the class's taps (kh * KW + kw), four bits each, in ascending order
    constexpr bool MT = gemm_is_mt(GATHER);             // ... its pipelined form (both operands k-contiguous)
    constexpr bool GLDS = gemm_is_glds(GATHER);         // direct-to-LDS ring (bf16, interior fast path, any of the four operand layouts)
    static_assert(!MT || (AK && BKC && !CS), "pipelined ring: NT products");
    static_assert(!GLDS || sizeof(T) == 2, "direct-to-LDS: bf16 operands");
    // row-contiguous image of an operand with R = BM / BN rows: [BK k-rows][R] bf16, S = 2 R bytes per k-row (128 or 256), no padding.
    // A 1-KiB LDS-DMA covers 1024 / S k-rows with S / 16 lanes each.  The transpose read of a 32-lane half touches 4 consecutive k-rows
    // x 64 bytes; with S a multiple of 128 those rows would share their banks, so the 64-byte granule g of k-row k holds source
    // granule g ^ x(k): x = k & 3 (S = 256), (k >> 1) & 1 (S = 128) -> the four rows land in four different quarters of the 64 banks.
    constexpr int SA = BM * 2, SB = BN * 2;
    constexpr int RPCA = 1024 / SA, RPCB = 1024 / SB, LPRA = SA / 16, LPRB = SB / 16;
    auto xkey = [](int S, int k) { return S == 256 ? (k & 3) : ((k >> 1) & 1); };
    constexpr int AT = gemm_at_mode(GATHER);
    constexpr bool AT2 = GATHER == 7 || GATHER == 8;   // a second operand is staged beside A
    static_assert(!AT || (AK && sizeof(T) == 2 && !RALF_GEMM_PERSISTENT), "operand transform: bf16, k-contiguous A, one tile per workgroup");
    constexpr bool fast = GATHER == 3 || AT != 0;   // compile-time: the general loaders (and their RowInfo registers) are not even compiled in
    const T* pa[NVA];
    const T* pb[NVB];
    // LEAN GATHERS.  ISA of the first version (one RowInfo per vector, 64-bit offsets, a division per tap): 90-300 VALU instructions per
    // k-tile and wave against 4-8 MFMAs -- the 64x64 gather kernels were bound by their address arithmetic (the weight gradient of a
    // 3x3 convolution: 1.6 us per k-step).  Now all offsets are 32-bit elements (checked at launch) and
    //   gather 1: per vector (image base + channel, pixel offset, y0, x0) are fixed per tile; a k-tile's tap adds ONE scalar delta;
    //   gather 2: the column (tap, channel) is fixed per thread, the pixel (oy, ox, image offset) advances by BK rows per k-tile with
    //             carries instead of divisions.
    int g_bc[NVA], g_p0[NVA], g_y0[NVA], g_x0[NVA];            // gather 1
    int h_oy[NVB], h_ox[NVB], h_bo[NVB], h_tyo = 0, h_txo = 0, h_c = 0;   // gather 2
    bool h_cok = false;
    const int64_t stepA = AK ? BK : (int64_t)BK * d.lda, stepB = BKC ? BK : (int64_t)BK * d.ldb;
    int split, m0, n0, kbeg, kend, nt;   // the tile whose operands are being LOADED
    auto setup = [&](int bid) {
        const int vid = xcd_remap(bid, total);
        split = vid / P.nwg;
        const int tile = vid - split * P.nwg;
        int tm, tn;
        if (P.mfast) { tn = tile / P.tiles_m; tm = tile - tn * P.tiles_m; }
        else         { tm = tile / P.tiles_n; tn = tile - tm * P.tiles_n; }
        if constexpr (GATHER == 14) {
            // consecutive row tiles take turns through the four parity classes (their reductions differ: 1 / 2 / 2 / 4 taps of a 3 x 3, 1 / 0 / 0 / 0
            // of a 1 x 1): in class-major order an XCD's share of the tile ids was ONE class, i.e. two of the eight XCDs did all of a 1 x 1's work
            const int q4 = P.tiles_m >> 2;
            tm = (tm & 3) * q4 + (tm >> 2);
        }
        m0 = tm * BM; n0 = tn * BN;
        kbeg = split * P.kchunk;
        kend = min(d.K, kbeg + P.kchunk);
        nt = (kend - kbeg + BK - 1) / BK;
        if constexpr (PAR) {   // the tile's class and its taps; the reduction is taps x channels
            const RalfConvGeom& g = d.g;
            const int cls = (int)P.fd_q.div((uint32_t)m0), py = cls >> 1, px = cls & 1;
            int n = 0;
            par_taps = 0;
            for (int kh = 0; kh < g.KH; ++kh)
                for (int kw = 0; kw < g.KW; ++kw)
                    if ((((py + g.pad - kh) | (px + g.pad - kw)) & 1) == 0) { par_taps |= (uint32_t)(kh * g.KW + kw) << (4 * n); ++n; }
            kbeg = 0; kend = n * g.SC;
            nt = kend / BK;
        }
        if constexpr (G1) {
            const RalfConvGeom& g = d.g;
#pragma unroll
            for (int i = 0; i < NVA; ++i) {
                const int v = tid + NT * i;
                RowInfo r;
                if constexpr (PAR) {
                    int cls, b, ry, rx;
                    const int row = m0 + v / KV;
                    par_decompose(P, (uint32_t)row, cls, b, ry, rx);
                    r.ok = row < d.M; r.base = (int64_t)b * g.SH * g.SW * g.SC; r.y0 = ry + g.pad; r.x0 = rx + g.pad;
                } else {
                    r = row_info<true>(P, m0 + v / KV, d.M, d.lda);
                }
                g_bc[i] = (int)r.base + (v % KV) * VEC;
                g_p0[i] = (r.y0 * g.SW + r.x0) * g.SC;
                g_y0[i] = r.ok ? r.y0 : -(1 << 30);     // a row beyond M fails every bounds test
                g_x0[i] = r.x0;
            }
#pragma unroll
            for (int i = 0; i < NVB; ++i) {   // weights [N][K], K a multiple of the k-tile: a pointer per vector, rows clamped (never stored)
                const int v = tid + NT * i;
                pb[i] = Bp + (int64_t)min(n0 + v / KV, d.N - 1) * d.ldb + kbeg + (v % KV) * VEC;
            }
        }
        if constexpr (PATCH) {   // the weights' k-tiles through the direct-to-LDS ring (the swizzled [rows][64] image of the GLDS loop)
            static_assert(!PATCH || BN / 8 / NW == NVB, "chunk = one 16-byte vector per lane");
#pragma unroll
            for (int i = 0; i < NVB; ++i) {
                const int row = (wave * NVB + i) * 8 + (lane >> 3);
                pb[i] = Bp + (int64_t)min(n0 + row, d.N - 1) * d.ldb + kbeg + (((lane & 7) ^ ((row >> 1) & 7)) * VEC);
            }
        }
        if constexpr (GATHER == 2) {
            const RalfConvGeom& g = d.g;
#pragma unroll
            for (int i = 0; i < NVA; ++i) {   // dy [K][M]: a pointer per vector (rows beyond the k range are not dereferenced)
                const int v = tid + NT * i;
                pa[i] = Ap + (int64_t)(kbeg + v / RVA) * d.lda + min(m0 + (v % RVA) * VEC, d.M - VEC);
            }
            const ColInfo ci = col_info(P, n0 + (tid % RVB) * VEC, d.N);   // NT % RVB == 0: one column for all vectors of a thread
            h_tyo = ci.kh - g.pad; h_txo = ci.kw - g.pad; h_c = ci.c; h_cok = ci.ok;
#pragma unroll
            for (int i = 0; i < NVB; ++i) {
                int b, rem;
                P.fd_hw.divmod((uint32_t)(kbeg + (tid + NT * i) / RVB), b, rem);
                P.fd_rw.divmod((uint32_t)rem, h_oy[i], h_ox[i]);
                h_bo[i] = b * P.img;
            }
        }
        if constexpr (GLDS) {
            // one global_load_lds_dwordx4 moves 64 lanes x 16 bytes into 1 KiB of LDS, lane-linear.
            // k-contiguous operand: 8 tile rows of 128 bytes; lane l fills slot (l & 7) of row (l >> 3) of its chunk with k-vector
            // (slot ^ ((row >> 1) & 7)): a 16-lane group of a ds_read_b128 (32 consecutive rows, one k-vector) then touches 16 distinct
            // (row parity, slot) pairs = all 64 banks once.  The 8 lanes of a row still read one whole 128-byte line.
            // row-contiguous operand: 1024 / S k-rows of S bytes; lane l fills slot (l % (S/16)) of k-row (l / (S/16)) with the source
            // column granule permuted as described at SA / SB above (whole 64-byte pieces of one contiguous S-byte run).
            static_assert(BM / 8 / NW == NVA && BN / 8 / NW == NVB, "chunk = one 16-byte vector per lane");
#pragma unroll
            for (int i = 0; i < NVA; ++i) {
                if constexpr (AK) {
                    const int row = (wave * NVA + i) * 8 + (lane >> 3);
                    pa[i] = Ap + (int64_t)min(m0 + row, d.M - 1) * d.lda + kbeg + (((lane & 7) ^ ((row >> 1) & 7)) * VEC);
                } else {
                    const int krow = (wave * NVA + i) * RPCA + lane / LPRA, slot = lane % LPRA;
                    const int col16 = (((slot >> 2) ^ xkey(SA, krow)) << 2) | (slot & 3);
                    pa[i] = Ap + (int64_t)(kbeg + krow) * d.lda + min(m0 + col16 * VEC, d.M - VEC);
                }
            }
#pragma unroll
            for (int i = 0; i < NVB; ++i) {
                if constexpr (BKC) {
                    const int row = (wave * NVB + i) * 8 + (lane >> 3);
                    pb[i] = Bp + (int64_t)min(n0 + row, d.N - 1) * d.ldb + kbeg + (((lane & 7) ^ ((row >> 1) & 7)) * VEC);
                } else {
                    const int krow = (wave * NVB + i) * RPCB + lane / LPRB, slot = lane % LPRB;
                    const int col16 = (((slot >> 2) ^ xkey(SB, krow)) << 2) | (slot & 3);
                    pb[i] = Bp + (int64_t)(kbeg + krow) * d.ldb + min(n0 + col16 * VEC, d.N - VEC);
                }
            }
        }
        if (fast) {
#pragma unroll
            for (int i = 0; i < NVA; ++i) {
                const int v = tid + NT * i;
                if (AK) pa[i] = Ap + (int64_t)min(m0 + v / KV, d.M - 1) * d.lda + kbeg + (v % KV) * VEC;
                else pa[i] = Ap + (int64_t)(kbeg + v / RVA) * d.lda + min(m0 + (v % RVA) * VEC, d.M - VEC);
            }
#pragma unroll
            for (int i = 0; i < NVB; ++i) {
                const int v = tid + NT * i;
                if (BKC) pb[i] = Bp + (int64_t)min(n0 + v / KV, d.N - 1) * d.ldb + kbeg + (v % KV) * VEC;
                else pb[i] = Bp + (int64_t)(kbeg + v / RVB) * d.ldb + min(n0 + (v % RVB) * VEC, d.N - VEC);
                if (d.kseg) pb[i] += (int64_t)(kbeg / d.kseg) * d.sBk;   // segmented K range of B (kbeg is a multiple of the k-tile)
            }
        }
        if constexpr (AT != 0) {
            at_a2d = AT2 ? ((const T*)d.at_a2 - (const T*)d.A) : 0;
#pragma unroll
            for (int i = 0; i < NVA; ++i) {
                const int v = tid + NT * i, row = m0 + v / KV;
                at_oo[i] = min(row, d.M - 1) * (int)d.lda + (v % KV) * VEC;   // (M * lda < 2^31: checked at launch)
                at_rowok |= (uint32_t)(row < d.M) << i;
            }
        }
        constexpr bool generic = GATHER == 0 || GATHER == 4;   // (GLDS: neither)
        if (AK && generic) {
#pragma unroll
            for (int i = 0; i < NVA; ++i) ia[i] = row_info<GATHER == 4>(P, m0 + (tid + NT * i) / KV, d.M, d.lda);
        }
        if (BKC && generic) {
#pragma unroll
            for (int i = 0; i < NVB; ++i) ib[i] = row_info<false>(P, n0 + (tid + NT * i) / KV, d.N, d.ldb);
        }
    };
    uint32_t okm0 = ~0u, okm1 = ~0u;   // validity bits of the staged vectors of set 0 / 1 (A: bits 0.., B: bits 16..): zeroed at stage time
    auto gload = [&](u32x4 (&ra)[NVA], u32x4 (&rb)[NVB], u32x4 (&rc)[NVA], int k0, uint32_t& okm) {
        (void)rc;
        if constexpr (GLDS || PATCH) {
            (void)ra; (void)rb; (void)k0; (void)okm;   // (the direct-to-LDS loops below issue their own loads)
        } else if constexpr (fast) {
            (void)okm;
            if constexpr (AT2) {
#pragma unroll
                for (int i = 0; i < NVA; ++i) rc[i] = *reinterpret_cast<const u32x4*>(pa[i] + at_a2d);
            }
#pragma unroll
            for (int i = 0; i < NVA; ++i) { ra[i] = *reinterpret_cast<const u32x4*>(pa[i]); pa[i] += stepA; }
#pragma unroll
            for (int i = 0; i < NVB; ++i) { rb[i] = *reinterpret_cast<const u32x4*>(pb[i]); pb[i] += stepB; }
            if (d.kseg && (k0 + BK) % d.kseg == 0) {   // (wave-uniform) the next k-tile opens a new segment of B
#pragma unroll
                for (int i = 0; i < NVB; ++i) pb[i] += d.sBk;
            }
        } else if constexpr (G1) {
            // every k-tile lies inside ONE tap (channels % BK == 0): its (kh, kw, c0) come from the scalar unit
            const RalfConvGeom& g = d.g;
            int t, c0, kh, kw;
            P.fd_sc.divmod((uint32_t)k0, t, c0);
            if constexpr (PAR) t = (int)((par_taps >> (4 * t)) & 15u);   // k0 counts the class's taps: the t-th of them
            P.fd_kw.divmod((uint32_t)t, kh, kw);
            const int skh = P.ln_sgn * kh, skw = P.ln_sgn * kw;
            const int delta = (skh * g.SW + skw) * g.SC;
            uint32_t m = 0xffff0000u;
#pragma unroll
            for (int i = 0; i < NVA; ++i) {
                // source pixel = ((y0 + skh) >> sh, (x0 + skw) >> sh), valid when both shifts are exact (data gradient of a strided
                // convolution) and inside the image; its offset (y*SW + x)*SC = (p0 + delta) >> sh for exactly those
                const int ty = g_y0[i] + skh, tx = g_x0[i] + skw;
                const bool ok = (((ty | tx) & P.ln_pm) == 0) & ((uint32_t)(ty >> P.ln_sh) < (uint32_t)g.SH) & ((uint32_t)(tx >> P.ln_sh) < (uint32_t)g.SW);
                const int off = g_bc[i] + ((g_p0[i] + delta) >> P.ln_sh) + c0;
                ra[i] = *reinterpret_cast<const u32x4*>(Ap + (ok ? off : 0));
                m |= (uint32_t)ok << i;
            }
            if constexpr (PAR) {
                const int kreal = t * g.SC + c0;                         // the weights' k range of this tile (uniform)
#pragma unroll
                for (int i = 0; i < NVB; ++i) rb[i] = *reinterpret_cast<const u32x4*>(pb[i] + kreal);
            } else {
#pragma unroll
                for (int i = 0; i < NVB; ++i) { rb[i] = *reinterpret_cast<const u32x4*>(pb[i]); pb[i] += BK; }
            }
            okm = m;
        } else if constexpr (GATHER == 2) {
            const RalfConvGeom& g = d.g;
            const int kleft = kend - k0;   // (uniform) rows of the k range still ahead
            uint32_t m = 0u;
#pragma unroll
            for (int i = 0; i < NVA; ++i) {
                const bool ok = (tid + NT * i) / RVA < kleft;
                ra[i] = *reinterpret_cast<const u32x4*>(ok ? pa[i] : Ap);
                pa[i] += stepA;
                m |= (uint32_t)ok << i;
            }
#pragma unroll
            for (int i = 0; i < NVB; ++i) {
                const int ty = __mul24(h_oy[i], g.stride) + h_tyo, tx = __mul24(h_ox[i], g.stride) + h_txo;
                const bool ok = ((tid + NT * i) / RVB < kleft) & h_cok & ((uint32_t)ty < (uint32_t)g.SH) & ((uint32_t)tx < (uint32_t)g.SW);
                const int off = h_bo[i] + __mul24(ty, P.swsc) + __mul24(tx, g.SC) + h_c;
                rb[i] = *reinterpret_cast<const u32x4*>(Bp + (ok ? off : 0));
                m |= (uint32_t)ok << (16 + i);
                // the pixel BK rows further on (one carry each: inc_x < RW, inc_y < RH)
                int ox = h_ox[i] + P.inc_x;
                const int wx = ox >= g.RW;
                h_ox[i] = ox - (wx ? g.RW : 0);
                int oy = h_oy[i] + P.inc_y + wx;
                const int wy = oy >= g.RH;
                h_oy[i] = oy - (wy ? g.RH : 0);
                h_bo[i] += P.inc_b + (wy ? P.img : 0);
            }
            okm = m;
        } else {
        // GATHER 4: the general gather (stem: 8 channels), per vector; GATHER 0: plain operands with edges
        uint32_t m = 0u;
#pragma unroll
        for (int i = 0; i < NVA; ++i) {
            const int v = tid + NT * i;
            bool ok;
            if (AK) {
                ra[i] = load_vec<T, GATHER == 4>(Ap, P, ia[i], k0 + (v % KV) * VEC, kend, a_al, ok);
            } else {
                const RowInfo r = row_info<false>(P, k0 + v / RVA, kend, d.lda);
                ra[i] = load_vec<T, false>(Ap, P, r, m0 + (v % RVA) * VEC, d.M, a_al, ok);
            }
            m |= (uint32_t)ok << i;
        }
#pragma unroll
        for (int i = 0; i < NVB; ++i) {
            const int v = tid + NT * i;
            bool ok;
            if (BKC) {
                if constexpr (GATHER == 4) rb[i] = load_vec_al<T>(Bp, ib[i], k0 + (v % KV) * VEC, kend, ok);
                else rb[i] = load_vec<T, false>(Bp, P, ib[i], k0 + (v % KV) * VEC, kend, b_al, ok);
            }
            else {
                const RowInfo r = row_info<false>(P, k0 + v / RVB, kend, d.ldb);
                rb[i] = load_vec<T, false>(Bp, P, r, n0 + (v % RVB) * VEC, d.N, b_al, ok);
            }
            m |= (uint32_t)ok << (16 + i);
        }
        okm = m;
        }
    };
    f32x16 acc[WFM][WFN];

    const int l31 = lane & 31, lh = lane >> 5;
    // transpose-read geometry (bf16 row-contiguous tiles): 16-lane group gi reads a [4 k][16 rows] block
    const int trL = lane & 15, tr_rowblk = ((lane >> 4) & 1) * 16, tr_k = lh * 8 + (trL >> 2), tr_c = (trL & 3) * 4;

    // direct-to-LDS images: this lane's read offsets (bytes from the stage's A / B base) for k-slice 0
    const unsigned gl_sw = (unsigned)((l31 >> 1) & 7) ^ (unsigned)lh;
    const unsigned gl_offa = (unsigned)(wm * 32 * WFM + l31) * 128u + (gl_sw << 4), gl_offb = (unsigned)(wn * 32 * WFN + l31) * 128u + (gl_sw << 4);
    // ... and, for row-contiguous images, the transpose-read offsets per fragment (k-slice 0)
    unsigned gl_tra[WFM], gl_trb[WFN];
#pragma unroll
    for (int i = 0; i < WFM; ++i) gl_tra[i] = (unsigned)(tr_k * SA) + (((unsigned)(wm * WFM + i) * 64u) ^ ((unsigned)xkey(SA, tr_k) << 6)) + (unsigned)(tr_rowblk + tr_c) * 2u;
#pragma unroll
    for (int j = 0; j < WFN; ++j) gl_trb[j] = (unsigned)(tr_k * SB) + (((unsigned)(wn * WFN + j) * 64u) ^ ((unsigned)xkey(SB, tr_k) << 6)) + (unsigned)(tr_rowblk + tr_c) * 2u;
    // one k-tile of MFMAs from the staged LDS tile
    // CS: the waves that own the tile's first column block (wn == 0, first column tile, when P.colsum is set) also sum the A fragments they read
    // over k: sum_k A[k][m] = the bias gradient that goes with dW = dy^T x.  4 v_dot2_f32_bf16 per fragment (against a vector of ones) beside
    // its 32-cycle MFMA; the separate column-sum launch re-read every dy from memory (0.19 ms per encoder-decoder step).
    bool cs_on = false;
    float csacc[FM];
    auto compute = [&](const T* la, const T* lb) {
        // the matrix core computes the TRANSPOSED tile: row operand = n-fragment (weights), column operand =
        // m-fragment, so accumulator register r of a lane holds (n = (r&3) + 8*(r>>2) + 4*(lane>>5), m = lane&31)
        if constexpr (sizeof(T) == 4) {
#pragma unroll
            for (int ks = 0; ks < BK / 2; ++ks) {
                const int kk = ks * 2 + lh;
                float a[WFM], b[WFN];
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    const int r = wm * 32 * WFM + i * 32 + l31;
                    a[i] = AK ? ((float*)la)[r * 32 + ((kk + r) & 31)] : ((float*)la)[kk * LDRA + r];
                }
#pragma unroll
                for (int j = 0; j < WFN; ++j) {
                    const int r = wn * 32 * WFN + j * 32 + l31;
                    b[j] = BKC ? ((float*)lb)[r * 32 + ((kk + r) & 31)] : ((float*)lb)[kk * LDRB + r];
                }
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < WFN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[j], a[i], acc[i][j], 0, 0, 0);
            }
        } else if constexpr (GLDS) {
            // k-contiguous: swizzled [rows][64] images, k-vector kv of row r in slot kv ^ ((r >> 1) & 7); with kv = 2 ks + lh that is byte
            // offset (r * 128 + ((lh ^ sw) << 4)) ^ (ks << 5): ONE xor per k-slice and operand, fragments 32 rows apart by immediate offsets.
            // row-contiguous: transpose reads at k-row (16 ks + tr_k [+ 4]), column bytes (fragment base ^ (x << 6)) + lane part.
            const unsigned char* pa8 = reinterpret_cast<const unsigned char*>(la);
            const unsigned char* pb8 = reinterpret_cast<const unsigned char*>(lb);
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                bf16x8 a[WFM], b[WFN];
                const unsigned oa = gl_offa ^ (unsigned)(ks << 5), ob = gl_offb ^ (unsigned)(ks << 5);
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    if constexpr (AK) a[i] = *reinterpret_cast<const bf16x8*>(pa8 + oa + i * 32 * 128);
                    else {
                        const unsigned char* q = pa8 + ks * 16 * SA + gl_tra[i];
                        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, q));
                        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, q + 4 * SA));
                        a[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    }
                }
                if constexpr (CS) {
                    if (cs_on) {   // (wave-uniform)
                        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                        const bf16x2 ones = {(__bf16)1.f, (__bf16)1.f};
#pragma unroll
                        for (int i = 0; i < FM; ++i) {
                            csacc[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a[i], a[i], 0, 1), ones, csacc[i], false);
                            csacc[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a[i], a[i], 2, 3), ones, csacc[i], false);
                            csacc[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a[i], a[i], 4, 5), ones, csacc[i], false);
                            csacc[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a[i], a[i], 6, 7), ones, csacc[i], false);
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < WFN; ++j) {
                    if constexpr (BKC) b[j] = *reinterpret_cast<const bf16x8*>(pb8 + ob + j * 32 * 128);
                    else {
                        const unsigned char* q = pb8 + ks * 16 * SB + gl_trb[j];
                        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, q));
                        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, q + 4 * SB));
                        b[j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    }
                }
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < WFN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                bf16x8 a[WFM], b[WFN];
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    if (AK) {
                        a[i] = *reinterpret_cast<const bf16x8*>(la + (wm * 32 * WFM + i * 32 + l31) * X::LDK + ks * 16 + lh * 8);
                    } else {
                        const bf16* q = (const bf16*)la + (ks * 16 + tr_k) * LDRA + wm * 32 * WFM + i * 32 + tr_rowblk + tr_c;
                        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, q));
                        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, q + 4 * LDRA));
                        a[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    }
                }
#pragma unroll
                for (int j = 0; j < WFN; ++j) {
                    if (BKC) {
                        b[j] = *reinterpret_cast<const bf16x8*>(lb + (wn * 32 * WFN + j * 32 + l31) * X::LDK + ks * 16 + lh * 8);
                    } else {
                        const bf16* q = (const bf16*)lb + (ks * 16 + tr_k) * LDRB + wn * 32 * WFN + j * 32 + tr_rowblk + tr_c;
                        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, q));
                        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, q + 4 * LDRB));
                        b[j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    }
                }
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < WFN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
            }
        }
    };
    // Software pipeline: prefetch distance TWO k-tiles through two staging register sets (a tile's global loads get two compute
    // phases to land: the k-loop is bound by load latency x tiles in flight per CU, not by MFMA rate) and TWO LDS buffers, so a
    // step is  [registers of tile t+1 -> the other buffer | request tile t+3 | MFMAs of tile t | ONE barrier]  -- the LDS writes of
    // the next tile overlap the matrix work of this one.  (One buffer with compute / barrier / stage / barrier per tile ran the
    // 256-tile launches -- one workgroup per CU, nothing to interleave with -- at 1.2 us per 128x128x64 step, 6x the MFMA time.)
    // Steady state is branch-free (an `if` around a prefetch made the compiler shuttle every accumulator
    // AGPR -> VGPR -> AGPR per iteration); the last 1-4 tiles are peeled.
    auto stage = [&](T* la, T* lb, u32x4 (&ra)[NVA], u32x4 (&rb)[NVB], u32x4 (&rc)[NVA], int k0, uint32_t okm) {
        (void)rc; (void)k0;
        if constexpr (AT != 0) {
            // this thread's 8 channels of the tile (the same for all of its vectors: NT % KV == 0); coefficients from the LDS table
            constexpr int OPS = gemm_lds_bytes<T, AK, BKC, FM, FN, gemm_nbuf<GATHER, FM>(), false>();
            const float* lc = reinterpret_cast<const float*>(lds_raw + OPS) + k0 + (tid % KV) * VEC;
            float c1[8], c2[8], c3[8];
            { const float4 a = *reinterpret_cast<const float4*>(lc), b = *reinterpret_cast<const float4*>(lc + 4);
              c1[0] = a.x; c1[1] = a.y; c1[2] = a.z; c1[3] = a.w; c1[4] = b.x; c1[5] = b.y; c1[6] = b.z; c1[7] = b.w; }
            { const float4 a = *reinterpret_cast<const float4*>(lc + AT_KMAX), b = *reinterpret_cast<const float4*>(lc + AT_KMAX + 4);
              c2[0] = a.x; c2[1] = a.y; c2[2] = a.z; c2[3] = a.w; c2[4] = b.x; c2[5] = b.y; c2[6] = b.z; c2[7] = b.w; }
            if constexpr (AT == 2) {
                const float4 a = *reinterpret_cast<const float4*>(lc + 2 * AT_KMAX), b = *reinterpret_cast<const float4*>(lc + 2 * AT_KMAX + 4);
                c3[0] = a.x; c3[1] = a.y; c3[2] = a.z; c3[3] = a.w; c3[4] = b.x; c3[5] = b.y; c3[6] = b.z; c3[7] = b.w;
            }
            const bool wr = n0 == 0 && d.at_out != nullptr;   // the first column tile leaves the transformed operand (and its ReLU bits) in memory
#pragma unroll
            for (int i = 0; i < NVA; ++i) {
                const bf16x8 xv = *reinterpret_cast<const bf16x8*>(&ra[i]);
                bf16x8 o;
                uint32_t bits = 0;
                if constexpr (AT == 1) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        float v = __builtin_fmaf((float)xv[q], c1[q], c2[q]);
                        if constexpr (AT2) v += (float)(*reinterpret_cast<const bf16x8*>(&rc[i]))[q];
                        if (d.at_relu) { bits |= (v > 0.f ? 1u : 0u) << q; v = fmaxf(v, 0.f); }
                        o[q] = (bf16)v;
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        o[q] = (bf16)__builtin_fmaf((float)xv[q], c1[q], __builtin_fmaf((float)(*reinterpret_cast<const bf16x8*>(&rc[i]))[q], c2[q], c3[q]));
                }
                ra[i] = *reinterpret_cast<const u32x4*>(&o);
                if (wr && ((at_rowok >> i) & 1u)) {
                    const int e = at_oo[i] + k0;
                    *reinterpret_cast<u32x4*>((T*)d.at_out + e) = ra[i];
                    if (AT == 1 && d.at_relu && d.at_mask) d.at_mask[e >> 3] = (unsigned char)bits;
                }
            }
        }
        if constexpr (!fast) {
#pragma unroll
            for (int i = 0; i < NVA; ++i) ra[i] = sel_vec((okm >> i) & 1u, ra[i]);
#pragma unroll
            for (int i = 0; i < NVB; ++i) rb[i] = sel_vec((okm >> (16 + i)) & 1u, rb[i]);
        }
        lds_stage<T, AK, NVA, RVA, LDRA, NT>(la, ra, tid);
        lds_stage<T, BKC, NVB, RVB, LDRB, NT>(lb, rb, tid);
    };
    // (the 64x64 im2col-gather kernels keep distance 1: their index registers + a second staging set cost a wave of occupancy
    //  and the layer1 3x3 convolutions got 20 % slower with distance 2)
    constexpr int PF = ((GATHER == 1 || GATHER == 4) && FM == 1) ? 1 : 2;
    RALF_PROBE(1);
    setup(bid0);
    gload(ra0, rb0, rc0, kbeg, okm0);
    if (PF == 2 && nt > 1) gload(ra1, rb1, rc1, kbeg + BK, okm1);
    if constexpr (AT != 0) {   // per-channel coefficients -> LDS (behind the first tiles' loads: no latency of their own)
        constexpr int OPS = gemm_lds_bytes<T, AK, BKC, FM, FN, gemm_nbuf<GATHER, FM>(), false>();
        float* lc = reinterpret_cast<float*>(lds_raw + OPS);
        for (int k = tid; k < d.K; k += NT) {
            lc[k] = d.at_c1[k];
            lc[AT_KMAX + k] = d.at_c2[k];
            if constexpr (AT == 2) lc[2 * AT_KMAX + k] = d.at_c3[k];
        }
        __syncthreads();
    }
    for (int bid = bid0;;) {
    const int c_m0 = m0, c_n0 = n0, c_split = split, c_kbeg = kbeg, c_nt = nt;   // the tile being COMPUTED
    if constexpr (CS) {
        cs_on = P.colsum != nullptr && c_n0 == 0 && wn == 0;
#pragma unroll
        for (int i = 0; i < FM; ++i) csacc[i] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < WFN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    if constexpr (MT) {
        // PIPELINED RING (round 6).  The direct-to-LDS ring below with three changes that take the LDS round trip and the
        // barrier out of the matrix pipe's way:
        //   * wave tiles of 64 x 64 (WFM = WFN = 2): 4 fragment reads per 4 MFMAs instead of 3 per 2 (the LDS carried ~190 of its 256 B/clk
        //     for fragment reads alone at the 64 x 32 tiles' full MFMA rate);
        //   * fragments double-buffered in registers: k-slice s + 1 is read while slice s multiplies (an in-order wave otherwise waits
        //     ~64-128 cycles of ds_read latency per slice with the matrix pipe idle);
        //   * the tile's one barrier sits in front of its LAST k-slice: the wave that arrives has 4 MFMAs (128 cycles of its SIMD's pipe)
        //     still to issue behind it, the first fragments of the next tile are requested right behind the barrier, and the stage that was
        //     just read is re-filled while those MFMAs run.
        // Ring of NST stages, NST - 2 tiles in flight across every barrier (counted vmcnt, raw s_barrier).  Same MFMA chain in the same
        // order as every other bf16 path: bit-identical results.
        // MEASURED (tools/gemm_lab.hip, profiles/r06_gemm_lab_*.txt): NOT faster than the plain ring on any shape of the model -- 128 x 128 on 4 waves
        // 722 vs 802 TFLOP/s (16384 x 256 x 2304), 256 x 256 on 8 waves 1182 vs 1167 (8192^3).  PMC: the LDS is 21 % busy, bank conflicts 0, MFMA pipes 27 %
        // busy, waves parked on vmcnt / the barrier half of their life: the loop waits for OPERANDS (a CU takes in ~20-23 B/clk with every tile
        // shape and ring depth), not for fragment reads.  Kept as lab variants (GATHER 10-13 are not dispatched by launch_cfg).
        static_assert(!RALF_GEMM_PERSISTENT, "the direct-to-LDS loop is one tile per workgroup");
        static_assert(BK == 64, "four 16-wide k-slices per tile");
        constexpr int NST = gemm_nbuf<GATHER, FM>();
        constexpr int STAGE = (BM + BN) * BK, NLD = NVA + NVB;
        const unsigned char* const ring = reinterpret_cast<const unsigned char*>(la0);
        auto issue = [&](int st) {
            T* sa = la0 + st * STAGE + wave * (NVA * 512);
            T* sb = la0 + st * STAGE + BM * BK + wave * (NVB * 512);
#pragma unroll
            for (int i = 0; i < NVA; ++i) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pa[i], LDS_PTR(void, sa + i * 512), 16, 0, 0);
                pa[i] += stepA;
            }
#pragma unroll
            for (int i = 0; i < NVB; ++i) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pb[i], LDS_PTR(void, sb + i * 512), 16, 0, 0);
                pb[i] += stepB;
            }
        };
        bf16x8 fa0[WFM], fb0[WFN], fa1[WFM], fb1[WFN];
        auto rd = [&](bf16x8 (&fa)[WFM], bf16x8 (&fb)[WFN], int st, int ks) {   // the fragments of k-slice ks of stage st
            const unsigned char* base = ring + (size_t)st * STAGE * 2;
            const unsigned oa = gl_offa ^ (unsigned)(ks << 5), ob = (unsigned)(BM * BK * 2) + (gl_offb ^ (unsigned)(ks << 5));
#pragma unroll
            for (int i = 0; i < WFM; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(base + oa + i * 32 * 128);
#pragma unroll
            for (int j = 0; j < WFN; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(base + ob + j * 32 * 128);
        };
        auto mm = [&](const bf16x8 (&fa)[WFM], const bf16x8 (&fb)[WFN]) {
#pragma unroll
            for (int i = 0; i < WFM; ++i)
#pragma unroll
                for (int j = 0; j < WFN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        };
        // prologue: tiles 0 .. NST-1 on their way, tile 0 landed, its first fragments in registers
#pragma unroll
        for (int q = 0; q < NST; ++q)
            if (q < c_nt) issue(q);
        {
            const int ahead = min(c_nt, NST) - 1;   // tiles that may stay in flight (wave-uniform)
            if (ahead >= 3) vm_wait<NLD * 3>(); else if (ahead == 2) vm_wait<NLD * 2>(); else if (ahead == 1) vm_wait<NLD>(); else vm_wait<0>();
        }
        lds_barrier();
        rd(fa0, fb0, 0, 0);
        int st = 0, t = 0;
        // one k-tile: slices 0..2 with the next slice's reads in front of each, then [wait for tile t+1 | barrier | first reads of tile t+1 |
        // slice 3 | re-fill of this tile's stage]
#define RALF_MT_TILE(WAIT, REFILL)                                                                      \
        {                                                                                               \
            const int sn = (st + 1 == NST) ? 0 : st + 1;                                                \
            rd(fa1, fb1, st, 1);                                                                        \
            mm(fa0, fb0);                                                                               \
            rd(fa0, fb0, st, 2);                                                                        \
            mm(fa1, fb1);                                                                               \
            rd(fa1, fb1, st, 3);                                                                        \
            mm(fa0, fb0);                                                                               \
            WAIT;                                                                                       \
            lds_barrier();                                                                              \
            rd(fa0, fb0, sn, 0);                                                                        \
            mm(fa1, fb1);                                                                               \
            REFILL;                                                                                     \
            st = sn;                                                                                    \
        }
        for (; t + NST < c_nt; ++t) RALF_MT_TILE(vm_wait<NLD * (NST - 2)>(), issue(st))   // steady state: tile t+NST follows tile t into its stage
        for (; t + 1 < c_nt; ++t) {                                                      // drain: nothing left to issue
            const int ahead = c_nt - t - 2;   // tiles behind t+1 that may stay in flight
            if (NST >= 4 && ahead >= 2) RALF_MT_TILE(vm_wait<NLD * 2>(), (void)0)
            else if (NST >= 3 && ahead >= 1) RALF_MT_TILE(vm_wait<NLD>(), (void)0)
            else RALF_MT_TILE(vm_wait<0>(), (void)0)
        }
#undef RALF_MT_TILE
        // last tile: its slice-0 fragments are in set 0
        rd(fa1, fb1, st, 1);
        mm(fa0, fb0);
        rd(fa0, fb0, st, 2);
        mm(fa1, fb1);
        rd(fa1, fb1, st, 3);
        mm(fa0, fb0);
        mm(fa1, fb1);
    } else if constexpr (PATCH) {
        const RalfConvGeom& g = d.g;
        constexpr int NST = gemm_patch_nst(FN), STAGE = BN * BK;   // (elements)
        unsigned char* const patch = lds_raw + gemm_patch_ring(FN);
        auto issue = [&](int st) {
            T* sb = la0 + st * STAGE + wave * (NVB * 512);
#pragma unroll
            for (int i = 0; i < NVB; ++i) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pb[i], LDS_PTR(void, sb + i * 512), 16, 0, 0);
                pb[i] += BK;
            }
        };
#pragma unroll
        for (int q = 0; q < NST; ++q) issue(q);     // (9 SC / 64 >= 9 k-tiles)
        {   // the patch: pieces of 16 bytes, piece v = (pixel, channel octet); image rows y0t - 1 .. y0t + BM / SW, columns -1 .. SW
            int bimg, rem;
            P.fd_hw.divmod((uint32_t)c_m0, bimg, rem);
            const int y0t = rem >> P.p_swsh;
            const int c8m = (1 << P.p_c8sh) - 1, npieces = (((BM >> P.p_swsh) + 2) * P.p_pw) << P.p_c8sh;
            const T* img = Ap + (int64_t)bimg * P.img;
            constexpr int PU = 8;
            for (int v0 = tid; v0 < npieces; v0 += PU * NT) {
                u32x4 r[PU];
                unsigned off[PU];
                uint32_t okv = 0;
#pragma unroll
                for (int u = 0; u < PU; ++u) {
                    const int v = v0 + u * NT;
                    const int pix = v >> P.p_c8sh, piece = v & c8m;
                    int py, px;
                    P.fd_pw.divmod((uint32_t)pix, py, px);
                    const int sy = y0t + py - 1, sx = px - 1;
                    const bool in = (v < npieces) & ((uint32_t)sy < (uint32_t)g.SH) & ((uint32_t)sx < (uint32_t)g.SW);
                    r[u] = *reinterpret_cast<const u32x4*>(img + (in ? (sy * g.SW + sx) * g.SC + piece * 8 : 0));
                    okv |= (uint32_t)in << u;
                    off[u] = (unsigned)(pix * P.p_str + piece * 16);
                }
#pragma unroll
                for (int u = 0; u < PU; ++u)
                    if (v0 + u * NT < npieces) *reinterpret_cast<u32x4*>(patch + off[u]) = sel_vec((okv >> u) & 1u, r[u]);
            }
        }
        // this lane's pixel of each fragment (byte offset in the patch at tap (0, 0), channel 0) and the k-half it reads
        unsigned pa_off[WFM];
#pragma unroll
        for (int i = 0; i < WFM; ++i) {
            const int r = wm * 32 * WFM + i * 32 + l31;
            pa_off[i] = (unsigned)(((r >> P.p_swsh) * P.p_pw + (r & (g.SW - 1))) * P.p_str + lh * 16);
        }
        const unsigned char* const ring8 = reinterpret_cast<const unsigned char*>(la0);
        auto tapbase = [&](int t) -> const unsigned char* {   // the patch at k-tile t's (tap, channel chunk): a scalar byte offset
            int tap, c0, kh, kw;
            P.fd_sc.divmod((uint32_t)(t * BK), tap, c0);
            P.fd_kw.divmod((uint32_t)tap, kh, kw);
            if (g.mode) { kh = 2 - kh; kw = 2 - kw; }   // data gradient: source pixel (y + 1 - kh, x + 1 - kw)
            return patch + (kh * P.p_pw + kw) * P.p_str + c0 * 2;
        };
#ifndef RALF_PATCH_ABL
#define RALF_PATCH_ABL 0   // (lab: ablation bits -- 1 no MFMA, 2 no A reads, 4 no B reads, 8 no waits / barriers / refills; results are wrong then)
#endif
        auto rd = [&](bf16x8 (&fa)[WFM], bf16x8 (&fb)[WFN], const unsigned char* pa8, int st, int ks) {
            const unsigned char* pb8 = ring8 + st * (STAGE * 2);
            const unsigned ob = gl_offb ^ (unsigned)(ks << 5);
            // (in the order the MFMAs want them: the first product needs fa[0] and fb[0])
            if constexpr (!(RALF_PATCH_ABL & 2)) fa[0] = *reinterpret_cast<const bf16x8*>(pa8 + pa_off[0] + ks * 32);
            if constexpr (!(RALF_PATCH_ABL & 4)) {
#pragma unroll
                for (int j = 0; j < WFN; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(pb8 + ob + j * 32 * 128);
            }
            if constexpr (!(RALF_PATCH_ABL & 2)) {
#pragma unroll
                for (int i = 1; i < WFM; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(pa8 + pa_off[i] + ks * 32);
            }
        };
        auto mm = [&](const bf16x8 (&fa)[WFM], const bf16x8 (&fb)[WFN]) {
            if constexpr (RALF_PATCH_ABL & 1) {
#pragma unroll
                for (int i = 0; i < WFM; ++i)
#pragma unroll
                    for (int j = 0; j < WFN; ++j) asm volatile("" :: "v"(fa[i]), "v"(fb[j]));
            } else {
#pragma unroll
            for (int i = 0; i < WFM; ++i)
#pragma unroll
                for (int j = 0; j < WFN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
            }
        };
        __syncthreads();                            // (drains the ring's first three stages and the patch writes)
        RALF_PROBE(2);
        if constexpr (WFM * WFN >= 4) {
            // 64 x 64 wave tiles (four fragment reads per four MFMAs): fragments double-buffered in registers, the k-tile's one barrier in front of
            // its LAST k-slice (the pipelined ring of GATHER 10-13, which did not pay while the loop waited for operands; here it does not)
            bf16x8 fa0[WFM] = {}, fb0[WFN] = {}, fa1[WFM] = {}, fb1[WFN] = {};
            const unsigned char* pa_c = tapbase(0);
            rd(fa0, fb0, pa_c, 0, 0);
            int st = 0, t = 0;
            // (the scheduler is fenced at every phase boundary: left alone it pulls each fragment read down to the MFMA that consumes it -- an
            //  `s_waitcnt lgkmcnt(0)` in front of nearly every MFMA, nothing overlapped: 40.5 k cycles per 36 k-tiles against 19.4 k of bare MFMAs)
#define RALF_PT_FENCE __builtin_amdgcn_sched_barrier(0)
            // one phase = the NEXT k-slice's WFM + WFN fragment reads woven into this slice's MFMAs: [MFMA, read] pairs, the rest of the MFMAs behind
#define RALF_PT_WEAVE                                                                                   \
            _Pragma("unroll") for (int q = 0; q < WFM + WFN; ++q) {                                    \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                      \
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                      \
            }                                                                                           \
            __builtin_amdgcn_sched_group_barrier(0x008, WFM * WFN - (WFM + WFN) > 0 ? WFM * WFN - (WFM + WFN) : 0, 0);
#define RALF_PT_TILE(WAIT, REFILL)                                                                      \
            {                                                                                           \
                const int sn = (st + 1 == NST) ? 0 : st + 1;                                            \
                const unsigned char* pa_n = tapbase(t + 1);                                             \
                RALF_PT_FENCE;                                                                          \
                rd(fa1, fb1, pa_c, st, 1);                                                              \
                mm(fa0, fb0);                                                                           \
                RALF_PT_WEAVE                                                                           \
                RALF_PT_FENCE;                                                                          \
                rd(fa0, fb0, pa_c, st, 2);                                                              \
                mm(fa1, fb1);                                                                           \
                RALF_PT_WEAVE                                                                           \
                RALF_PT_FENCE;                                                                          \
                rd(fa1, fb1, pa_c, st, 3);                                                              \
                mm(fa0, fb0);                                                                           \
                RALF_PT_WEAVE                                                                           \
                RALF_PT_FENCE;                                                                          \
                WAIT;                                                                                   \
                if constexpr (!(RALF_PATCH_ABL & 8)) lds_barrier();                                     \
                RALF_PT_FENCE;                                                                          \
                rd(fa0, fb0, pa_n, sn, 0);                                                              \
                mm(fa1, fb1);                                                                           \
                REFILL;                                                                                 \
                _Pragma("unroll") for (int q = 0; q < WFM + WFN; ++q) {                                 \
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                  \
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                  \
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                  \
                }                                                                                       \
                RALF_PT_FENCE;                                                                          \
                st = sn; pa_c = pa_n;                                                                   \
            }
            if constexpr (RALF_PATCH_ABL & 8) { for (; t + 1 < c_nt; ++t) RALF_PT_TILE((void)0, (void)0) }
            for (; t + NST < c_nt; ++t) RALF_PT_TILE(vm_wait<NVB * (NST - 2)>(), issue(st))
            for (; t + 1 < c_nt; ++t) {
                const int ahead = c_nt - t - 2;   // tiles behind t + 1 that may stay in flight
                if (NST >= 5 && ahead >= 3) RALF_PT_TILE(vm_wait<NVB * 3>(), (void)0)
                else if (NST >= 4 && ahead >= 2) RALF_PT_TILE(vm_wait<NVB * 2>(), (void)0)
                else if (ahead >= 1) RALF_PT_TILE(vm_wait<NVB>(), (void)0)
                else RALF_PT_TILE(vm_wait<0>(), (void)0)
            }
#undef RALF_PT_TILE
            RALF_PT_FENCE;
            rd(fa1, fb1, pa_c, st, 1);
            mm(fa0, fb0);
            RALF_PT_WEAVE
            RALF_PT_FENCE;
            rd(fa0, fb0, pa_c, st, 2);
            mm(fa1, fb1);
            RALF_PT_WEAVE
            RALF_PT_FENCE;
            rd(fa1, fb1, pa_c, st, 3);
            mm(fa0, fb0);
            RALF_PT_WEAVE
            RALF_PT_FENCE;
            mm(fa1, fb1);
#undef RALF_PT_FENCE
#undef RALF_PT_WEAVE
        } else {
            auto compute_patch = [&](int t, int st) {
                const unsigned char* pa8 = tapbase(t);
#pragma unroll
                for (int ks = 0; ks < BK / 16; ++ks) {
                    bf16x8 a[WFM], b[WFN];
                    rd(a, b, pa8, st, ks);
                    mm(a, b);
                }
            };
            int st = 0, t = 0;
            for (; t + NST < c_nt; ++t) {               // steady state: tiles t+1, t+2 landed or in flight
                compute_patch(t, st);
                vm_wait<NVB * (NST - 2)>();
                lds_barrier();
                issue(st);
                st = (st + 1 == NST) ? 0 : st + 1;
            }
            for (; t + 1 < c_nt; ++t) {
                compute_patch(t, st);
                const int ahead = c_nt - t - 2;
                if (NST >= 5 && ahead >= 3) vm_wait<NVB * 3>(); else if (NST >= 4 && ahead >= 2) vm_wait<NVB * 2>(); else if (ahead >= 1) vm_wait<NVB>(); else vm_wait<0>();
                lds_barrier();
                st = (st + 1 == NST) ? 0 : st + 1;
            }
            compute_patch(t, st);
        }
    } else if constexpr (GLDS) {
        // DIRECT-TO-LDS RING.  No staging registers and no ds_write: every wave issues NVA + NVB global_load_lds_dwordx4 per k-tile (1 KiB
        // each) into a ring of NST stages.  Waits are COUNTED (s_waitcnt vmcnt(N) leaves the younger tiles in flight) and the barrier is
        // the raw s_barrier: __syncthreads() would drain the queue (an LDS-DMA is a pending LDS write on the VM counter).  A tile is
        // read only after [its issuing wave's vmcnt] + [a barrier]; a stage is re-filled only after the barrier that ends its reads.
        static_assert(!RALF_GEMM_PERSISTENT, "the direct-to-LDS loop is one tile per workgroup");
        constexpr int NST = gemm_nbuf<GATHER, FM>();
        constexpr int STAGE = (BM + BN) * BK, NLD = NVA + NVB;
        auto issue = [&](int st) {
            T* sa = la0 + st * STAGE + wave * (NVA * 512);
            T* sb = la0 + st * STAGE + BM * BK + wave * (NVB * 512);
#pragma unroll
            for (int i = 0; i < NVA; ++i) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pa[i], LDS_PTR(void, sa + i * 512), 16, 0, 0);
                pa[i] += stepA;
            }
#pragma unroll
            for (int i = 0; i < NVB; ++i) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pb[i], LDS_PTR(void, sb + i * 512), 16, 0, 0);
                pb[i] += stepB;
            }
        };
        issue(0);
        if (c_nt > 1) issue(1);
        if (NST == 3 && c_nt > 2) issue(2);
        if (c_nt >= NST) vm_wait<NLD * (NST - 1)>();
        else if (NST == 3 && c_nt == 2) vm_wait<NLD>();
        else vm_wait<0>();
        lds_barrier();
        int st = 0, t = 0;
        for (; t + NST < c_nt; ++t) {   // steady state: tiles t+1 .. t+NST-1 in flight
            compute(la0 + st * STAGE, la0 + st * STAGE + BM * BK);
            vm_wait<NLD * (NST - 2)>();   // tile t+1 has landed (this wave's share; the barrier covers the others')
            lds_barrier();
            issue(st);                    // tile t+NST into the stage whose reads just ended
            st = (st + 1 == NST) ? 0 : st + 1;
        }
        for (; t + 1 < c_nt; ++t) {       // drain: nothing left to issue
            compute(la0 + st * STAGE, la0 + st * STAGE + BM * BK);
            if (NST == 3 && c_nt - t - 2 == 1) vm_wait<NLD>();
            else vm_wait<0>();
            lds_barrier();
            st = (st + 1 == NST) ? 0 : st + 1;
        }
        compute(la0 + st * STAGE, la0 + st * STAGE + BM * BK);
    } else if constexpr (PF == 1) {   // one LDS buffer (a second one measured 46.6 -> 54.3 us on the layer1 3x3 convolutions)
        stage(la0, lb0, ra0, rb0, rc0, c_kbeg, okm0);
        __syncthreads();
        for (int t = 0; t + 1 < c_nt; ++t) {
            gload(ra0, rb0, rc0, c_kbeg + (t + 1) * BK, okm0);
            compute(la0, lb0);
            __syncthreads();
            stage(la0, lb0, ra0, rb0, rc0, c_kbeg + (t + 1) * BK, okm0);
            __syncthreads();
        }
        compute(la0, lb0);
    } else if (PAR && c_nt == 0) {
        // (a parity class without taps -- the odd pixels of a 1 x 1 stride-2 data gradient: the product is zero, the epilogue still runs)
    } else {
        stage(la0, lb0, ra0, rb0, rc0, c_kbeg, okm0);
        if (c_nt > 2) gload(ra0, rb0, rc0, c_kbeg + 2 * BK, okm0);
        __syncthreads();
        RALF_PROBE(2);
        // top of an even step t: buffer 0 = tile t, set 1 = tile t+1 and set 0 = tile t+2 (both on their way)
        int t = 0;
        for (; t + 4 < c_nt; t += 2) {
            stage(la1, lb1, ra1, rb1, rc1, c_kbeg + (t + 1) * BK, okm1);
            gload(ra1, rb1, rc1, c_kbeg + (t + 3) * BK, okm1);
            compute(la0, lb0);
            __syncthreads();
            stage(la0, lb0, ra0, rb0, rc0, c_kbeg + (t + 2) * BK, okm0);
            gload(ra0, rb0, rc0, c_kbeg + (t + 4) * BK, okm0);
            compute(la1, lb1);
            __syncthreads();
        }
        const int rem = c_nt - t;             // 1..4 tiles left; tile t+3 (rem == 4) has not been requested yet
        if (rem >= 2) stage(la1, lb1, ra1, rb1, rc1, c_kbeg + (t + 1) * BK, okm1);
        if (rem == 4) gload(ra1, rb1, rc1, c_kbeg + (t + 3) * BK, okm1);
        compute(la0, lb0);
        if (rem >= 2) {
            __syncthreads();
            if (rem >= 3) stage(la0, lb0, ra0, rb0, rc0, c_kbeg + (t + 2) * BK, okm0);
            compute(la1, lb1);
        }
        if (rem >= 3) {
            __syncthreads();
            if (rem == 4) stage(la1, lb1, ra1, rb1, rc1, c_kbeg + (t + 3) * BK, okm1);
            compute(la0, lb0);
        }
        if (rem == 4) {
            __syncthreads();
            compute(la1, lb1);
        }
    }
    // the operands of this workgroup's next tile start their way while this tile's results are stored
    const int nbid = bid + grid_x;
    const bool more = RALF_GEMM_PERSISTENT && nbid < total;
    if (more) {
        setup(nbid);
        gload(ra0, rb0, rc0, kbeg, okm0);
        if (PF == 2 && nt > 1) gload(ra1, rb1, rc1, kbeg + BK, okm1);
    }

    if constexpr (CS) {
        if (cs_on) {   // lane (l31, lh) holds the sum over its k-slices of fragment row l31: the two k-halves meet, the lower half stores
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                float a = csacc[i], b = csacc[i];
                wave::swap32(a, b);
                const float v = a + b;
                const int m = c_m0 + wm * 32 * WFM + i * 32 + (lane & 31);
                if (lane < 32 && m < d.M) {
                    if (d.splitk > 1) P.colsum_partial[(int64_t)c_split * d.M + m] = v;
                    else P.colsum[m] += v;
                }
            }
        }
    }
    RALF_PROBE(3);
    // ---- epilogue: lane holds 4 consecutive columns n per register group g = r>>2 ----
    // (no `continue`/`break` in these loops: they must unroll completely or the accumulators spill to scratch)
    const bool slab = d.splitk > 1 && !d.atomic_out;
    if (EPI == 4 || (P.vec_epi >= 2 && c_n0 + BN <= d.N)) {   // (the filter checks the column range itself: partial tiles take the staged form too)
        // tile interior in n: the accumulators go through LDS so every lane stores 8 consecutive columns of one row
        // (8 lanes = one 128-byte line of bf16) instead of 32 rows x 8 bytes per store instruction; residual / mask /
        // accumulate reads get the same shape.  64 tile rows per round.
        float* cs = reinterpret_cast<float*>(lds_raw);
        float* pbase = slab ? P.partial + ((int64_t)c_split * nbatch + z) * d.M * d.N : nullptr;
        // (BatchNorm-backward sums: at most 32 thread rows -- 64-column tiles on 8 waves leave half their threads out of this epilogue -- so that every
        //  thread sums rows lr, lr + 32 and the block sum runs over 32 partials in one order whatever the tile shape)
        constexpr int CG = BN / 8, RPP = (EPI == 3 && NT / CG > 32) ? 32 : NT / CG;
        const bool epi_on = tid < RPP * CG;
#pragma clang loop unroll(full)
        for (int h = 0; h < FM; ++h) {
            __syncthreads();
            // round h = tile rows 64 h .. 64 h + 63 = two 32-row fragments: of both wave rows (FM = 1), or fragments (2h) % WFM, + 1 of
            // wave row (2h) / WFM (FM = 2: the wave row's two fragments; FM = 4: half of a wave row's four)
            if (FM == 1 || wm == (2 * h) / WFM) {
#pragma clang loop unroll(full)
                for (int ii = 0; ii < (FM == 1 ? 1 : 2); ++ii) {
                    const int i = FM == 1 ? 0 : (2 * h) % WFM + ii;
                    const int lr = (FM == 1 ? wm * 32 : ii * 32) + l31;
#pragma clang loop unroll(full)
                    for (int j = 0; j < WFN; ++j) {
#pragma clang loop unroll(full)
                        for (int g = 0; g < 4; ++g)
                            *reinterpret_cast<float4*>(cs + lr * CP + wn * 32 * WFN + j * 32 + 8 * g + 4 * lh) =
                                make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
                    }
                }
            }
            __syncthreads();
            if (d.colstats) {   // BatchNorm batch statistics of this 64-row block, on the values as stored (rounded to T)
                // (16 columns per wave at least -- 64-column tiles on 8 waves leave four of them idle here -- so that a column's rows are summed in the
                //  order of the 64 x 64 / 128 x 128 tiles: the statistics do not depend on the tile shape)
                constexpr int CPW = BN / NW < 16 ? 16 : BN / NW, RG = 64 / CPW;   // columns per wave, row groups per column
                const int col = wave * CPW + (lane % CPW), rg = lane / CPW;
                const int nvalid = wave * CPW < BN ? min(64, d.M - (c_m0 + h * 64)) : 0;   // <= 0: this 64-row block lies beyond M (no partial row exists)
                float s1 = 0.f, s2 = 0.f;
#pragma unroll 8
                for (int r = rg; r < nvalid; r += RG) {
                    const float v = (float)(T)cs[r * CP + col];
                    s1 += v; s2 += v * v;
                }
#pragma unroll
                for (int o = CPW; o < 64; o <<= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
                if (rg == 0 && nvalid > 0) {
                    float* pr = d.colstats + ((int64_t)(c_m0 / 64 + h) * 2) * d.N + c_n0 + col;
                    pr[0] = s1; pr[d.N] = s2;
                }
            }
            unsigned* fcnt = reinterpret_cast<unsigned*>(cs + 64 * CP);   // EPI 4: hit counters of the round's 64 rows (behind the C staging tile)
            if constexpr (EPI == 4) {
                static_assert(64 * CP * 4 + 64 * 4 <= gemm_lds_bytes<T, AK, BKC, FM, FN, gemm_nbuf<GATHER, FM>(), gemm_is_glds(GATHER)>(), "filter counters behind the staging tile");
                if (tid < 64) fcnt[tid] = 0u;
                __syncthreads();
            }
            float bs1[8], bs2[8], bmu[8];   // EPI 3: this thread's column sums over its rows of the block, the columns' BatchNorm means
            if constexpr (EPI == 3) {
                VIO<float, 8>::ld(d.bnb_mean, c_n0 + (tid % CG) * 8, bmu);
#pragma unroll
                for (int q = 0; q < 8; ++q) bs1[q] = bs2[q] = 0.f;
            }
#pragma unroll
            for (int p = 0; p < 64 / RPP; ++p) {
                const int lr = p * RPP + tid / CG, c = (tid % CG) * 8;
                const int mv = c_m0 + h * 64 + lr;                         // (PAR: the virtual row; memory is addressed by the real one)
                if (mv < d.M && epi_on) {
                    const int m = PAR ? par_real_row(P, mv) : mv;
                    const float4 lo = *reinterpret_cast<const float4*>(cs + lr * CP + c), hi = *reinterpret_cast<const float4*>(cs + lr * CP + c + 4);
                    float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                    if constexpr (EPI == 3) epilogue_storev_bnb<T>(d, m, c_n0 + c, v, bmu, bs1, bs2);
                    else if constexpr (EPI == 4)
                        epilogue_filter<8>(d, m, c_n0 + c, v, d.N, d.flt_thresh[(int64_t)m * (d.flt_thresh_ld > 0 ? d.flt_thresh_ld : 1)], fcnt + lr,
                                           reinterpret_cast<int2*>(d.flt_list) + ((int64_t)m * P.tiles_n + c_n0 / BN) * d.flt_cap);
                    else if (slab) VIO<float, 8>::st(pbase, (int64_t)m * d.N + c_n0 + c, v);
                    else epilogue_storev<T, (EPI >= 3 ? 0 : EPI), 8>(d, z0, z1, m, c_n0 + c, v);
                }
            }
            if constexpr (EPI == 4) {
                __syncthreads();
                const int m = c_m0 + h * 64 + tid;
                if (tid < 64 && m < d.M) d.flt_count[(int64_t)m * P.tiles_n + c_n0 / BN] = (int)fcnt[tid];
            }
            if constexpr (EPI == 3) {
                // the RPP threads that share a column group hand their sums over through the (now idle) staging tile, fixed order
                static_assert(RPP * 2 * BN <= 64 * CP, "the partial sums reuse the C staging tile");
                __syncthreads();
                float* ps = cs + (tid / CG) * 2 * BN + (tid % CG) * 8;
                if (epi_on) {
                    *reinterpret_cast<float4*>(ps) = make_float4(bs1[0], bs1[1], bs1[2], bs1[3]);
                    *reinterpret_cast<float4*>(ps + 4) = make_float4(bs1[4], bs1[5], bs1[6], bs1[7]);
                    *reinterpret_cast<float4*>(ps + BN) = make_float4(bs2[0], bs2[1], bs2[2], bs2[3]);
                    *reinterpret_cast<float4*>(ps + BN + 4) = make_float4(bs2[4], bs2[5], bs2[6], bs2[7]);
                }
                __syncthreads();
                if (tid < 2 * BN && c_m0 + h * 64 < d.M) {
                    float t = 0.f;
#pragma unroll 8
                    for (int r = 0; r < RPP; ++r) t += cs[r * 2 * BN + tid];
                    d.bnb_part[((int64_t)(c_m0 / 64 + h) * 2 + tid / BN) * d.N + c_n0 + tid % BN] = t;
                }
            }
        }
    } else {
#pragma clang loop unroll(full)
    for (int i = 0; i < FM; ++i) {
        const int mv = c_m0 + wm * 32 * WFM + i * 32 + l31;
        const int m = (PAR && mv < d.M) ? par_real_row(P, mv) : mv;
#pragma clang loop unroll(full)
        for (int j = 0; j < WFN; ++j) {
#pragma clang loop unroll(full)
            for (int g = 0; g < 4; ++g) {
                const int n = c_n0 + wn * 32 * WFN + j * 32 + 8 * g + 4 * lh;
                float v[4] = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                if (mv < d.M && n < d.N) {
                    if (slab) {
                        float* pp = P.partial + (((int64_t)c_split * nbatch + z) * d.M + m) * d.N + n;
                        if (P.vec_epi && n + 3 < d.N) *reinterpret_cast<float4*>(pp) = make_float4(v[0], v[1], v[2], v[3]);
                        else {
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                if (n + q < d.N) pp[q] = v[q];
                        }
                    } else if (P.vec_epi && n + 3 < d.N) {
                        epilogue_store4<T, (EPI >= 3 ? 0 : EPI)>(d, z0, z1, m, n, v);
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (n + q < d.N) epilogue_store<T, (EPI >= 3 ? 0 : EPI)>(d, z0, z1, m, n + q, v[q]);
                    }
                }
            }
        }
    }
    }
    RALF_PROBE(4);
    if (!more) break;
    bid = nbid;
    __syncthreads();   // the C staging tile shares the LDS with the operand tiles
    }
}

template <typename T, bool AK, bool BKC, int GATHER, int FM, int FN, int EPI, int NW = 4, int WGM = 2>
__global__ __launch_bounds__(64 * NW, (GATHER == 15 && FN == 1) ? 2 : NW == 8 ? (GATHER == 15 ? 1 : (GATHER == 6 || FM == 4 || gemm_at_mode(GATHER) || gemm_is_mt(GATHER)) ? 2 : 4) : 1) void gemm_kernel(const KParams P) {
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[GATHER == 15 ? gemm_patch_ring(FN) + gemm_patch_bytes(FN) :
                                                                  gemm_lds_bytes<T, AK, BKC, FM, FN, gemm_nbuf<GATHER, FM>(), gemm_is_glds(GATHER)>() +
                                                                  (gemm_at_mode(GATHER) ? AT_LDS_BYTES : 0)];   // ONE LDS object (+ the operand transform's coefficient table)
    gemm_body<T, AK, BKC, GATHER, FM, FN, EPI, NW, false, WGM>(P, (int)blockIdx.x, (int)gridDim.x, (int)blockIdx.z, (int)gridDim.z, lds_raw);
}

// ---- grouped weight gradients ---------------------------------------------------------------------------------------
// Many C_j[M_j, N_j] (fp32) += A_j^T B_j products in ONE launch: the weight gradients dW = dy^T x of several linear layers
// (A_j = dy_j [K_j rows][M_j], B_j = x_j [K_j rows][N_j], both row-major as the activations are stored).  Every output tile is
// owned by one workgroup that walks the whole reduction (or one k-split of it, into a slab), so the result is deterministic
// and needs no atomics; with a few dozen layers per launch there are enough tiles without splitting the reduction of the
// transformer-sized problems.  Job records travel by value in the kernel arguments (graph-capture friendly).
struct GJob {
    const void* A; const void* B; float* C; float* partial;
    float* db; float* db_partial;   // bias gradient [M] += column sums of A (NULL: none), its split-K slabs [splitk][M]
    int M, N, K, lda, ldb, ldc, splitk, kchunk, tiles_n, nwg, first;   // first = first workgroup of the job
    int pad;
};
constexpr int GROUP_MAX = 40;   // (the records travel in the kernel arguments: 4 KiB)
struct GParams { int njobs; int pad[3]; GJob j[GROUP_MAX]; };

template <typename T, int FM, int FN, int NW, int GATHER = 3>   // GATHER 3: register-staged, 5: direct-to-LDS ring
__global__ __launch_bounds__(64 * NW, NW == 8 ? 4 : 1) void gemm_grouped_kernel(const GParams G) {
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[gemm_lds_bytes<T, false, false, FM, FN, 2, GATHER == 5>()];
    const int b = (int)blockIdx.x;
    int ji = 0;
    while (ji + 1 < G.njobs && b >= G.j[ji + 1].first) ++ji;   // (uniform: scalar loads from the kernel arguments)
    const GJob& J = G.j[ji];
    KParams P;
    RalfGemmDesc& d = P.d;
    d.A = J.A; d.B = J.B; d.C = J.C; d.C2 = nullptr; d.bias = nullptr; d.res = nullptr; d.aux = nullptr;
    d.lda = J.lda; d.ldb = J.ldb; d.ldc = J.ldc; d.ldr = 0;
    d.sA0 = d.sA1 = d.sB0 = d.sB1 = d.sC0 = d.sC1 = d.sR0 = d.sR1 = 0;
    d.M = J.M; d.N = J.N; d.K = J.K; d.nb0 = d.nb1 = 1;
    d.dtype = RALF_BF16; d.a_kcontig = 0; d.b_kcontig = 0; d.gather = 0;
    d.act = 0; d.aux_mode = 0; d.out_f32 = 1; d.accumulate = 1; d.splitk = J.splitk;
    d.alpha = 1.f; d.aux_scale = 1.f; d.seed = nullptr; d.call_id = 0; d.drop_p = 0.f; d.atomic_out = 0; d.colstats = nullptr;
    d.sBias0 = 0; d.sBk = 0; d.kseg = 0; d.colscale = nullptr;
    d.bnb_x = nullptr; d.bnb_mask = nullptr; d.bnb_mean = nullptr; d.bnb_part = nullptr;
    P.tiles_n = J.tiles_n; P.tiles_m = J.nwg / J.tiles_n; P.nwg = J.nwg; P.kchunk = J.kchunk; P.partial = J.partial; P.mfast = 0;
    P.vec_epi = 2; P.fast = 1; P.tapuni = 0;
    P.colsum = J.db; P.colsum_partial = J.db_partial;
    gemm_body<T, false, false, GATHER, FM, FN, 0, NW, GATHER == 5>(P, b - J.first, 1, 0, 1, lds_raw);
}

// C_j += sum over the k-splits of job j's slabs (jobs with splitk > 1 only); first = first workgroup, 2048 outputs per workgroup
struct GRed { const float* partial; float* C; int64_t per; int ldc, N, splitk, first; };
struct GRedParams { int njobs; int pad[3]; GRed j[2 * GROUP_MAX]; };   // (a job's weight slabs and its bias slabs)
__global__ __launch_bounds__(256) void gemm_grouped_reduce_kernel(const GRedParams G) {
    const int b = (int)blockIdx.x;
    int ji = 0;
    while (ji + 1 < G.njobs && b >= G.j[ji + 1].first) ++ji;
    const GRed& J = G.j[ji];
    const int64_t e0 = ((int64_t)(b - J.first) * 256 + threadIdx.x) * 8;
    if (e0 >= J.per) return;
    float a[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) a[q] = 0.f;
    for (int sp = 0; sp < J.splitk; ++sp) {   // slabs in split order: deterministic
        const float4 lo = *reinterpret_cast<const float4*>(J.partial + (int64_t)sp * J.per + e0), hi = *reinterpret_cast<const float4*>(J.partial + (int64_t)sp * J.per + e0 + 4);
        a[0] += lo.x; a[1] += lo.y; a[2] += lo.z; a[3] += lo.w; a[4] += hi.x; a[5] += hi.y; a[6] += hi.z; a[7] += hi.w;
    }
    float* c = J.C + (e0 / J.N) * J.ldc + (e0 % J.N);   // N % 8 == 0: the 8 outputs share a row
    float4 lo = *reinterpret_cast<float4*>(c), hi = *reinterpret_cast<float4*>(c + 4);
    lo.x += a[0]; lo.y += a[1]; lo.z += a[2]; lo.w += a[3]; hi.x += a[4]; hi.y += a[5]; hi.z += a[6]; hi.w += a[7];
    *reinterpret_cast<float4*>(c) = lo; *reinterpret_cast<float4*>(c + 4) = hi;
}

// ---- few-row products (KV-cached decode: M = batch rows) ---------------------------------------------------------------------
// One WAVE per 32x32 output tile, no LDS, no barrier: lane (r = lane & 31, h = lane >> 5) loads the 16-byte k-slices of ITS fragment
// row of A and of B straight from global memory (both k-contiguous) and the wave runs the K/16 MFMAs of its tile -- the same MFMA
// chain in the same order as the tiled kernel (bit-identical results).  A decode-step linear layer (256 x 256..1024 x 256..1024) is
// pure latency: the tiled kernel's index set-up, two staging hops and barriers cost 7.2 us per launch (1 266 launches per batch);
// here a tile is one load round trip + K/16 MFMAs.
// KW = 4 (long reductions, K % 512 == 0: the feed-forward block's second product, 256 x 256 x 1024): four waves per tile, each runs a
// quarter of the k range, the partial tiles are summed through LDS in wave order (deterministic; NOT the tiled kernel's single chain any
// more -- bf16 throughput mode only) -- a 64-MFMA chain behind 128 KB of loads per wave took 16 us, half of all few-row time in a decode step.
// LN (K == 256, CH = 8, KW = 1: both register sets together hold the lane's half of its row): LayerNorm of the 32 A rows in registers before
// the MFMAs -- the two lanes of a row (l31, lh = 0 / 1) own the k-octets of their parity, one shuffle joins their sums; formulas of ln_fwd_kernel.
template <typename T, int EPI, int CH, int KW, bool LN = false>
__global__ __launch_bounds__(64 * KW) void gemm_skinny_kernel(const KParams P) {
    static_assert(sizeof(T) == 2, "bf16 only");
    const RalfGemmDesc& d = P.d;
    const int lane = threadIdx.x & 63, l31 = lane & 31, lh = lane >> 5, kw = KW > 1 ? (int)(threadIdx.x >> 6) : 0;
    const int tm = (int)blockIdx.x / P.tiles_n, tn = (int)blockIdx.x - tm * P.tiles_n;
    const int m0 = tm * 32, n0 = tn * 32;
    const int kspan = d.K / KW;                                                              // this wave's k range: [kw * kspan, (kw + 1) * kspan)
    const bf16* a = (const bf16*)d.A + (int64_t)min(m0 + l31, d.M - 1) * d.lda + lh * 8 + kw * kspan;   // (rows beyond M / N are clamped: never stored)
    const bf16* b = (const bf16*)d.B + (int64_t)min(n0 + l31, d.N - 1) * d.ldb + lh * 8 + kw * kspan;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // chunks of CH 16-wide k-slices, two register sets: 2 x CH x 2 loads in flight (K = 256 with CH = 8: everything at once)
    bf16x8 a0[CH], b0[CH], a1[CH], b1[CH];
    auto load = [&](bf16x8 (&ar)[CH], bf16x8 (&br)[CH], int c) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            ar[i] = *reinterpret_cast<const bf16x8*>(a + (c * CH + i) * 16);
            br[i] = *reinterpret_cast<const bf16x8*>(b + (c * CH + i) * 16);
        }
    };
    auto comp = [&](const bf16x8 (&ar)[CH], const bf16x8 (&br)[CH]) {
#pragma unroll
        for (int i = 0; i < CH; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(br[i], ar[i], acc, 0, 0, 0);
    };
    const int nch = kspan / (16 * CH);
    load(a0, b0, 0);
    if constexpr (LN) {
        static_assert(CH == 8 && KW == 1, "the LayerNorm prologue keeps a 256-wide row in the two register sets");
        load(a1, b1, 1);
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < CH; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) sum += (float)a0[i][e] + (float)a1[i][e];
        sum += __shfl_xor(sum, 32);
        const float mu = sum * (1.f / 256.f);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < CH; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d0 = (float)a0[i][e] - mu, d1 = (float)a1[i][e] - mu;
                q = __fmaf_rn(d0, d0, q);
                q = __fmaf_rn(d1, d1, q);
            }
        q += __shfl_xor(q, 32);
        const float rs = rsqrtf(__fmaf_rn(q, 1.f / 256.f, d.ln_eps));
#pragma unroll
        for (int i = 0; i < CH; ++i) {   // element e of chunk i sits at column 16 i + 8 lh + e (set 0) / 128 + that (set 1)
            const int k0 = 16 * i + 8 * lh;
            const float4 g0 = *reinterpret_cast<const float4*>(d.ln_g + k0), g1 = *reinterpret_cast<const float4*>(d.ln_g + k0 + 4);
            const float4 h0 = *reinterpret_cast<const float4*>(d.ln_b + k0), h1 = *reinterpret_cast<const float4*>(d.ln_b + k0 + 4);
            const float4 g2 = *reinterpret_cast<const float4*>(d.ln_g + 128 + k0), g3 = *reinterpret_cast<const float4*>(d.ln_g + 128 + k0 + 4);
            const float4 h2 = *reinterpret_cast<const float4*>(d.ln_b + 128 + k0), h3 = *reinterpret_cast<const float4*>(d.ln_b + 128 + k0 + 4);
            const float ga[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, ba[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
            const float gb[8] = {g2.x, g2.y, g2.z, g2.w, g3.x, g3.y, g3.z, g3.w}, bb[8] = {h2.x, h2.y, h2.z, h2.w, h3.x, h3.y, h3.z, h3.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                a0[i][e] = (bf16)__fmaf_rn(((float)a0[i][e] - mu) * rs, ga[e], ba[e]);
                a1[i][e] = (bf16)__fmaf_rn(((float)a1[i][e] - mu) * rs, gb[e], bb[e]);
            }
        }
        comp(a0, b0);
        comp(a1, b1);
    } else {
    int c = 0;
    for (; c + 2 < nch; c += 2) {
        load(a1, b1, c + 1);
        comp(a0, b0);
        load(a0, b0, c + 2);
        comp(a1, b1);
    }
    if (c + 1 < nch) {
        load(a1, b1, c + 1);
        comp(a0, b0);
        comp(a1, b1);
    } else {
        comp(a0, b0);
    }
    }
    if constexpr (KW > 1) {
        __shared__ float red[KW - 1][16][64];
        if (kw > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[kw - 1][r][lane] = acc[r];
        }
        __syncthreads();
        if (kw > 0) return;
#pragma unroll
        for (int w = 0; w < KW - 1; ++w)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += red[w][r][lane];
    }
    const int m = m0 + l31;
    // (written out per register group: a loop the compiler leaves rolled indexes the accumulator dynamically = scratch memory)
#define RALF_SKINNY_STORE(g)                                                                              \
    {                                                                                                     \
        const int n = n0 + 8 * (g) + 4 * lh;                                                              \
        float v[4] = {acc[4 * (g)], acc[4 * (g) + 1], acc[4 * (g) + 2], acc[4 * (g) + 3]};                \
        if (m < d.M && n < d.N) {                                                                         \
            if (P.vec_epi && n + 3 < d.N) {                                                               \
                epilogue_store4<T, EPI>(d, 0, 0, m, n, v);                                                \
            } else {                                                                                      \
                if (n + 0 < d.N) epilogue_store<T, EPI>(d, 0, 0, m, n + 0, v[0]);                         \
                if (n + 1 < d.N) epilogue_store<T, EPI>(d, 0, 0, m, n + 1, v[1]);                         \
                if (n + 2 < d.N) epilogue_store<T, EPI>(d, 0, 0, m, n + 2, v[2]);                         \
                if (n + 3 < d.N) epilogue_store<T, EPI>(d, 0, 0, m, n + 3, v[3]);                         \
            }                                                                                             \
        }                                                                                                 \
    }
    RALF_SKINNY_STORE(0) RALF_SKINNY_STORE(1) RALF_SKINNY_STORE(2) RALF_SKINNY_STORE(3)
#undef RALF_SKINNY_STORE
}

// fp32 (the parity mode) few-row products: one wave per 16x16 output tile on v_mfma_f32_16x16x4_f32 -- a chain of K / 4 matrix instructions on
// one accumulator is the ascending-k fma chain of the tiled kernel's v_mfma_f32_32x32x2_f32 (same bits: tests/test_gemm_gpu.py), but 256 waves
// instead of 16 workgroups share a 256 x 256 product and a wave's chain is 64 x 32 cycles instead of 128 x 64.  Operands travel global ->
// registers (a whole 256-wide k chunk of both operands requested at once, two chunks in flight) -> LDS (rotated rows as in knn.hip:
// conflict-free 4-byte fragment reads) one 32-wide k-tile at a time; a single wave needs no barrier beyond the compiler's ordering.
// The tiled kernel took 15.3 us per launch on the decode step's 256 x 256..1024 x 256 products (16-64 workgroups on 256 CUs).
template <int EPI>
__global__ __launch_bounds__(64) void gemm_skinny_f32_kernel(const KParams P) {
    const RalfGemmDesc& d = P.d;
    __shared__ float la[16 * 32], lb[16 * 32];
    const int lane = threadIdx.x, lr = lane & 15, lk = lane >> 4;
    const int tm = (int)blockIdx.x / P.tiles_n, tn = (int)blockIdx.x - tm * P.tiles_n;
    const int m0 = tm * 16, n0 = tn * 16;
    // loader: float4 #i of a 16 x 32 k-tile = row (lane + 64 i) >> 3, k-quad (lane + 64 i) & 7
    const int r0 = lane >> 3, kq = lane & 7;
    const float* a0p = (const float*)d.A + (int64_t)min(m0 + r0, d.M - 1) * d.lda + kq * 4;
    const float* a1p = (const float*)d.A + (int64_t)min(m0 + r0 + 8, d.M - 1) * d.lda + kq * 4;
    const float* b0p = (const float*)d.B + (int64_t)min(n0 + r0, d.N - 1) * d.ldb + kq * 4;
    const float* b1p = (const float*)d.B + (int64_t)min(n0 + r0 + 8, d.N - 1) * d.ldb + kq * 4;
    constexpr int CT = 8;                                   // k-tiles per chunk (256 k)
    struct Chunk { float4 a[CT][2], b[CT][2]; };
    Chunk c0, c1;
    auto load = [&](Chunk& c, int chunk) {
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            const int k = (chunk * CT + t) * 32;
            c.a[t][0] = *reinterpret_cast<const float4*>(a0p + k); c.a[t][1] = *reinterpret_cast<const float4*>(a1p + k);
            c.b[t][0] = *reinterpret_cast<const float4*>(b0p + k); c.b[t][1] = *reinterpret_cast<const float4*>(b1p + k);
        }
    };
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // element (row, k) of a tile lives at row * 32 + ((k + 2 row) & 31)
    const int w0 = r0 * 32, w1 = (r0 + 8) * 32, rot0 = kq * 4 + 2 * r0, rot1 = kq * 4 + 2 * (r0 + 8);
    const int foff = lr * 32, frot = 2 * lr + lk;
    auto comp = [&](const Chunk& c, int ntile) {
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            if (t < ntile) {
                la[w0 + ((rot0 + 0) & 31)] = c.a[t][0].x; la[w0 + ((rot0 + 1) & 31)] = c.a[t][0].y; la[w0 + ((rot0 + 2) & 31)] = c.a[t][0].z; la[w0 + ((rot0 + 3) & 31)] = c.a[t][0].w;
                la[w1 + ((rot1 + 0) & 31)] = c.a[t][1].x; la[w1 + ((rot1 + 1) & 31)] = c.a[t][1].y; la[w1 + ((rot1 + 2) & 31)] = c.a[t][1].z; la[w1 + ((rot1 + 3) & 31)] = c.a[t][1].w;
                lb[w0 + ((rot0 + 0) & 31)] = c.b[t][0].x; lb[w0 + ((rot0 + 1) & 31)] = c.b[t][0].y; lb[w0 + ((rot0 + 2) & 31)] = c.b[t][0].z; lb[w0 + ((rot0 + 3) & 31)] = c.b[t][0].w;
                lb[w1 + ((rot1 + 0) & 31)] = c.b[t][1].x; lb[w1 + ((rot1 + 1) & 31)] = c.b[t][1].y; lb[w1 + ((rot1 + 2) & 31)] = c.b[t][1].z; lb[w1 + ((rot1 + 3) & 31)] = c.b[t][1].w;
                __syncthreads();
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const float av = la[foff + ((frot + 4 * s) & 31)], bv = lb[foff + ((frot + 4 * s) & 31)];
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(bv, av, acc, 0, 0, 0);   // rows = n (4 consecutive per lane), columns = m
                }
                __syncthreads();
            }
        }
    };
    const int ntiles = d.K / 32, nch = (ntiles + CT - 1) / CT;
    load(c0, 0);                                            // (a chunk may reach past K: K % 256 == 0 is the dispatch rule)
    int c = 0;
    for (; c + 2 < nch; c += 2) {
        load(c1, c + 1);
        comp(c0, CT);
        load(c0, c + 2);
        comp(c1, CT);
    }
    if (c + 1 < nch) {
        load(c1, c + 1);
        comp(c0, CT);
        comp(c1, CT);
    } else {
        comp(c0, CT);
    }
    const int m = m0 + lr, n = n0 + 4 * lk;
    if (m < d.M && n < d.N) {
        float v[4] = {acc[0], acc[1], acc[2], acc[3]};
        if (P.vec_epi && n + 3 < d.N) {
            epilogue_store4<float, EPI>(d, 0, 0, m, n, v);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (n + q < d.N) epilogue_store<float, EPI>(d, 0, 0, m, n + q, v[q]);
        }
    }
}

template <typename T, int EPI>
int launch_skinny(KParams& P, hipStream_t st) {
    P.tiles_m = ceil_div(P.d.M, 32);
    P.tiles_n = ceil_div(P.d.N, 32);
    if constexpr (sizeof(T) == 4) {
        P.tiles_m = ceil_div(P.d.M, 16);
        P.tiles_n = ceil_div(P.d.N, 16);
        hipLaunchKernelGGL((gemm_skinny_f32_kernel<EPI>), dim3(P.tiles_m * P.tiles_n), dim3(64), 0, st, P);
        return ralf::check_launch("gemm (few rows, fp32)");
    } else {
    static const int split = [] { const char* e = getenv("RALF_GEMM_SKINNY_SPLIT"); return e ? atoi(e) : 1; }();   // 0 = off (A/B runs)
    if (P.d.ln_g) hipLaunchKernelGGL((gemm_skinny_kernel<T, EPI, 8, 1, true>), dim3(P.tiles_m * P.tiles_n), dim3(64), 0, st, P);   // (K == 256: checked at entry)
    else if (split && P.d.few_row_split && P.d.K % 512 == 0) hipLaunchKernelGGL((gemm_skinny_kernel<T, EPI, 8, 4>), dim3(P.tiles_m * P.tiles_n), dim3(256), 0, st, P);
    else if (P.d.K % 128 == 0) hipLaunchKernelGGL((gemm_skinny_kernel<T, EPI, 8, 1>), dim3(P.tiles_m * P.tiles_n), dim3(64), 0, st, P);
    else hipLaunchKernelGGL((gemm_skinny_kernel<T, EPI, 4, 1>), dim3(P.tiles_m * P.tiles_n), dim3(64), 0, st, P);
    return ralf::check_launch("gemm (few rows)");
    }
}

template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const KParams P, int nbatch) {
    const RalfGemmDesc& d = P.d;
    const int64_t per = (int64_t)d.M * d.N, total = per * nbatch;
    if (P.vec_epi && d.N % 4 == 0) {
        for (int64_t e4 = (int64_t)blockIdx.x * 256 + threadIdx.x; e4 < total / 4; e4 += (int64_t)gridDim.x * 256) {
            const int64_t e = e4 * 4;
            // slabs summed in split order (deterministic); 8 loads in flight: a tiny output with 64 splits was one dependent
            // load after the other (33-63 us for the layer1 weight gradients)
            float4 a = *reinterpret_cast<const float4*>(P.partial + e);
            int s = 1;
            for (; s + 8 <= d.splitk; s += 8) {
                float4 t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = *reinterpret_cast<const float4*>(P.partial + (int64_t)(s + u) * total + e);
#pragma unroll
                for (int u = 0; u < 8; ++u) { a.x += t[u].x; a.y += t[u].y; a.z += t[u].z; a.w += t[u].w; }
            }
            for (; s < d.splitk; ++s) {
                const float4 t = *reinterpret_cast<const float4*>(P.partial + (int64_t)s * total + e);
                a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
            }
            const int z = (int)(e / per);
            const int64_t r = e - (int64_t)z * per;
            float v[4] = {a.x, a.y, a.z, a.w};
            epilogue_store4<T, 2>(d, z % d.nb0, z / d.nb0, (int)(r / d.N), (int)(r % d.N), v);
        }
        return;
    }
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        float v = 0.f;
        for (int s = 0; s < d.splitk; ++s) v += P.partial[(int64_t)s * total + e];
        const int z = (int)(e / per);
        const int64_t r = e - (int64_t)z * per;
        epilogue_store<T, 2>(d, z % d.nb0, z / d.nb0, (int)(r / d.N), (int)(r % d.N), v);
    }
}

// resident workgroups the chip holds for one kernel instantiation (queried once): the persistent grid
template <typename K>
int resident_workgroups(K kernel, int threads) {
    int dev = 0, cus = 256, per_cu = 1;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    (void)hipGetLastError();
    return (cus / 8 * 8) * per_cu;
}

template <typename T, bool AK, bool BKC, int GATHER, int FM, int FN, int EPI, int NW = 4, int WGM = 2>
int launch(KParams& P, int nbatch, hipStream_t st) {
    P.tiles_m = ceil_div(P.d.M, 64 * FM);
    // which operand an XCD keeps in its L2 while the other streams past: workgroups get consecutive tile ids per XCD (xcd_remap), so the
    // tile index that runs fastest is the one whose operand is re-read.  Activations x weights: the weights (B) are the small operand ->
    // column tiles fastest.  A few hundred queries x a 220 MB index (the two-stage k-NN's coarse pass): with column tiles fastest every XCD
    // owned one row tile and streamed the WHOLE index (8 x 220 MB per call: 390 us); row tiles fastest keeps the 3.7 MB of queries resident
    // and streams every index row once.
    P.mfast = (AK && BKC && GATHER != 1 && GATHER != 4 && GATHER != 14 && GATHER != 15 && nbatch == 1 && P.d.splitk == 1 && !P.d.kseg && (int64_t)P.d.M * 4 <= (int64_t)P.d.N) ? 1 : 0;
    P.tiles_n = ceil_div(P.d.N, 64 * FN);
    P.nwg = P.tiles_m * P.tiles_n;
    constexpr bool persist = RALF_GEMM_PERSISTENT != 0;   // off: measured -4...+6 % (the prefetch across the epilogue costs a wave of occupancy)
    static const int cap = persist ? resident_workgroups(gemm_kernel<T, AK, BKC, GATHER, FM, FN, EPI, NW, WGM>, 64 * NW) : 0;
    const int total = P.nwg * P.d.splitk;
    int grid = total;
    const int room = std::max(8, cap / nbatch / 8 * 8);
    if (persist && total > room) {   // equal shares: every workgroup gets `rounds` (or rounds - 1) tiles
        const int rounds = ceil_div(total, room);
        grid = std::min(room, ceil_div(ceil_div(total, rounds), 8) * 8);
    }
    hipLaunchKernelGGL((gemm_kernel<T, AK, BKC, GATHER, FM, FN, EPI, NW, WGM>), dim3(grid, 1, nbatch), dim3(64 * NW), 0, st, P);
    return ralf::check_launch("gemm");
}

template <typename T, bool AK, bool BKC, int GATHER, int FM, int FN, int NW>
int launch_epi(KParams& P, int nbatch, hipStream_t st) {
    const RalfGemmDesc& d = P.d;
    if (d.bnb_part) {   // BatchNorm-backward statistics: data-gradient products only (A k-contiguous: 1x1 on the interior path, k x k through the tap gather)
        if constexpr (AK && (GATHER == 0 || GATHER == 1 || GATHER == 3 || GATHER == 5 || GATHER == 6 || GATHER == 8)) return launch<T, AK, BKC, GATHER, FM, FN, 3, NW>(P, nbatch, st);
        else { ralf::set_error("gemm: bnb_* needs a k-contiguous A (no general per-vector gather, no weight-gradient layout)"); return RALF_ERR_INVALID; }
    }
    if (d.flt_list) {   // threshold filter: the coarse pass of the two-stage top-k search (bf16 NT products on the aligned path only)
        if constexpr (AK && BKC && sizeof(T) == 2 && (GATHER == 3 || GATHER == 5 || GATHER == 6)) return launch<T, AK, BKC, GATHER, FM, FN, 4, NW>(P, nbatch, st);
        else { ralf::set_error("gemm: flt_* needs bf16 operands, both k-contiguous, on the aligned interior path"); return RALF_ERR_INVALID; }
    }
    const bool lvl2 = d.C2 || d.act == RALF_ACT_GELU || d.aux_mode == RALF_AUX_GELU_GRAD || d.atomic_out;
    const bool lvl1 = d.drop_p > 0.f || d.aux;
    if (lvl2) return launch<T, AK, BKC, GATHER, FM, FN, 2, NW>(P, nbatch, st);
    if (lvl1) return launch<T, AK, BKC, GATHER, FM, FN, 1, NW>(P, nbatch, st);
    return launch<T, AK, BKC, GATHER, FM, FN, 0, NW>(P, nbatch, st);
}

// Tile choice, measured on MI355X (tools/gemm_probe.hip, tools/gemm_bench.py).  Two configurations:
//   64x64, 4 waves (4-5 workgroups per CU): s_memtime stamps + LDS cycle counts show it LDS-bound on the model's mid-size
//     products (per tile and k-step: 16 KB of ds_write_b128 + 32 ds_read_b128 for 4 MFMAs per wave);
//   128x128, 8 waves (2 x 4, 64x32 per wave, 2 workgroups = 16 waves per CU): a third fewer LDS cycles per flop and as
//     many waves in flight -> 10-15 % faster wherever >= ~200 such tiles exist.  (128x128 with 4 waves of 64x64 has the
//     fewest LDS cycles but only 8 waves per CU: latency-bound, no faster than 64x64 below K ~ 4096; 128x64 / 64x128: +-5 %.)
// 128 x 128 tiles (8 waves) or 64 x 64 (4 waves) for this product -- ONE rule for launch_cfg and for callers that must know the tile width
// (ralf_gemm_filter_tile: the slot layout of the threshold filter)
inline bool gemm_use128(const RalfGemmDesc& d, int nbatch) {
    static const int forced = [] { const char* e = getenv("RALF_GEMM_TILE"); return e ? atoi(e) : 0; }();  // tuning / test aid: 22 / 11
    const bool ok22 = (!d.colstats && !d.bnb_part) || d.N % 128 == 0;   // column statistics come from the staged epilogue: every tile interior in n
    if (forced == 22 && ok22) return true;
    if (forced == 11) return false;
    const int64_t big = (int64_t)ceil_div(d.M, 128) * ceil_div(d.N, 128) * d.splitk * nbatch;
    const int kspan = ceil_div(d.K, d.splitk);
    const bool shape_ok = ok22 && d.M >= 128 && (d.N % 128 == 0 || d.N >= 512);
    static const int big1 = [] { const char* e = getenv("RALF_GEMM_BIG"); return e ? atoi(e) : 192; }();   // tuning aid (tools/knob_sweep.sh)
    return shape_ok && ((d.splitk == 1 && big >= big1) || (kspan >= 1024 && big >= 512));
}

inline int gemm_env_tile() { static const int v = [] { const char* e = getenv("RALF_GEMM_TILE"); return e ? atoi(e) : 0; }(); return v; }   // tuning / test aid: 22 / 11
inline int gemm_env_glds() { static const int v = [] { const char* e = getenv("RALF_GEMM_GLDS"); return e ? atoi(e) : 1; }(); return v; }   // 0: off (A/B runs, tests)
// the shape side of the 256 x 256-tile rule of launch_cfg (shared with ralf_gemm_filter_tile: a filtered product's slot lists are per column tile)
inline bool gemm_tile256_shape(const RalfGemmDesc& d, int nbatch) {
    static const int t256 = [] { const char* e = getenv("RALF_GEMM_TILE256"); return e ? atoi(e) : 400; }();   // tiles needed; 0 = off
    const int64_t n256 = (int64_t)ceil_div(d.M, 256) * ceil_div(d.N, 256) * nbatch;
    return t256 && d.splitk == 1 && d.M >= 256 && d.K >= 512 && n256 >= t256;
}

template <typename T, bool AK, bool BKC, int GATHER>
int launch_cfg(KParams& P, int nbatch, hipStream_t st) {
    const RalfGemmDesc& d = P.d;
    const int forced = gemm_env_tile();
    const bool use128 = gemm_use128(d, nbatch);
    const int64_t big = (int64_t)ceil_div(d.M, 128) * ceil_div(d.N, 128) * d.splitk * nbatch;
    const int kspan = ceil_div(d.K, d.splitk);
    if (forced == 22 && use128) return launch_epi<T, AK, BKC, GATHER, 2, 2, 8>(P, nbatch, st);
    if (forced == 11) return launch_epi<T, AK, BKC, GATHER, 1, 1, 4>(P, nbatch, st);
    if constexpr (GATHER == 3 && AK && BKC && sizeof(T) == 2) {
        // DIRECT-TO-LDS main loop for the aligned NT products on 128x128 tiles (forward linear layers / 1x1 convolutions): bit-identical to
        // the register-staged kernel.  Measured on MI355X -- tools/gemm_lab.hip (back-to-back launches of one shape, interleaved A/B):
        // 5-10 % faster on 17 of the model's 19 NT shapes, 8192^3 1293 -> 1189 us = 925 TFLOP/s; INSIDE the graph-replayed step (windowed
        // rocprofv3 A/B, profiles/r03*): 128x128 NT -5 %, 64x64 tiles +4-27 % (slower), NN / the im2col gathers through the same ring
        // neutral or slower (the padded register-staged images read with fewer conflicts than the XOR-swizzled transpose reads), the
        // grouped weight gradients (TN, 128x128) -13 %.  So: 128x128 NT here, the grouped kernel in gemm_bf16.hip, nothing else.
        // Three stages (one workgroup per CU) where a launch has at most one workgroup per CU anyway and a long reduction to pipeline.
        const int glds = gemm_env_glds();
        if (glds && use128 && !d.kseg && !d.bnb_part && !forced) {
            // 256 x 256 tiles (8 waves of 128 x 64, two 64-KiB stages) where at least ~1.5 of them exist per CU and the reduction is long: the operand
            // feed of a CU tops out at ~20-23 B/clk whatever the loop looks like (profiles/r06_gemm_lab_2.txt, r06_l2_feed_bench.txt), i.e. a tile's flops per loaded
            // byte set the rate -- 8192^3: 929 -> 1167 TFLOP/s, the two-stage k-NN's coarse pass (1024 x 61548 x 1792): 266 -> 230 us.  Plain
            // epilogues only (the model's own products have too few such tiles: they lose on 256-row tiles, tools/gemm_lab.hip) and the threshold filter
            // of the k-NN's coarse pass (no output at all: 260 -> ... us; its slot lists are per 256-column tile then, ralf_gemm_filter_tile()).
            const bool plain = !d.colstats && !d.flt_list && !d.C2 && d.act != RALF_ACT_GELU && d.aux_mode != RALF_AUX_GELU_GRAD && !d.atomic_out && d.drop_p == 0.f && !d.aux;
            if (gemm_tile256_shape(d, nbatch)) {
                if (plain) return launch<T, AK, BKC, 5, 4, 4, 0, 8>(P, nbatch, st);
                if (d.flt_list) return launch<T, AK, BKC, 5, 4, 4, 4, 8>(P, nbatch, st);
            }
            if (big <= 256 && kspan >= 1024) return launch_epi<T, AK, BKC, 6, 2, 2, 8>(P, nbatch, st);
            return launch_epi<T, AK, BKC, 5, 2, 2, 8>(P, nbatch, st);
        }
    }
    if (use128) return launch_epi<T, AK, BKC, GATHER, 2, 2, 8>(P, nbatch, st);
    return launch_epi<T, AK, BKC, GATHER, 1, 1, 4>(P, nbatch, st);
}

// products with an A-operand transform (RalfGemmDesc.at_*): the tile rule of launch_cfg, register-staged loaders only, and the two epilogues
// their callers use (plain / column statistics for the forward, the BatchNorm-backward reductions for the data gradient)
template <typename T, bool BKC, int G>
int launch_at(KParams& P, int nbatch, hipStream_t st) {
    const RalfGemmDesc& d = P.d;
    const bool use128 = gemm_use128(d, nbatch);   // (the tile rule of launch_cfg)
    if (d.bnb_part) {
        if constexpr (G == 8) {
            if (use128) return launch<T, true, BKC, G, 2, 2, 3, 8>(P, nbatch, st);
            return launch<T, true, BKC, G, 1, 1, 3, 4>(P, nbatch, st);
        } else { ralf::set_error("gemm: at_mode 1 with bnb_*: not built"); return RALF_ERR_INVALID; }
    }
    if (use128) return launch<T, true, BKC, G, 2, 2, 0, 8>(P, nbatch, st);
    return launch<T, true, BKC, G, 1, 1, 0, 4>(P, nbatch, st);
}

template <typename T>
int dispatch(KParams& P, int nbatch, hipStream_t st) {
    const RalfGemmDesc& d = P.d;
    const int key = (d.a_kcontig ? 4 : 0) | (d.b_kcontig ? 2 : 0);
    if (d.gather == 1) {
        if (key != 6) { ralf::set_error("gemm: gather=1 needs A and B k-contiguous"); return RALF_ERR_INVALID; }
        if constexpr (sizeof(T) == 2) {
            // data gradient of a stride-2 convolution: parity classes (gemm_body GATHER 14) where the tile rule picks 128 x 128 tiles anyway
            const bool plain = !d.colstats && !d.flt_list && !d.C2 && d.act != RALF_ACT_GELU && d.aux_mode != RALF_AUX_GELU_GRAD && !d.atomic_out && d.drop_p == 0.f && !d.aux;
            if (P.tapuni && P.par && plain && gemm_use128(d, nbatch)) {
                if (d.bnb_part) return launch<T, true, true, 14, 2, 2, 3, 8>(P, nbatch, st);
                return launch<T, true, true, 14, 2, 2, 0, 8>(P, nbatch, st);
            }
        }
        if constexpr (sizeof(T) == 2) {   // 3 x 3 / stride 1: the tile's input patch resident in LDS (gemm_body GATHER 15; ralf_gemm picked the tile)
            const bool plain = !d.flt_list && !d.C2 && d.act != RALF_ACT_GELU && d.aux_mode != RALF_AUX_GELU_GRAD && !d.atomic_out && d.drop_p == 0.f && !d.aux;
            if (P.tapuni && P.patch && plain && P.vec_epi >= 2) {
                const bool bnb = d.bnb_part != nullptr;
                if (P.patch == 1) return bnb ? launch<T, true, true, 15, 2, 2, 3, 8>(P, nbatch, st) : launch<T, true, true, 15, 2, 2, 0, 8>(P, nbatch, st);
                if (P.patch == 2) return bnb ? launch<T, true, true, 15, 4, 2, 3, 8, 4>(P, nbatch, st) : launch<T, true, true, 15, 4, 2, 0, 8, 4>(P, nbatch, st);
                return bnb ? launch<T, true, true, 15, 4, 1, 3, 8, 4>(P, nbatch, st) : launch<T, true, true, 15, 4, 1, 0, 8, 4>(P, nbatch, st);
            }
        }
        if (P.tapuni) return launch_cfg<T, true, true, 1>(P, nbatch, st);
        return launch_cfg<T, true, true, 4>(P, nbatch, st);
    }
    if (d.gather == 2) {
        if (key != 0) { ralf::set_error("gemm: gather=2 needs A and B row-contiguous"); return RALF_ERR_INVALID; }
        return launch_cfg<T, false, false, 2>(P, nbatch, st);
    }
    if constexpr (sizeof(T) == 4) {   // fp32 parity mode: few rows x K % 256 == 0 (a decode step's linear layers)
        static const int skinny_rows32 = [] { const char* e = getenv("RALF_GEMM_SKINNY_ROWS"); return e ? atoi(e) : 512; }();
        if (P.fast && key == 6 && d.M <= skinny_rows32 && d.K % 256 == 0 && nbatch == 1 && d.splitk == 1 && !d.colstats && !d.bnb_part && !d.kseg && !d.atomic_out && !d.flt_list && !d.ln_g) {
            const bool lvl2 = d.C2 || d.act == RALF_ACT_GELU || d.aux_mode == RALF_AUX_GELU_GRAD;
            const bool lvl1 = d.drop_p > 0.f || d.aux;
            return lvl2 ? launch_skinny<T, 2>(P, st) : lvl1 ? launch_skinny<T, 1>(P, st) : launch_skinny<T, 0>(P, st);
        }
    }
    if constexpr (sizeof(T) == 2) {   // bf16: the interior fast path is its own (leaner) set of kernels; fp32 is the parity mode
        static const int skinny_rows = [] { const char* e = getenv("RALF_GEMM_SKINNY_ROWS"); return e ? atoi(e) : 512; }();   // 0 = off (A/B runs)
        // (few rows AND a weight-sized second operand: one wave per 32 x 32 tile re-reads both operands per tile -- a 64..512-row product against the
        //  61548-row k-NN index ran 3-4x slower here than on the tiled kernel, profiles/r06_knn_route_sweep_before.txt)
        if (P.fast && key == 6 && d.M <= skinny_rows && d.N <= 8192 && nbatch == 1 && d.splitk == 1 && !d.colstats && !d.bnb_part && !d.kseg && !d.atomic_out && !d.flt_list) {
            const bool lvl2 = d.C2 || d.act == RALF_ACT_GELU || d.aux_mode == RALF_AUX_GELU_GRAD;
            const bool lvl1 = d.drop_p > 0.f || d.aux;
            return lvl2 ? launch_skinny<T, 2>(P, st) : lvl1 ? launch_skinny<T, 1>(P, st) : launch_skinny<T, 0>(P, st);
        }
        if (d.ln_g) { ralf::set_error("gemm: ln_* is part of the few-row kernel (M <= RALF_GEMM_SKINNY_ROWS, N <= 8192), which does not take this product"); return RALF_ERR_INVALID; }
        if (P.fast) {
            switch (key) {
                case 6: return launch_cfg<T, true, true, 3>(P, nbatch, st);
                case 4: return launch_cfg<T, true, false, 3>(P, nbatch, st);
                case 0: return launch_cfg<T, false, false, 3>(P, nbatch, st);
                default: return launch_cfg<T, false, true, 3>(P, nbatch, st);
            }
        }
    }
    switch (key) {
        case 6: return launch_cfg<T, true, true, 0>(P, nbatch, st);
        case 4: return launch_cfg<T, true, false, 0>(P, nbatch, st);
        case 0: return launch_cfg<T, false, false, 0>(P, nbatch, st);
        default: return launch_cfg<T, false, true, 0>(P, nbatch, st);
    }
}


}  // namespace

// one translation unit per element type (parallel builds): gemm_f32.hip / gemm_bf16.hip define these
int ralf_gemm_dispatch_f32(void* kparams, int nbatch, hipStream_t st);
int ralf_gemm_dispatch_bf16(void* kparams, int nbatch, hipStream_t st);
int ralf_gemm_reduce_f32(void* kparams, int nbatch, int blocks, hipStream_t st);
int ralf_gemm_reduce_bf16(void* kparams, int nbatch, int blocks, hipStream_t st);
int ralf_gemm_grouped_bf16(const void* jobs, int njobs, void* workspace, size_t workspace_bytes, hipStream_t st);
int ralf_gemm_dispatch_at(void* kparams, int nbatch, hipStream_t st);   // gemm_at.hip: the operand-transform kernels (their own translation unit)
