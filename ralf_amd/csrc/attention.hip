// Fused scaled-dot-product attention forward / backward for the short sequences of RALF
// (Sq <= 256, Sk <= ~700, head dim 32 or 64) on gfx950.
//
// Replaces the attention core of nn.MultiheadAttention inside nn.TransformerEncoderLayer /
// nn.TransformerDecoderLayer (image2layout/train/models/retrieval_augmented_autoreg.py:116-126,
// common/common.py:25-34,116-123, fid/model.py:26-33) and of Attention.forward
// (common/attention.py:62-70): softmax(scale * Q K^T + causal/key-padding mask) -> dropout -> @ V.
//
// Round-1 structure (correctness first, fp32 VALU math, scores never touch HBM):
//   forward / dQ : one thread per query row, the workgroup's 4 waves split each 128-key tile staged
//                  in LDS (K/V rows are wave-uniform -> LDS broadcast reads), online softmax per
//                  32-key slice, the 4 partial states merged through LDS.
//   dK/dV        : one thread per key row, waves split each 128-query tile (Q, dO, lse, delta in LDS).
// P is recomputed from (Q, K, lse) in the backward kernels; the dropout mask is regenerated from the
// counter-based RNG, so nothing but lse/delta [B,H,Sq] is saved.
#include "common.h"

namespace {
typedef __bf16 bf16;

template <typename T> struct V4;
template <> struct V4<float> {
    static __device__ __forceinline__ float4 load(const float* p) { return *reinterpret_cast<const float4*>(p); }
    static __device__ __forceinline__ void store(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
};
template <> struct V4<bf16> {
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ float4 load(const bf16* p) { bf16x4 t = *reinterpret_cast<const bf16x4*>(p); return make_float4((float)t[0], (float)t[1], (float)t[2], (float)t[3]); }
    static __device__ __forceinline__ void store(bf16* p, float4 v) { bf16x4 t; t[0] = (bf16)v.x; t[1] = (bf16)v.y; t[2] = (bf16)v.z; t[3] = (bf16)v.w; *reinterpret_cast<bf16x4*>(p) = t; }
};


constexpr int TILE = 128;  // keys (or queries) staged per block iteration
constexpr int SL = 32;     // slice of the tile owned by one wave

// stage rows [t0, t0+TILE) of a [S, H*dh]-strided operand (head h) into LDS as fp32 [TILE][DH]; optional scale
template <typename T, int DH>
__device__ __forceinline__ void stage_rows(float* dst, const T* base, int64_t row_stride, int t0, int S, float mul) {
    constexpr int VPR = DH / 4;
    // only the 32-row slices of waves that have work are touched: zero-fill up to the next slice boundary
    // (masked entries multiply these rows by p = 0, so they must be finite), skip the rest of the tile
    const int nrows = min(TILE, S - t0), nfill = min(TILE, (nrows + SL - 1) / SL * SL);  // (bwd kernels walk whole slices)
    for (int e = threadIdx.x; e < nfill * VPR; e += 256) {
        const int r = e / VPR, c = e % VPR;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < nrows) v = V4<T>::load(base + (int64_t)(t0 + r) * row_stride + c * 4);
        v.x *= mul; v.y *= mul; v.z *= mul; v.w *= mul;
        *reinterpret_cast<float4*>(dst + r * DH + c * 4) = v;
    }
}

template <typename T, int DH>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const RalfAttnDesc d) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;
    float* Vs = smem + TILE * DH;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.z, h = blockIdx.y, qi = blockIdx.x * 64 + lane;
    const bool qok = qi < d.Sq;
    const T* Qp = (const T*)d.q + b * d.q_bs + (int64_t)h * DH;
    const T* Kp = (const T*)d.k + b * d.k_bs + (int64_t)h * DH;
    const T* Vp = (const T*)d.v + b * d.v_bs + (int64_t)h * DH;
    const uint8_t* kpm = d.kpm ? d.kpm + (int64_t)b * (d.kpm_bs ? d.kpm_bs : (int64_t)d.Sk) : nullptr;
    const uint64_t seed = d.p_drop > 0.f ? (uint64_t)d.seed[0] : 0;
    const uint32_t thr = attn_thr16(d.p_drop);
    const float inv_keep = 1.f / (1.f - d.p_drop);

    float q[DH], o[DH];
#pragma unroll
    for (int c = 0; c < DH / 4; ++c) {
        float4 v = qok ? V4<T>::load(Qp + (int64_t)qi * d.q_rs + c * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        q[4 * c] = v.x * d.scale; q[4 * c + 1] = v.y * d.scale; q[4 * c + 2] = v.z * d.scale; q[4 * c + 3] = v.w * d.scale;
        o[4 * c] = o[4 * c + 1] = o[4 * c + 2] = o[4 * c + 3] = 0.f;
    }
    float m = -__builtin_inff(), l = 0.f;

    for (int t0 = 0; t0 < d.Sk; t0 += TILE) {
        stage_rows<T, DH>(Ks, Kp, d.k_rs, t0, d.Sk, 1.f);
        stage_rows<T, DH>(Vs, Vp, d.v_rs, t0, d.Sk, 1.f);
        __syncthreads();
        const int kbase = t0 + wave * SL;
        if (kbase < d.Sk && !(d.causal && kbase > blockIdx.x * 64 + 63)) {
            // online softmax in sub-slices of SUB keys: small static register arrays (no spills), one
            // rescale of the running output per sub-slice
            constexpr int SUB = DH == 64 ? 4 : 8;
#pragma unroll 1
            for (int j0 = 0; j0 < SL; j0 += SUB) {
                if (kbase + j0 >= d.Sk) break;
                float s[SUB];
                float mt = -__builtin_inff();
#pragma unroll
                for (int j = 0; j < SUB; ++j) {
                    const int key = kbase + j0 + j;
                    const float* kr = Ks + (wave * SL + j0 + j) * DH;
                    float acc = 0.f;
#pragma unroll
                    for (int c = 0; c < DH / 4; ++c) {
                        const float4 kv = *reinterpret_cast<const float4*>(kr + c * 4);
                        acc += q[4 * c] * kv.x + q[4 * c + 1] * kv.y + q[4 * c + 2] * kv.z + q[4 * c + 3] * kv.w;
                    }
                    const bool masked = key >= d.Sk || (d.causal && key > qi) || (kpm && kpm[key]);
                    s[j] = masked ? -__builtin_inff() : acc;
                    mt = fmaxf(mt, s[j]);
                }
                const float mn = fmaxf(m, mt);
                if (mn > -__builtin_inff()) {
                    const float corr = __expf(m - mn);
                    l *= corr;
#pragma unroll
                    for (int c = 0; c < DH; ++c) o[c] *= corr;
#pragma unroll
                    for (int j = 0; j < SUB; ++j) {
                        const float p = __expf(s[j] - mn);
                        l += p;
                        float pd = p;
                        if (d.p_drop > 0.f) {
                            const uint32_t rk = attn_rowkey(seed, d.call_id, ((uint64_t)b * d.H + h) * d.Sq + qi);
                            pd = attn_keep(rk, (uint32_t)(kbase + j0 + j), thr) ? p * inv_keep : 0.f;
                        }
                        const float* vr = Vs + (wave * SL + j0 + j) * DH;
#pragma unroll
                        for (int c = 0; c < DH / 4; ++c) {
                            const float4 vv = *reinterpret_cast<const float4*>(vr + c * 4);
                            o[4 * c] += pd * vv.x; o[4 * c + 1] += pd * vv.y; o[4 * c + 2] += pd * vv.z; o[4 * c + 3] += pd * vv.w;
                        }
                    }
                    m = mn;
                }
            }
        }
        __syncthreads();
    }
    // merge the 4 per-wave states: buf[c][wave*64+lane], c in [0, DH+2)
    float* buf = smem;
#pragma unroll
    for (int c = 0; c < DH; ++c) buf[c * 256 + threadIdx.x] = o[c];
    buf[DH * 256 + threadIdx.x] = m;
    buf[(DH + 1) * 256 + threadIdx.x] = l;
    __syncthreads();
    if (wave == 0 && qok) {
        float mm = -__builtin_inff();
#pragma unroll
        for (int w = 0; w < 4; ++w) mm = fmaxf(mm, buf[DH * 256 + w * 64 + lane]);
        float f[4], ll = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float mw = buf[DH * 256 + w * 64 + lane];
            f[w] = mw > -__builtin_inff() ? __expf(mw - mm) : 0.f;
            ll += f[w] * buf[(DH + 1) * 256 + w * 64 + lane];
        }
        const float inv = 1.f / ll;
        T* Op = (T*)d.o + b * d.o_bs + (int64_t)qi * d.o_rs + (int64_t)h * DH;
#pragma unroll
        for (int c = 0; c < DH / 4; ++c) {
            float r[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float a = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) a += f[w] * buf[(4 * c + i) * 256 + w * 64 + lane];
                r[i] = a * inv;
            }
            V4<T>::store(Op + c * 4, make_float4(r[0], r[1], r[2], r[3]));
        }
        if (d.lse) d.lse[((int64_t)b * d.H + h) * d.Sq + qi] = mm + __logf(ll);
    }
}

// dQ (and delta = rowsum(dO * O)): same thread/tile structure as the forward kernel
template <typename T, int DH>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const RalfAttnDesc d) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;
    float* Vs = smem + TILE * DH;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.z, h = blockIdx.y, qi = blockIdx.x * 64 + lane;
    const bool qok = qi < d.Sq;
    const T* Qp = (const T*)d.q + b * d.q_bs + (int64_t)h * DH;
    const T* Kp = (const T*)d.k + b * d.k_bs + (int64_t)h * DH;
    const T* Vp = (const T*)d.v + b * d.v_bs + (int64_t)h * DH;
    const T* Op = (const T*)d.o + b * d.o_bs + (int64_t)h * DH;
    const T* dOp = (const T*)d.dout + b * d.do_bs + (int64_t)h * DH;
    const uint8_t* kpm = d.kpm ? d.kpm + (int64_t)b * (d.kpm_bs ? d.kpm_bs : (int64_t)d.Sk) : nullptr;
    const uint64_t seed = d.p_drop > 0.f ? (uint64_t)d.seed[0] : 0;
    const uint32_t thr = attn_thr16(d.p_drop);
    const float inv_keep = 1.f / (1.f - d.p_drop);
    const int64_t stat = ((int64_t)b * d.H + h) * d.Sq + qi;

    float q[DH], go[DH], dq[DH];
    float delta = 0.f;
#pragma unroll
    for (int c = 0; c < DH / 4; ++c) {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 v = qok ? V4<T>::load(Qp + (int64_t)qi * d.q_rs + c * 4) : z;
        const float4 g = qok ? V4<T>::load(dOp + (int64_t)qi * d.do_rs + c * 4) : z;
        const float4 ov = qok ? V4<T>::load(Op + (int64_t)qi * d.o_rs + c * 4) : z;
        q[4 * c] = v.x * d.scale; q[4 * c + 1] = v.y * d.scale; q[4 * c + 2] = v.z * d.scale; q[4 * c + 3] = v.w * d.scale;
        go[4 * c] = g.x; go[4 * c + 1] = g.y; go[4 * c + 2] = g.z; go[4 * c + 3] = g.w;
        delta += g.x * ov.x + g.y * ov.y + g.z * ov.z + g.w * ov.w;
        dq[4 * c] = dq[4 * c + 1] = dq[4 * c + 2] = dq[4 * c + 3] = 0.f;
    }
    const float lse = qok ? d.lse[stat] : 0.f;
    if (wave == 0 && qok) d.delta[stat] = delta;

    for (int t0 = 0; t0 < d.Sk; t0 += TILE) {
        stage_rows<T, DH>(Ks, Kp, d.k_rs, t0, d.Sk, 1.f);
        stage_rows<T, DH>(Vs, Vp, d.v_rs, t0, d.Sk, 1.f);
        __syncthreads();
        const int kbase = t0 + wave * SL;
        if (kbase < d.Sk && !(d.causal && kbase > blockIdx.x * 64 + 63)) {
#pragma unroll 2
            for (int j = 0; j < SL; ++j) {
                const int key = kbase + j;
                const float* kr = Ks + (wave * SL + j) * DH;
                const float* vr = Vs + (wave * SL + j) * DH;
                float s = 0.f, dp = 0.f;
#pragma unroll
                for (int c = 0; c < DH / 4; ++c) {
                    const float4 kv = *reinterpret_cast<const float4*>(kr + c * 4);
                    const float4 vv = *reinterpret_cast<const float4*>(vr + c * 4);
                    s += q[4 * c] * kv.x + q[4 * c + 1] * kv.y + q[4 * c + 2] * kv.z + q[4 * c + 3] * kv.w;
                    dp += go[4 * c] * vv.x + go[4 * c + 1] * vv.y + go[4 * c + 2] * vv.z + go[4 * c + 3] * vv.w;
                }
                const bool masked = key >= d.Sk || (d.causal && key > qi) || (kpm && kpm[key]);
                const float p = masked ? 0.f : __expf(s - lse);
                if (d.p_drop > 0.f) {
                    const uint32_t rk = attn_rowkey(seed, d.call_id, ((uint64_t)b * d.H + h) * d.Sq + qi);
                    dp = attn_keep(rk, (uint32_t)key, thr) ? dp * inv_keep : 0.f;
                }
                const float ds = p * (dp - delta);
#pragma unroll
                for (int c = 0; c < DH / 4; ++c) {
                    const float4 kv = *reinterpret_cast<const float4*>(kr + c * 4);
                    dq[4 * c] += ds * kv.x; dq[4 * c + 1] += ds * kv.y; dq[4 * c + 2] += ds * kv.z; dq[4 * c + 3] += ds * kv.w;
                }
            }
        }
        __syncthreads();
    }
    float* buf = smem;
#pragma unroll
    for (int c = 0; c < DH; ++c) buf[c * 256 + threadIdx.x] = dq[c];
    __syncthreads();
    if (wave == 0 && qok) {
        T* dQp = (T*)d.dq + b * d.dq_bs + (int64_t)qi * d.dq_rs + (int64_t)h * DH;
#pragma unroll
        for (int c = 0; c < DH / 4; ++c) {
            float r[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                r[i] = d.scale * (buf[(4 * c + i) * 256 + lane] + buf[(4 * c + i) * 256 + 64 + lane] + buf[(4 * c + i) * 256 + 128 + lane] + buf[(4 * c + i) * 256 + 192 + lane]);
            V4<T>::store(dQp + c * 4, make_float4(r[0], r[1], r[2], r[3]));
        }
    }
}

// dK, dV: one thread per key; waves split each 128-query tile
template <typename T, int DH>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const RalfAttnDesc d) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Qs = smem;                       // scaled q rows
    float* Gs = smem + TILE * DH;           // dO rows
    float* Ls = smem + 2 * TILE * DH;       // lse
    float* Ds = Ls + TILE;                  // delta
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.z, h = blockIdx.y, kj = blockIdx.x * 64 + lane;
    const bool kok = kj < d.Sk;
    const T* Qp = (const T*)d.q + b * d.q_bs + (int64_t)h * DH;
    const T* Kp = (const T*)d.k + b * d.k_bs + (int64_t)h * DH;
    const T* Vp = (const T*)d.v + b * d.v_bs + (int64_t)h * DH;
    const T* dOp = (const T*)d.dout + b * d.do_bs + (int64_t)h * DH;
    const bool kmasked = !kok || (d.kpm && d.kpm[(int64_t)b * (d.kpm_bs ? d.kpm_bs : (int64_t)d.Sk) + kj]);
    const uint64_t seed = d.p_drop > 0.f ? (uint64_t)d.seed[0] : 0;
    const uint32_t thr = attn_thr16(d.p_drop);
    const float inv_keep = 1.f / (1.f - d.p_drop);

    float k[DH], v[DH], dk[DH], dv[DH];
#pragma unroll
    for (int c = 0; c < DH / 4; ++c) {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 kv = kok ? V4<T>::load(Kp + (int64_t)kj * d.k_rs + c * 4) : z;
        const float4 vv = kok ? V4<T>::load(Vp + (int64_t)kj * d.v_rs + c * 4) : z;
        k[4 * c] = kv.x; k[4 * c + 1] = kv.y; k[4 * c + 2] = kv.z; k[4 * c + 3] = kv.w;
        v[4 * c] = vv.x; v[4 * c + 1] = vv.y; v[4 * c + 2] = vv.z; v[4 * c + 3] = vv.w;
        dk[4 * c] = dk[4 * c + 1] = dk[4 * c + 2] = dk[4 * c + 3] = 0.f;
        dv[4 * c] = dv[4 * c + 1] = dv[4 * c + 2] = dv[4 * c + 3] = 0.f;
    }
    const int64_t stat0 = ((int64_t)b * d.H + h) * d.Sq;
    for (int t0 = 0; t0 < d.Sq; t0 += TILE) {
        stage_rows<T, DH>(Qs, Qp, d.q_rs, t0, d.Sq, d.scale);
        stage_rows<T, DH>(Gs, dOp, d.do_rs, t0, d.Sq, 1.f);
        if (threadIdx.x < TILE) {
            const int qi = t0 + threadIdx.x;
            Ls[threadIdx.x] = qi < d.Sq ? d.lse[stat0 + qi] : 0.f;
            Ds[threadIdx.x] = qi < d.Sq ? d.delta[stat0 + qi] : 0.f;
        }
        __syncthreads();
        const int qbase = t0 + wave * SL;
        // causal: queries before the first key of this block see none of its keys
        if (qbase < d.Sq && !(d.causal && qbase + SL - 1 < blockIdx.x * 64)) {
#pragma unroll 2
            for (int j = 0; j < SL; ++j) {
                const int qi = qbase + j;
                const float* qr = Qs + (wave * SL + j) * DH;
                const float* gr = Gs + (wave * SL + j) * DH;
                float s = 0.f, dp = 0.f;
#pragma unroll
                for (int c = 0; c < DH / 4; ++c) {
                    const float4 qv = *reinterpret_cast<const float4*>(qr + c * 4);
                    const float4 gv = *reinterpret_cast<const float4*>(gr + c * 4);
                    s += k[4 * c] * qv.x + k[4 * c + 1] * qv.y + k[4 * c + 2] * qv.z + k[4 * c + 3] * qv.w;
                    dp += v[4 * c] * gv.x + v[4 * c + 1] * gv.y + v[4 * c + 2] * gv.z + v[4 * c + 3] * gv.w;
                }
                const bool masked = kmasked || qi >= d.Sq || (d.causal && kj > qi);
                const float p = masked ? 0.f : __expf(s - Ls[wave * SL + j]);
                float pd = p;
                if (d.p_drop > 0.f) {
                    const uint32_t rk = attn_rowkey(seed, d.call_id, ((uint64_t)b * d.H + h) * d.Sq + qi);
                    const bool keep = attn_keep(rk, (uint32_t)kj, thr);
                    pd = keep ? p * inv_keep : 0.f;
                    dp = keep ? dp * inv_keep : 0.f;
                }
                const float ds = p * (dp - Ds[wave * SL + j]);
#pragma unroll
                for (int c = 0; c < DH / 4; ++c) {
                    const float4 qv = *reinterpret_cast<const float4*>(qr + c * 4);
                    const float4 gv = *reinterpret_cast<const float4*>(gr + c * 4);
                    dk[4 * c] += ds * qv.x; dk[4 * c + 1] += ds * qv.y; dk[4 * c + 2] += ds * qv.z; dk[4 * c + 3] += ds * qv.w;
                    dv[4 * c] += pd * gv.x; dv[4 * c + 1] += pd * gv.y; dv[4 * c + 2] += pd * gv.z; dv[4 * c + 3] += pd * gv.w;
                }
            }
        }
        __syncthreads();
    }
    float* buf = smem;  // [DH][256]
    T* dKp = (T*)d.dk + b * d.dk_bs + (int64_t)kj * d.dk_rs + (int64_t)h * DH;
    T* dVp = (T*)d.dv + b * d.dv_bs + (int64_t)kj * d.dv_rs + (int64_t)h * DH;
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int c = 0; c < DH; ++c) buf[c * 256 + threadIdx.x] = pass == 0 ? dk[c] : dv[c];
        __syncthreads();
        if (wave == 0 && kok) {
            T* outp = pass == 0 ? dKp : dVp;
#pragma unroll
            for (int c = 0; c < DH / 4; ++c) {
                float r[4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    r[i] = buf[(4 * c + i) * 256 + lane] + buf[(4 * c + i) * 256 + 64 + lane] + buf[(4 * c + i) * 256 + 128 + lane] + buf[(4 * c + i) * 256 + 192 + lane];
                V4<T>::store(outp + c * 4, make_float4(r[0], r[1], r[2], r[3]));
            }
        }
        __syncthreads();
    }
}

// One query row per batch element (a KV-cached decode step in the fp32 parity mode: retrieval_augmented_autoreg.py:274-279, the reference's
// full-prefix call restricted to its last row), H * dh = 256: a workgroup per batch element, a lane per 4 consecutive columns of the 256-wide
// row (so a wave-load is one whole K or V row of all heads: 1 KB, coalesced), the 8 waves take blocks of 8 keys each -- 8 row loads in flight
// per lane, 64 KB per workgroup.  Pass 1: scores (dh / 4 lanes of a head reduce by shuffles) -> LDS; every wave reduces max / sum of its
// lanes' head over all keys; pass 2: o += p * V, the 8 partial rows merged through LDS.  The general kernel above gives a query row to ONE
// lane: with Sq = 1 it ran 1 lane in 64 and took 345-980 us per call on the decoder's 532-row memory at B = 256 (70 % of the fp32 decode).
constexpr int DQ_WAVES = 8;
// STREAM (long memories: the decoder's cross-attention, 279 MB of K / V per call at B = 256): the rows are read once per step -- streaming
// (non-temporal) loads, so that they do not push the step's weights out of the L2 / infinity cache on their way through (as in the bf16 block)
template <bool STREAM>
__device__ __forceinline__ float4 dq_load(const float* p) {
    if constexpr (STREAM) {
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
        return make_float4(v.x, v.y, v.z, v.w);
    } else {
        return *reinterpret_cast<const float4*>(p);
    }
}
template <bool STREAM>
__global__ __launch_bounds__(512) void attn_decode_f32_kernel(const RalfAttnDesc d) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sc = smem;                              // [Sk][H]
    float* part = smem + (size_t)d.Sk * d.H;       // [DQ_WAVES][256]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
    const int G = d.dh >> 2, head = lane / G, gl = lane - head * G;     // lanes per head, this lane's head, its place in the group
    const float* Kp = (const float*)d.k + (int64_t)b * d.k_bs + lane * 4;
    const float* Vp = (const float*)d.v + (int64_t)b * d.v_bs + lane * 4;
    const uint8_t* kpm = d.kpm ? d.kpm + (int64_t)b * (d.kpm_bs ? d.kpm_bs : (int64_t)d.Sk) : nullptr;
    float4 q = *reinterpret_cast<const float4*>((const float*)d.q + (int64_t)b * d.q_bs + lane * 4);
    q.x *= d.scale; q.y *= d.scale; q.z *= d.scale; q.w *= d.scale;
    const int Sk = d.Sk, last = Sk - 1;
    for (int key0 = wave * 8; key0 < Sk; key0 += DQ_WAVES * 8) {
        float4 kv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) kv[u] = dq_load<STREAM>(Kp + (int64_t)min(key0 + u, last) * d.k_rs);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            float s = q.x * kv[u].x + q.y * kv[u].y + q.z * kv[u].z + q.w * kv[u].w;
            s += __shfl_xor(s, 1);
            s += __shfl_xor(s, 2);
            s += __shfl_xor(s, 4);
            if (G == 16) s += __shfl_xor(s, 8);
            const int key = key0 + u;
            if (gl == 0 && key < Sk) sc[key * d.H + head] = (kpm && kpm[key]) ? -__builtin_inff() : s;
        }
    }
    __syncthreads();
    // max and sum of this lane's head over all keys (the G lanes of the group stride the keys)
    float m = -__builtin_inff();
    for (int key = gl; key < Sk; key += G) m = fmaxf(m, sc[key * d.H + head]);
    m = fmaxf(m, __shfl_xor(m, 1));
    m = fmaxf(m, __shfl_xor(m, 2));
    m = fmaxf(m, __shfl_xor(m, 4));
    if (G == 16) m = fmaxf(m, __shfl_xor(m, 8));
    float l = 0.f;
    for (int key = gl; key < Sk; key += G) l += __expf(sc[key * d.H + head] - m);
    l += __shfl_xor(l, 1);
    l += __shfl_xor(l, 2);
    l += __shfl_xor(l, 4);
    if (G == 16) l += __shfl_xor(l, 8);
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int key0 = wave * 8; key0 < Sk; key0 += DQ_WAVES * 8) {
        float4 vv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) vv[u] = dq_load<STREAM>(Vp + (int64_t)min(key0 + u, last) * d.v_rs);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int key = key0 + u;
            const float p = key < Sk ? __expf(sc[min(key, last) * d.H + head] - m) : 0.f;
            o.x += p * vv[u].x; o.y += p * vv[u].y; o.z += p * vv[u].z; o.w += p * vv[u].w;
        }
    }
    *reinterpret_cast<float4*>(part + wave * 256 + lane * 4) = o;
    __syncthreads();
    if (wave == 0) {
        float4 a = *reinterpret_cast<const float4*>(part + lane * 4);
#pragma unroll
        for (int w = 1; w < DQ_WAVES; ++w) {
            const float4 t = *reinterpret_cast<const float4*>(part + w * 256 + lane * 4);
            a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
        }
        const float inv = 1.f / l;
        *reinterpret_cast<float4*>((float*)d.o + (int64_t)b * d.o_bs + lane * 4) = make_float4(a.x * inv, a.y * inv, a.z * inv, a.w * inv);
    }
}
constexpr int DQ_MAX_KEYS = 3072;   // scores [Sk][8] + the partial rows within 160 KB of LDS (H = 8: 96 KB + 8 KB)

template <int DH> constexpr size_t fwd_lds() { return sizeof(float) * (size_t)((2 * TILE * DH) > ((DH + 2) * 256) ? (2 * TILE * DH) : ((DH + 2) * 256)); }
template <int DH> constexpr size_t dkv_lds() { return sizeof(float) * (size_t)(2 * TILE * DH + 2 * TILE); }

int validate(const RalfAttnDesc* d, bool bwd) {
    RALF_REQUIRE(d, "attention: null descriptor");
    RALF_REQUIRE(d->q && d->k && d->v && d->o, "attention: null operand");
    RALF_REQUIRE(d->B > 0 && d->H > 0 && d->Sq > 0 && d->Sk > 0, "attention: empty problem");
    RALF_REQUIRE(d->dh == 32 || d->dh == 64, "attention: head dim %d not in {32, 64}", d->dh);
    RALF_REQUIRE(d->dtype == RALF_F32 || d->dtype == RALF_BF16, "attention: dtype %d", d->dtype);
    RALF_REQUIRE(d->p_drop >= 0.f && d->p_drop < 1.f && (d->p_drop == 0.f || d->seed), "attention: dropout needs 0 <= p < 1 and a seed");
    if (bwd) RALF_REQUIRE(d->dout && d->dq && d->dk && d->dv && d->lse && d->delta, "attention_bwd: null gradient/statistics pointer");
    return RALF_OK;
}

template <typename K>
void allow_lds(K kernel, size_t bytes) {  // > 64 KiB of dynamic LDS must be opted into (160 KiB per CU on gfx950)
    if (bytes > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

bool decode_f32_fits(const RalfAttnDesc& d) {
    return d.Sq == 1 && d.H * d.dh == 256 && !d.causal && d.p_drop == 0.f && !d.lse && d.Sk <= DQ_MAX_KEYS &&
           ((d.k_rs | d.v_rs | d.k_bs | d.v_bs | d.q_bs | d.o_bs) & 3) == 0 && (((uintptr_t)d.q | (uintptr_t)d.k | (uintptr_t)d.v | (uintptr_t)d.o) & 15) == 0;
}
int run_decode_f32(const RalfAttnDesc& d, hipStream_t st) {
    const size_t lds = sizeof(float) * ((size_t)d.Sk * d.H + DQ_WAVES * 256);
    static bool allowed = false;
    if (!allowed) {
        const int most = (int)(sizeof(float) * ((size_t)DQ_MAX_KEYS * 8 + DQ_WAVES * 256));
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_decode_f32_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, most);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_decode_f32_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, most);
        allowed = true;
    }
    if (d.Sk > 128) hipLaunchKernelGGL(attn_decode_f32_kernel<true>, dim3(d.B), dim3(512), lds, st, d);
    else hipLaunchKernelGGL(attn_decode_f32_kernel<false>, dim3(d.B), dim3(512), lds, st, d);
    return ralf::check_launch("attention_decode_f32");
}
template <typename T, int DH>
int run_fwd(const RalfAttnDesc& d, hipStream_t st) {
    allow_lds(attn_fwd_kernel<T, DH>, fwd_lds<DH>());
    hipLaunchKernelGGL((attn_fwd_kernel<T, DH>), dim3(ceil_div(d.Sq, 64), d.H, d.B), dim3(256), fwd_lds<DH>(), st, d);
    return ralf::check_launch("attention_fwd");
}
template <typename T, int DH>
int run_bwd(const RalfAttnDesc& d, hipStream_t st) {
    allow_lds(attn_bwd_dq_kernel<T, DH>, fwd_lds<DH>());
    allow_lds(attn_bwd_dkv_kernel<T, DH>, dkv_lds<DH>());
    hipLaunchKernelGGL((attn_bwd_dq_kernel<T, DH>), dim3(ceil_div(d.Sq, 64), d.H, d.B), dim3(256), fwd_lds<DH>(), st, d);
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, DH>), dim3(ceil_div(d.Sk, 64), d.H, d.B), dim3(256), dkv_lds<DH>(), st, d);
    return ralf::check_launch("attention_bwd");
}
}  // namespace

// bf16 throughput mode: matrix-core kernels (attention_mfma.hip)
int ralf_attention_fwd_mfma(const RalfAttnDesc& d, hipStream_t st);
int ralf_attention_bwd_mfma(const RalfAttnDesc& d, hipStream_t st);

extern "C" int ralf_attention_fwd(const RalfAttnDesc* d, void* stream) {
    if (int rc = validate(d, false)) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (d->dtype == RALF_F32) {
        if (decode_f32_fits(*d)) return run_decode_f32(*d, st);
        return d->dh == 32 ? run_fwd<float, 32>(*d, st) : run_fwd<float, 64>(*d, st);
    }
    return ralf_attention_fwd_mfma(*d, st);
}

extern "C" int ralf_attention_bwd(const RalfAttnDesc* d, void* stream) {
    if (int rc = validate(d, true)) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (d->dtype == RALF_F32) return d->dh == 32 ? run_bwd<float, 32>(*d, st) : run_bwd<float, 64>(*d, st);
    return ralf_attention_bwd_mfma(*d, st);
}
