// bf16 matrix-core attention (forward, dQ, dK/dV) for gfx950 -- the throughput-mode path behind
// ralf_attention_fwd / ralf_attention_bwd (the fp32 parity mode keeps the VALU kernels of attention.hip).
//
// Everything is computed TRANSPOSED so the softmax index stays lane-local:
//   S^T[key][query] = K Q^T      v_mfma_f32_16x16x32_bf16, row operand = 16 key rows (LDS, ds_read_b128),
//                                column operand = 16 query rows (registers)
//   -> a lane holds ONE query (lane & 15) and 4 consecutive keys per 16-key block; row max / sum need
//      two xor-shuffles (lanes 16/32 apart), the running (m, l) and every rescale are lane-local.
//   O^T[d][query]  = V^T P^T     row operand = V^T via ds_read_b64_tr_b16 on the row-major V tile,
//                                column operand = P^T straight from the S^T accumulators: the MFMA k-slot
//                                8g+j is DEFINED as key {4g+j (j<4), 16+4g+j-4 (j>=4)} of a 32-key step, so
//                                no data movement is needed between the two products.
// Backward reuses the same layout: dQ^T = K^T dS^T (per-query kernel), dK^T = Q^T dS, dV^T = dO^T P
// (per-key kernel, lane = one key).  P is recomputed from (Q, K, lse); dropout masks come from the same
// counter-based generator as the VALU kernels, so forward/backward of either implementation agree.
#include "common.h"
#include "attn_core.h"

namespace {
typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))


// Workgroup -> (tile, head, batch): one linear grid, remapped so the row tiles of ONE (batch, head) get consecutive
// virtual ids on ONE XCD (hardware deals workgroups round-robin over the 8 XCDs): they stream the same K/V (or Q/dO)
// rows, which then come from that XCD's L2 instead of being fetched from HBM once per tile.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}
struct WgId { int tile, h, b; };
__device__ __forceinline__ WgId wg_id(int ntiles, int H) {
    const int vid = xcd_remap(blockIdx.x, gridDim.x);
    WgId w;
    w.tile = vid % ntiles;
    const int t = vid / ntiles;
    w.h = t % H; w.b = t / H;
    return w;
}

constexpr int KT = 64;  // rows of the streamed operand staged per iteration (keys in fwd/dQ, queries in dK/dV)

template <int DH> struct L {
    static constexpr int LD = DH + 8;  // LDS row stride in elements (16-byte pad: conflict-light b128 / tr reads)
};

// stage rows [t0, t0+KT) of a [S, *]-strided bf16 operand (head offset applied) into LDS [KT][LD]; rows >= S zero
template <int DH>
__device__ __forceinline__ void stage(bf16* dst, const bf16* base, int64_t rs, int t0, int S) {
    constexpr int VPR = DH / 8;
    for (int e = threadIdx.x; e < KT * VPR; e += 256) {
        const int r = e / VPR, c = e % VPR;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (t0 + r < S) v = *reinterpret_cast<const uint4*>(base + (int64_t)(t0 + r) * rs + c * 8);
        *reinterpret_cast<uint4*>(dst + r * L<DH>::LD + c * 8) = v;
    }
}

// the same in two halves, so that the global loads of tile t+1 fly while tile t is being computed: tile_load (unconditional loads of
// clamped rows, zero selected afterwards) into registers, tile_store into LDS at the top of the next iteration
template <int DH> struct TileRegs { uint4 v[KT * (DH / 8) / 256]; uint32_t ok; };
template <int DH>
__device__ __forceinline__ void tile_load(TileRegs<DH>& r, const bf16* base, int64_t rs, int t0, int S) {
    constexpr int VPR = DH / 8, NV = KT * VPR / 256;
    uint32_t ok = 0u;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int e = threadIdx.x + 256 * i, row = e / VPR, c = e % VPR;
        const int rr = t0 + row;
        r.v[i] = *reinterpret_cast<const uint4*>(base + (int64_t)(rr < S ? rr : S - 1) * rs + c * 8);
        ok |= (rr < S ? 1u : 0u) << i;      // (the zeroing waits until tile_store: a select here would wait for the load at once)
    }
    r.ok = ok;
}
template <int DH>
__device__ __forceinline__ void tile_store(bf16* dst, const TileRegs<DH>& r) {
    constexpr int VPR = DH / 8, NV = KT * VPR / 256;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int e = threadIdx.x + 256 * i;
        const bool ok = (r.ok >> i) & 1u;
        const uint4 v = r.v[i];
        *reinterpret_cast<uint4*>(dst + (e / VPR) * L<DH>::LD + (e % VPR) * 8) = make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u);
    }
}

// row-operand fragment: 16 rows x 32 k (row r = lane&15, k = 8*(lane>>4)..+7) from a k-contiguous LDS tile
template <int DH>
__device__ __forceinline__ bf16x8 frag_rows(const bf16* tile, int row0, int k0, int lane) {
    return *reinterpret_cast<const bf16x8*>(tile + (row0 + (lane & 15)) * L<DH>::LD + k0 + (lane >> 4) * 8);
}
// the same fragment read straight from global memory (operand kept in registers for the whole kernel)
__device__ __forceinline__ bf16x8 frag_global(const bf16* base, int64_t rs, int row, int nrows, int k0, int lane) {
    const int r = row + (lane & 15);
    if (r >= nrows) { bf16x8 z; for (int i = 0; i < 8; ++i) z[i] = (bf16)0.f; return z; }
    return *reinterpret_cast<const bf16x8*>(base + (int64_t)r * rs + k0 + (lane >> 4) * 8);
}
// TRANSPOSED fragment of a row-major LDS tile: matrix rows = 16 columns c0..c0+15 of the tile, k-slots
// 8g+j = tile rows {r0+4g+j (j<4), r0+16+4g+(j-4) (j>=4)}  (g = lane>>4): two transpose reads
template <int DH>
__device__ __forceinline__ bf16x8 frag_cols_T(const bf16* tile, int r0, int c0, int lane) {
    const int Ls = lane & 15, g = lane >> 4;
    const bf16* q = tile + (r0 + 4 * g + (Ls >> 2)) * L<DH>::LD + c0 + (Ls & 3) * 4;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, q));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, q + 16 * L<DH>::LD));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float xor_max(float v) { return wave::max_x16_x32(v); }   // (v_permlane16/32_swap instead of ds_bpermute: wave_ops.h)
__device__ __forceinline__ float xor_sum(float v) { return wave::sum_x16_x32(v); }

// ------------------------------------------------------------------------------------------------
// forward: grid (ceil(Sq/64), H, B), 4 waves x 16 queries
// ------------------------------------------------------------------------------------------------
template <int DH>
__global__ __launch_bounds__(256) void attn_fwd_mfma(const RalfAttnDesc d) {
    __shared__ __attribute__((aligned(16))) bf16 Ks[KT * L<DH>::LD];
    __shared__ __attribute__((aligned(16))) bf16 Vs[KT * L<DH>::LD];
    __shared__ __attribute__((aligned(16))) uint8_t Ms[KT];   // per key of the staged tile: padded / beyond Sk (one byte load per key per TILE)
    __shared__ int tile_any;                                    // any such key in the tile
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
    const WgId wg = wg_id((d.Sq + 63) / 64, d.H);
    const int b = wg.b, h = wg.h;
    const int q0 = wg.tile * 64 + wave * 16, qi = q0 + (lane & 15);
    const bf16* Qp = (const bf16*)d.q + b * d.q_bs + (int64_t)h * DH;
    const bf16* Kp = (const bf16*)d.k + b * d.k_bs + (int64_t)h * DH;
    const bf16* Vp = (const bf16*)d.v + b * d.v_bs + (int64_t)h * DH;
    const uint8_t* kpm = d.kpm ? d.kpm + (int64_t)b * (d.kpm_bs ? d.kpm_bs : (int64_t)d.Sk) : nullptr;
    const uint64_t seed = d.p_drop > 0.f ? (uint64_t)d.seed[0] : 0;
    const uint32_t thr = attn_thr16(d.p_drop);
    const float inv_keep = 1.f / (1.f - d.p_drop);
    const uint32_t rowkey = d.p_drop > 0.f ? attn_rowkey(seed, d.call_id, ((uint64_t)b * d.H + h) * d.Sq + qi) : 0u;   // once per query row
    const bool active = q0 < d.Sq;  // wave-uniform

    bf16x8 qf[DH / 32];
#pragma unroll
    for (int c = 0; c < DH / 32; ++c) qf[c] = frag_global(Qp, d.q_rs, q0, d.Sq, c * 32, lane);
    f32x4 o[DH / 16];
#pragma unroll
    for (int c = 0; c < DH / 16; ++c) o[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // the per-score work bounds this kernel (VALU, not the matrix cores): attn::fwd_step (attn_core.h)
    float m = -__builtin_inff(), l = 0.f;   // running max of the RAW scores / this lane's share of the sum of 2^((score - m) * scale * log2 e)
    const float scale2 = d.scale * 1.4426950408889634f;
    const bool drop = d.p_drop > 0.f;

    TileRegs<DH> kreg, vreg;
    tile_load<DH>(kreg, Kp, d.k_rs, 0, d.Sk);
    tile_load<DH>(vreg, Vp, d.v_rs, 0, d.Sk);
    for (int t0 = 0; t0 < d.Sk; t0 += KT) {
        tile_store<DH>(Ks, kreg);
        tile_store<DH>(Vs, vreg);
        if (threadIdx.x < KT) {   // wave 0
            const bool mk = t0 + (int)threadIdx.x >= d.Sk || (kpm && kpm[t0 + threadIdx.x]);
            Ms[threadIdx.x] = mk ? 1 : 0;
            const unsigned long long any = __ballot(mk);
            if (threadIdx.x == 0) tile_any = any != 0ull ? 1 : 0;
        }
        __syncthreads();
        if (t0 + KT < d.Sk) {   // the next tile's rows start their way now and land under this tile's matrix work
            tile_load<DH>(kreg, Kp, d.k_rs, t0 + KT, d.Sk);
            tile_load<DH>(vreg, Vp, d.v_rs, t0 + KT, d.Sk);
        }
        const bool tile_masked = tile_any != 0;
        if (active && !(d.causal && t0 > q0 + 15)) {
#pragma unroll
            for (int s0 = 0; s0 < KT; s0 += 32) {
                if (t0 + s0 >= d.Sk) break;
                // S^T for 32 keys: two 16-key blocks
                f32x4 s[2];
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) {
                    s[blk] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int c = 0; c < DH / 32; ++c) s[blk] = mfma16(frag_rows<DH>(Ks, s0 + blk * 16, c * 32, lane), qf[c], s[blk]);
                }
                const bool msk = tile_masked || d.causal;   // wave-uniform: most tiles of the model (no padded key, not causal) skip the mask work
                uint32_t mw0 = 0u, mw1 = 0u;
                if (msk) { mw0 = *reinterpret_cast<const uint32_t*>(Ms + s0 + 4 * g); mw1 = *reinterpret_cast<const uint32_t*>(Ms + s0 + 16 + 4 * g); }
                const bf16x8 pf = attn::fwd_step<DH / 16>(s, m, l, o, scale2, msk, mw0, mw1, d.causal != 0, t0 + s0 + 4 * g, qi, drop, rowkey, thr);
#pragma unroll
                for (int c = 0; c < DH / 16; ++c) o[c] = mfma16(frag_cols_T<DH>(Vs, s0, c * 16, lane), pf, o[c]);
            }
        }
        __syncthreads();
    }
    l = xor_sum(l);   // the four lane groups' shares (every wave, whether it writes or not: cross-lane operations want all lanes)
    if (active && qi < d.Sq) {
        // (a query row whose keys are ALL masked has l == 0: its output is defined as 0 and its lse as -inf -- this translation unit is compiled
        //  with -fno-honor-nans, under which a 0 * inf here would be poison, not a NaN the training loop's checks could see)
        const float inv = l > 0.f ? (drop ? inv_keep : 1.f) / l : 0.f;   // (the kept probabilities went into P V unscaled)
        bf16* Op = (bf16*)d.o + b * d.o_bs + (int64_t)qi * d.o_rs + (int64_t)h * DH;
#pragma unroll
        for (int c = 0; c < DH / 16; ++c) {
            bf16x4 t;
#pragma unroll
            for (int r = 0; r < 4; ++r) t[r] = (bf16)(o[c][r] * inv);
            *reinterpret_cast<bf16x4*>(Op + c * 16 + 4 * g) = t;
        }
        if (d.lse && g == 0) d.lse[((int64_t)b * d.H + h) * d.Sq + qi] = l > 0.f ? __fmaf_rn(m, d.scale, __logf(l)) : -INFINITY;   // (explicit fma: tlayer.hip must write the same bits)
    }
}

// ------------------------------------------------------------------------------------------------
// dQ, dK, dV in ONE pass for short sequences (Sq, Sk <= 256, dh = 32): one workgroup of 8 waves per (batch, head).
// The two-kernel backward recomputes S, P, the dropout mask and dS twice (once with a lane per query for dQ, once with a lane per key for
// dK / dV), and that per-score VALU work -- not the matrix cores -- bounds both kernels.  Here wave w owns keys 32w .. 32w+31 as in the
// per-key kernel (dK, dV accumulate in registers over the query steps), and the SAME dS feeds dQ: the wave writes its dS block
// [32 keys][32 queries] to a private LDS tile, reads it back through the transpose path as the key-contracted operand
// (dQ^T[dim][query] = K^T dS^T), and the eight waves' partial dQ of a 32-query step meet in LDS slots that all threads sum IN WAVE ORDER
// (deterministic; double-buffered slots: one barrier per step).  delta = rowsum(dO o O) is computed in the prologue.
// ------------------------------------------------------------------------------------------------
constexpr int FB_S = 256, FB_LD = 40, FB_SLOT_LD = 36;
__global__ __launch_bounds__(512) void attn_bwd_fused_mfma(const RalfAttnDesc d) {
    constexpr int DH = 32;
    __shared__ __attribute__((aligned(16))) bf16 Qs[FB_S * FB_LD];
    __shared__ __attribute__((aligned(16))) bf16 Gs[FB_S * FB_LD];
    __shared__ __attribute__((aligned(16))) bf16 Kt[8][32 * FB_LD];      // per wave: its 32 key rows (transpose-read for dQ)
    __shared__ __attribute__((aligned(16))) bf16 St[8][32 * FB_LD];      // per wave: dS [key][query] of the current step
    __shared__ __attribute__((aligned(16))) float Slot[2][8][32 * FB_SLOT_LD];   // partial dQ [query][dim] per wave, double-buffered
    __shared__ __attribute__((aligned(16))) float Ls[FB_S], Ds[FB_S];
    __shared__ __attribute__((aligned(16))) uint32_t Rk[FB_S];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, Ln = lane & 15;
    const WgId wg = wg_id(1, d.H);
    const int b = wg.b, h = wg.h;
    const bf16* Qp = (const bf16*)d.q + b * d.q_bs + (int64_t)h * DH;
    const bf16* Kp = (const bf16*)d.k + b * d.k_bs + (int64_t)h * DH;
    const bf16* Vp = (const bf16*)d.v + b * d.v_bs + (int64_t)h * DH;
    const bf16* Op = (const bf16*)d.o + b * d.o_bs + (int64_t)h * DH;
    const bf16* Gp = (const bf16*)d.dout + b * d.do_bs + (int64_t)h * DH;
    const uint8_t* kpm = d.kpm ? d.kpm + (int64_t)b * (d.kpm_bs ? d.kpm_bs : (int64_t)d.Sk) : nullptr;
    const uint64_t seed = d.p_drop > 0.f ? (uint64_t)d.seed[0] : 0;
    const uint32_t thr = attn_thr16(d.p_drop);
    const float inv_keep = 1.f / (1.f - d.p_drop);
    const float scale2 = d.scale * 1.4426950408889634f;
    const int64_t stat0 = ((int64_t)b * d.H + h) * d.Sq;

    // ---- prologue: Q and dO rows into LDS, delta, lse (log2 domain), dropout row keys; the wave's key rows ----
    {   // 2 threads per query row: 16 dims each
        const int q = tid >> 1, half = tid & 1;
        uint4 qv[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)}, gv[2] = {qv[0], qv[0]}, ov[2] = {qv[0], qv[0]};
        if (q < d.Sq) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                qv[i] = *reinterpret_cast<const uint4*>(Qp + (int64_t)q * d.q_rs + half * 16 + i * 8);
                gv[i] = *reinterpret_cast<const uint4*>(Gp + (int64_t)q * d.do_rs + half * 16 + i * 8);
                ov[i] = *reinterpret_cast<const uint4*>(Op + (int64_t)q * d.o_rs + half * 16 + i * 8);
            }
        }
        float dl = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<uint4*>(Qs + q * FB_LD + half * 16 + i * 8) = qv[i];
            *reinterpret_cast<uint4*>(Gs + q * FB_LD + half * 16 + i * 8) = gv[i];
            const bf16x8 gg = *reinterpret_cast<const bf16x8*>(&gv[i]), oo = *reinterpret_cast<const bf16x8*>(&ov[i]);
#pragma unroll
            for (int e = 0; e < 8; ++e) dl += (float)gg[e] * (float)oo[e];
        }
        dl += __shfl_xor(dl, 1);
        if (half == 0) {
            Ds[q] = dl;
            Ls[q] = q < d.Sq ? d.lse[stat0 + q] * 1.4426950408889634f : 0.f;
            Rk[q] = d.p_drop > 0.f ? attn_rowkey(seed, d.call_id, ((uint64_t)b * d.H + h) * d.Sq + q) : 0u;
            if (d.delta && q < d.Sq) d.delta[stat0 + q] = dl;
        }
    }
    const int kw0 = wave * 32;   // the wave's first key
    {   // the wave's 32 key rows (4 x 16-byte vectors each): 128 vectors, 2 per lane
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = lane + 64 * i, r = e >> 2, c = e & 3;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (kw0 + r < d.Sk) v = *reinterpret_cast<const uint4*>(Kp + (int64_t)(kw0 + r) * d.k_rs + c * 8);
            *reinterpret_cast<uint4*>(Kt[wave] + r * FB_LD + c * 8) = v;
        }
    }
    bf16x8 kf[2], vf[2];
    bool kmasked[2];
#pragma unroll
    for (int hk = 0; hk < 2; ++hk) {
        kf[hk] = frag_global(Kp, d.k_rs, kw0 + hk * 16, d.Sk, 0, lane);
        vf[hk] = frag_global(Vp, d.v_rs, kw0 + hk * 16, d.Sk, 0, lane);
        const int kj = kw0 + hk * 16 + Ln;
        kmasked[hk] = kj >= d.Sk || (kpm && kpm[kj < d.Sk ? kj : 0]);
    }
    f32x4 dk[2][2], dv[2][2];
#pragma unroll
    for (int hk = 0; hk < 2; ++hk)
#pragma unroll
        for (int c = 0; c < 2; ++c) dk[hk][c] = dv[hk][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    const bool wave_active = kw0 < d.Sk;   // (wave-uniform)
    const bool drop = d.p_drop > 0.f;
    const bool any_kmasked = __builtin_amdgcn_ballot_w64(kmasked[0] || kmasked[1]) != 0ull;   // (wave-uniform)
    int step = 0;
    for (int s0 = 0; s0 < d.Sq; s0 += 32, ++step) {
        float* slot = Slot[step & 1][wave];
        if (wave_active) {
            const f32x4 L4[2] = {*reinterpret_cast<const f32x4*>(Ls + s0 + 4 * g), *reinterpret_cast<const f32x4*>(Ls + s0 + 16 + 4 * g)};
            const f32x4 D4[2] = {*reinterpret_cast<const f32x4*>(Ds + s0 + 4 * g), *reinterpret_cast<const f32x4*>(Ds + s0 + 16 + 4 * g)};
            const attn::u32x4 R4[2] = {*reinterpret_cast<const attn::u32x4*>(Rk + s0 + 4 * g), *reinterpret_cast<const attn::u32x4*>(Rk + s0 + 16 + 4 * g)};
#pragma unroll
            for (int hk = 0; hk < 2; ++hk) {
                const int kj = kw0 + hk * 16 + Ln;
                // S[query][key], dP[query][key]: rows = 16 queries (LDS), columns = 16 of the wave's keys
                f32x4 s[2], dp[2];
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) {
                    const bf16x8 qa = *reinterpret_cast<const bf16x8*>(Qs + (s0 + blk * 16 + Ln) * FB_LD + g * 8);
                    const bf16x8 ga = *reinterpret_cast<const bf16x8*>(Gs + (s0 + blk * 16 + Ln) * FB_LD + g * 8);
                    s[blk] = mfma16(qa, kf[hk], (f32x4){0.f, 0.f, 0.f, 0.f});
                    dp[blk] = mfma16(ga, vf[hk], (f32x4){0.f, 0.f, 0.f, 0.f});
                }
                bf16x8 pf, dsf;
                // (masked probabilities are SELECTED zeros: a padded key whose score exceeds the row's log-sum-exp by > 128 in the log2 domain gives
                //  exp2 = +inf, and 0 * inf = NaN would poison dQ / dK / dV of the whole (batch, head))
                if (d.causal || s0 + 32 > d.Sq || any_kmasked)
                    attn::bwd_step_keys<true>(s, dp, L4, D4, R4, scale2, kmasked[hk], d.causal != 0, kj, s0 + 4 * g, d.Sq, drop, thr, inv_keep, pf, dsf);
                else
                    attn::bwd_step_keys<false>(s, dp, L4, D4, R4, scale2, false, false, kj, s0 + 4 * g, d.Sq, drop, thr, inv_keep, pf, dsf);
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    // transposed fragments of the query tiles: matrix rows = 16 dims, k-slots = the step's 32 queries (frag_cols_T's slot order
                    // = the order pf / dsf hold them in)
                    const bf16* qg = Gs + (s0 + 4 * g + (Ln >> 2)) * FB_LD + c * 16 + (Ln & 3) * 4;
                    const bf16* qq = Qs + (s0 + 4 * g + (Ln >> 2)) * FB_LD + c * 16 + (Ln & 3) * 4;
                    const bf16x4 glo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, qg)), ghi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, qg + 16 * FB_LD));
                    const bf16x4 qlo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, qq)), qhi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, qq + 16 * FB_LD));
                    dv[hk][c] = mfma16(__builtin_shufflevector(glo, ghi, 0, 1, 2, 3, 4, 5, 6, 7), pf, dv[hk][c]);
                    dk[hk][c] = mfma16(__builtin_shufflevector(qlo, qhi, 0, 1, 2, 3, 4, 5, 6, 7), dsf, dk[hk][c]);
                }
                // dS block -> the wave's tile [key][query]: the lane's key row, queries 4g .. 4g+3 and 16+4g .. 16+4g+3
                bf16* sr = St[wave] + (hk * 16 + Ln) * FB_LD;
                *reinterpret_cast<bf16x4*>(sr + 4 * g) = __builtin_shufflevector(dsf, dsf, 0, 1, 2, 3);
                *reinterpret_cast<bf16x4*>(sr + 16 + 4 * g) = __builtin_shufflevector(dsf, dsf, 4, 5, 6, 7);
            }
            // partial dQ^T[dim][query] = K^T dS^T over the wave's 32 keys (both operands through the transpose read: same k-slot order)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave's own LDS writes above (no other wave touches St[wave])
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                const bf16* sp = St[wave] + (4 * g + (Ln >> 2)) * FB_LD + qb * 16 + (Ln & 3) * 4;
                const bf16x4 slo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, sp)), shi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, sp + 16 * FB_LD));
                const bf16x8 sf = __builtin_shufflevector(slo, shi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const bf16* kp2 = Kt[wave] + (4 * g + (Ln >> 2)) * FB_LD + db * 16 + (Ln & 3) * 4;
                    const bf16x4 klo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, kp2)), khi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, kp2 + 16 * FB_LD));
                    const f32x4 part = mfma16(__builtin_shufflevector(klo, khi, 0, 1, 2, 3, 4, 5, 6, 7), sf, (f32x4){0.f, 0.f, 0.f, 0.f});
                    // lane = query qb*16 + Ln, dims db*16 + 4g .. +3
                    *reinterpret_cast<f32x4*>(slot + (qb * 16 + Ln) * FB_SLOT_LD + db * 16 + 4 * g) = part;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {   // an idle wave's slot must still read as zero: 32 x 32 floats, 4 x float4 per lane
                const int e = lane + 64 * i, r = e >> 3, c = e & 7;
                *reinterpret_cast<f32x4*>(slot + r * FB_SLOT_LD + c * 4) = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        __syncthreads();
        {   // dQ rows of this step: the eight partials in wave order, two dims per thread
            const int q = tid >> 4, dp2 = (tid & 15) * 2;
            float a0 = 0.f, a1 = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                const float2 v = *reinterpret_cast<const float2*>(Slot[step & 1][w] + q * FB_SLOT_LD + dp2);
                a0 += v.x; a1 += v.y;
            }
            if (s0 + q < d.Sq) {
                typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                bf16x2 t;
                t[0] = (bf16)(a0 * d.scale); t[1] = (bf16)(a1 * d.scale);
                *reinterpret_cast<bf16x2*>((bf16*)d.dq + b * d.dq_bs + (int64_t)(s0 + q) * d.dq_rs + (int64_t)h * DH + dp2) = t;
            }
        }
    }
    const float vscale = drop ? inv_keep : 1.f;   // (P went into the dV product unscaled)
#pragma unroll
    for (int hk = 0; hk < 2; ++hk) {
        const int kj = kw0 + hk * 16 + Ln;
        if (kj < d.Sk) {
            bf16* dKp = (bf16*)d.dk + b * d.dk_bs + (int64_t)kj * d.dk_rs + (int64_t)h * DH;
            bf16* dVp = (bf16*)d.dv + b * d.dv_bs + (int64_t)kj * d.dv_rs + (int64_t)h * DH;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                bf16x4 tk, tv;
#pragma unroll
                for (int r = 0; r < 4; ++r) { tk[r] = (bf16)(dk[hk][c][r] * d.scale); tv[r] = (bf16)(dv[hk][c][r] * vscale); }
                *reinterpret_cast<bf16x4*>(dKp + c * 16 + 4 * g) = tk;
                *reinterpret_cast<bf16x4*>(dVp + c * 16 + 4 * g) = tv;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The same one-pass backward with SIXTEEN waves of 16 keys each (1 024 threads: four waves per SIMD, <= 128 registers).  The 8-wave form
// above holds 32 keys per wave in 147 registers and 157 KB of LDS (the double-buffered fp32 partial-dQ slots are 73 KB of it): two waves per
// SIMD, and the per-score chain of one wave (LDS -> MFMA -> ~200 VALU instructions -> MFMA -> LDS) has nobody to hide behind (rocprofv3:
// ~40 % VALU issue).  Here
//   * a wave's S / dP / dV / dK work is the same code on HALF the keys (the registers of one 16-key block); with four waves per SIMD the
//     score arithmetic runs at the VALU issue rate (cycle stamps: 3 750 cycles per 32-query step = 4 waves x ~215 instructions x 4 cycles
//     + the quarter-rate exponentials);
//   * dS of all 256 keys of a 64-query batch goes to ONE shared tile St[key][query] (bf16, double-buffered over the batches); behind the
//     batch's barrier waves 0 .. 7 -- two per SIMD -- each form one 16 x 16 tile of dQ^T = K^T dS^T over ALL keys in a single accumulator
//     chain (ascending key blocks: deterministic) and store it, while the other eight are already in the next batch's score arithmetic:
//     no fp32 partials, no reduction pass, one barrier per 64 queries.  137 KB of LDS.
// What is left outside the VALU rate (cycle stamps of one workgroup, 256 x 256, p = 0.1: 50 000 cycles per head): the prologue's loads
// (10 000 cycles: every workgroup of the chip loads its head at the same moment, HBM rate; a persistent form that prefetched the next
// head's rows -- into registers or, by one dword per line, into L2 -- moved the wait but did not shorten the kernel: 32 heads per XCD are
// 6 MB of lines against 4 MB of L2) and the dK / dV stores.
// ------------------------------------------------------------------------------------------------
constexpr int F16_LD = 40, F16_SLD = 72, F16_BATCH = 64;
__global__ __launch_bounds__(1024) void attn_bwd_fused16_mfma(const RalfAttnDesc d) {
    constexpr int DH = 32;
    __shared__ __attribute__((aligned(16))) bf16 Qs[FB_S * F16_LD];
    __shared__ __attribute__((aligned(16))) bf16 Gs[FB_S * F16_LD];
    __shared__ __attribute__((aligned(16))) bf16 Kt[FB_S * F16_LD];         // all key rows (transpose-read for dQ)
    __shared__ __attribute__((aligned(16))) bf16 St[2][FB_S * F16_SLD];     // dS [key][query of the batch], double-buffered over the batches
    __shared__ __attribute__((aligned(16))) float Ls[FB_S], Ds[FB_S];
    __shared__ __attribute__((aligned(16))) uint32_t Rk[FB_S];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, Ln = lane & 15;
    const WgId wg = wg_id(1, d.H);
    const int b = wg.b, h = wg.h;
    const bf16* Qp = (const bf16*)d.q + b * d.q_bs + (int64_t)h * DH;
    const bf16* Kp = (const bf16*)d.k + b * d.k_bs + (int64_t)h * DH;
    const bf16* Vp = (const bf16*)d.v + b * d.v_bs + (int64_t)h * DH;
    const bf16* Op = (const bf16*)d.o + b * d.o_bs + (int64_t)h * DH;
    const bf16* Gp = (const bf16*)d.dout + b * d.do_bs + (int64_t)h * DH;
    const uint8_t* kpm = d.kpm ? d.kpm + (int64_t)b * (d.kpm_bs ? d.kpm_bs : (int64_t)d.Sk) : nullptr;
    const uint64_t seed = d.p_drop > 0.f ? (uint64_t)d.seed[0] : 0;
    const uint32_t thr = attn_thr16(d.p_drop);
    const float inv_keep = 1.f / (1.f - d.p_drop);
    const float scale2 = d.scale * 1.4426950408889634f;
    const int64_t stat0 = ((int64_t)b * d.H + h) * d.Sq;

    // ---- prologue: Q and dO rows into LDS, delta, lse (log2 domain), dropout row keys; the key rows; zeroed dS rows of idle waves ----
    const int kw0 = wave * 16;   // the wave's first key
    const int kj = kw0 + Ln;
    const bf16x8 kf = frag_global(Kp, d.k_rs, kw0, d.Sk, 0, lane), vf = frag_global(Vp, d.v_rs, kw0, d.Sk, 0, lane);
    const bool kmasked = kj >= d.Sk || (kpm && kpm[kj < d.Sk ? kj : 0]);
    {   // 4 threads per query row: 8 dims each
        const int q = tid >> 2, part = tid & 3;
        uint4 qv = make_uint4(0, 0, 0, 0), gv = qv, ov = qv;
        float lse_q = 0.f;
        if (q < d.Sq) {
            qv = *reinterpret_cast<const uint4*>(Qp + (int64_t)q * d.q_rs + part * 8);
            gv = *reinterpret_cast<const uint4*>(Gp + (int64_t)q * d.do_rs + part * 8);
            ov = *reinterpret_cast<const uint4*>(Op + (int64_t)q * d.o_rs + part * 8);
            lse_q = d.lse[stat0 + q];
        }
        // key row q, vector `part` (256 rows x 4 vectors = one per thread)
        uint4 kv = make_uint4(0, 0, 0, 0);
        if (q < d.Sk) kv = *reinterpret_cast<const uint4*>(Kp + (int64_t)q * d.k_rs + part * 8);
        *reinterpret_cast<uint4*>(Qs + q * F16_LD + part * 8) = qv;
        *reinterpret_cast<uint4*>(Gs + q * F16_LD + part * 8) = gv;
        *reinterpret_cast<uint4*>(Kt + q * F16_LD + part * 8) = kv;
        const bf16x8 gg = *reinterpret_cast<const bf16x8*>(&gv), oo = *reinterpret_cast<const bf16x8*>(&ov);
        float dl = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) dl += (float)gg[e] * (float)oo[e];
        dl += wave::dpp<wave::QUAD_XOR1>(dl);
        dl += wave::dpp<wave::QUAD_XOR2>(dl);
        if (part == 0) {
            Ds[q] = dl;
            Ls[q] = lse_q * 1.4426950408889634f;
            Rk[q] = d.p_drop > 0.f ? attn_rowkey(seed, d.call_id, ((uint64_t)b * d.H + h) * d.Sq + q) : 0u;
            if (d.delta && q < d.Sq) d.delta[stat0 + q] = dl;
        }
        // a wave beyond Sk never writes its dS rows: they must read as zero in both buffers (16 of the row's 64 queries per thread)
        if (q >= (d.Sk & ~15)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(St[i >> 1] + q * F16_SLD + part * 16 + (i & 1) * 8) = make_uint4(0, 0, 0, 0);
        }
    }
    f32x4 dk[2], dv[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) dk[c] = dv[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    const bool wave_active = kw0 < d.Sk;   // (wave-uniform)
    const bool drop = d.p_drop > 0.f;
    const bool any_kmasked = __builtin_amdgcn_ballot_w64(kmasked) != 0ull;   // (wave-uniform)
    const int nkb = (d.Sk + 31) >> 5;      // 32-key blocks the dQ chain walks
    int nb = 0;
    for (int b0 = 0; b0 < d.Sq; b0 += F16_BATCH, ++nb) {
        bf16* stw = St[nb & 1];
        if (wave_active) {
            for (int s0 = b0; s0 < d.Sq && s0 < b0 + F16_BATCH; s0 += 32) {
                const f32x4 L4[2] = {*reinterpret_cast<const f32x4*>(Ls + s0 + 4 * g), *reinterpret_cast<const f32x4*>(Ls + s0 + 16 + 4 * g)};
                const f32x4 D4[2] = {*reinterpret_cast<const f32x4*>(Ds + s0 + 4 * g), *reinterpret_cast<const f32x4*>(Ds + s0 + 16 + 4 * g)};
                const attn::u32x4 R4[2] = {*reinterpret_cast<const attn::u32x4*>(Rk + s0 + 4 * g), *reinterpret_cast<const attn::u32x4*>(Rk + s0 + 16 + 4 * g)};
                f32x4 s[2], dp[2];
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) {
                    const bf16x8 qa = *reinterpret_cast<const bf16x8*>(Qs + (s0 + blk * 16 + Ln) * F16_LD + g * 8);
                    const bf16x8 ga = *reinterpret_cast<const bf16x8*>(Gs + (s0 + blk * 16 + Ln) * F16_LD + g * 8);
                    s[blk] = mfma16(qa, kf, (f32x4){0.f, 0.f, 0.f, 0.f});
                    dp[blk] = mfma16(ga, vf, (f32x4){0.f, 0.f, 0.f, 0.f});
                }
                bf16x8 pf, dsf;
                // (masked probabilities are SELECTED zeros, see attn_bwd_fused_mfma)
                if (d.causal || s0 + 32 > d.Sq || any_kmasked)
                    attn::bwd_step_keys<true>(s, dp, L4, D4, R4, scale2, kmasked, d.causal != 0, kj, s0 + 4 * g, d.Sq, drop, thr, inv_keep, pf, dsf);
                else
                    attn::bwd_step_keys<false>(s, dp, L4, D4, R4, scale2, false, false, kj, s0 + 4 * g, d.Sq, drop, thr, inv_keep, pf, dsf);
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    // transposed fragments of the query tiles: matrix rows = 16 dims, k-slots = the step's 32 queries (the order pf / dsf hold them in)
                    const bf16* qg = Gs + (s0 + 4 * g + (Ln >> 2)) * F16_LD + c * 16 + (Ln & 3) * 4;
                    const bf16* qq = Qs + (s0 + 4 * g + (Ln >> 2)) * F16_LD + c * 16 + (Ln & 3) * 4;
                    const bf16x4 glo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, qg)), ghi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, qg + 16 * F16_LD));
                    const bf16x4 qlo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, qq)), qhi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, qq + 16 * F16_LD));
                    dv[c] = mfma16(__builtin_shufflevector(glo, ghi, 0, 1, 2, 3, 4, 5, 6, 7), pf, dv[c]);
                    dk[c] = mfma16(__builtin_shufflevector(qlo, qhi, 0, 1, 2, 3, 4, 5, 6, 7), dsf, dk[c]);
                }
                bf16* sr = stw + kj * F16_SLD + (s0 - b0);   // the lane's key row: queries 4g .. 4g+3 and 16+4g .. 16+4g+3 of the step
                *reinterpret_cast<bf16x4*>(sr + 4 * g) = __builtin_shufflevector(dsf, dsf, 0, 1, 2, 3);
                *reinterpret_cast<bf16x4*>(sr + 16 + 4 * g) = __builtin_shufflevector(dsf, dsf, 4, 5, 6, 7);
            }
        }
        __syncthreads();
        // (the NEXT batch writes the other buffer; this one is rewritten two batches on, behind the next barrier, which the dQ waves reach only
        //  after their reads below)
        const int qb = wave >> 1, db = wave & 1;   // dQ^T tile of wave w < 8: dims db*16 .. +15 x queries b0 + qb*16 .. +15, over all keys
        if (wave < 8 && b0 + qb * 16 < d.Sq) {     // (wave-uniform; the columns of a skipped step hold an earlier batch's values)
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const bf16* sp = stw + (4 * g + (Ln >> 2)) * F16_SLD + qb * 16 + (Ln & 3) * 4;
            const bf16* kp2 = Kt + (4 * g + (Ln >> 2)) * F16_LD + db * 16 + (Ln & 3) * 4;
            // four key blocks' operands in flight at a time; blocks beyond Sk are zero rows of both tiles
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                if (half * 4 < nkb) {
                    bf16x4 sl[4], sh[4], kl[4], kh[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int kb = half * 4 + i;
                        sl[i] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, sp + kb * 32 * F16_SLD));
                        sh[i] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, sp + (kb * 32 + 16) * F16_SLD));
                        kl[i] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, kp2 + kb * 32 * F16_LD));
                        kh[i] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, kp2 + (kb * 32 + 16) * F16_LD));
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        acc = mfma16(__builtin_shufflevector(kl[i], kh[i], 0, 1, 2, 3, 4, 5, 6, 7), __builtin_shufflevector(sl[i], sh[i], 0, 1, 2, 3, 4, 5, 6, 7), acc);
                }
            }
            const int q = b0 + qb * 16 + Ln;   // lane = query, dims db*16 + 4g .. +3
            if (q < d.Sq) {
                bf16x4 t;
#pragma unroll
                for (int r = 0; r < 4; ++r) t[r] = (bf16)(acc[r] * d.scale);
                *reinterpret_cast<bf16x4*>((bf16*)d.dq + b * d.dq_bs + (int64_t)q * d.dq_rs + (int64_t)h * DH + db * 16 + 4 * g) = t;
            }
        }
    }
    const float vscale = drop ? inv_keep : 1.f;   // (P went into the dV product unscaled)
    if (kj < d.Sk) {
        bf16* dKp = (bf16*)d.dk + b * d.dk_bs + (int64_t)kj * d.dk_rs + (int64_t)h * DH;
        bf16* dVp = (bf16*)d.dv + b * d.dv_bs + (int64_t)kj * d.dv_rs + (int64_t)h * DH;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            bf16x4 tk, tv;
#pragma unroll
            for (int r = 0; r < 4; ++r) { tk[r] = (bf16)(dk[c][r] * d.scale); tv[r] = (bf16)(dv[c][r] * vscale); }
            *reinterpret_cast<bf16x4*>(dKp + c * 16 + 4 * g) = tk;
            *reinterpret_cast<bf16x4*>(dVp + c * 16 + 4 * g) = tv;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// dQ, dK, dV in ONE pass for the decoder's CROSS-attention (a few dozen queries over hundreds of memory rows; dh = 32, not causal): one
// workgroup of 8 waves per (batch, head), two of them per CU.  The per-query + per-key pair spends 24 + 33 us per layer on it (Sq = 51,
// Sk = 532, B = 64): the per-key kernel is 4 608 workgroups that each stage the 51 query rows again for two steps of matrix work, the
// per-query kernel walks nine key tiles behind two barriers each.  Here the <= 64 query rows (Q, dO, delta, lse, dropout row keys) are staged
// ONCE; wave w owns the 32-key blocks w, w + 8, w + 16, ..: per block it forms S, P, dP, dS exactly as attn_bwd_fused_mfma does (lane = one
// key), accumulates dK / dV of the block in registers over the two query steps and stores them, and adds the block's contribution to ITS
// partial dQ (registers, all query steps) through the transpose-read of its private dS / K tiles -- no barrier inside the key loop.  The
// eight partial dQ meet once at the end in LDS (the K / dS tiles' space) and are summed in wave order (deterministic).
// dK / dV carry the bits of attn_bwd_dkv_mfma (same arithmetic, same query order); dQ sums the keys in another association than
// attn_bwd_dq_mfma (per-wave partials) -- both are fp32 sums of the same bf16-rounded dS K products.
// ------------------------------------------------------------------------------------------------
constexpr int XB_Q = 64, XB_LD = 40, XB_SLD = XB_Q + 8, XB_SLOT_LD = 36;
constexpr int XB_TILE = 32 * XB_LD + 32 * XB_SLD;   // per wave (elements): 32 key rows [32][40] | dS [32 keys][64 queries + pad]
template <int WPS>   // waves per SIMD the register allocation aims at: 4 = two workgroups per CU (128 registers), 2 = one
__global__ __launch_bounds__(512, WPS) void attn_bwd_cross_mfma(const RalfAttnDesc d) {
    constexpr int DH = 32;
    __shared__ __attribute__((aligned(16))) bf16 Qs[XB_Q * XB_LD];
    __shared__ __attribute__((aligned(16))) bf16 Gs[XB_Q * XB_LD];
    __shared__ __attribute__((aligned(16))) bf16 Tile[8 * XB_TILE];   // at the end: the partial dQ [32 queries][dim] per wave (fp32)
    static_assert(sizeof(Tile) >= 8 * 32 * XB_SLOT_LD * 4, "the dQ slots reuse the tiles' space");
    __shared__ __attribute__((aligned(16))) float Ls[XB_Q], Ds[XB_Q];
    __shared__ __attribute__((aligned(16))) uint32_t Rk[XB_Q];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, Ln = lane & 15;
    bf16* Kt = Tile + wave * XB_TILE;
    bf16* St = Kt + 32 * XB_LD;
    const WgId wg = wg_id(1, d.H);
    const int b = wg.b, h = wg.h;
    const bf16* Qp = (const bf16*)d.q + b * d.q_bs + (int64_t)h * DH;
    const bf16* Kp = (const bf16*)d.k + b * d.k_bs + (int64_t)h * DH;
    const bf16* Vp = (const bf16*)d.v + b * d.v_bs + (int64_t)h * DH;
    const bf16* Op = (const bf16*)d.o + b * d.o_bs + (int64_t)h * DH;
    const bf16* Gp = (const bf16*)d.dout + b * d.do_bs + (int64_t)h * DH;
    const uint8_t* kpm = d.kpm ? d.kpm + (int64_t)b * (d.kpm_bs ? d.kpm_bs : (int64_t)d.Sk) : nullptr;
    const uint64_t seed = d.p_drop > 0.f ? (uint64_t)d.seed[0] : 0;
    const uint32_t thr = attn_thr16(d.p_drop);
    const float inv_keep = 1.f / (1.f - d.p_drop);
    const float scale2 = d.scale * 1.4426950408889634f;
    const int64_t stat0 = ((int64_t)b * d.H + h) * d.Sq;

    // the fragments of the wave's first 16 keys: requested before the query rows are staged (one round trip for both)
    bf16x8 kf_n = frag_global(Kp, d.k_rs, wave * 32, d.Sk, 0, lane), vf_n = frag_global(Vp, d.v_rs, wave * 32, d.Sk, 0, lane);

    // ---- prologue: Q and dO rows into LDS, delta, lse (log2 domain), dropout row keys: 2 threads per query row, 16 dims each ----
    if (tid < 2 * XB_Q) {
        const int q = tid >> 1, half = tid & 1;
        uint4 qv[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)}, gv[2] = {qv[0], qv[0]}, ov[2] = {qv[0], qv[0]};
        if (q < d.Sq) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                qv[i] = *reinterpret_cast<const uint4*>(Qp + (int64_t)q * d.q_rs + half * 16 + i * 8);
                gv[i] = *reinterpret_cast<const uint4*>(Gp + (int64_t)q * d.do_rs + half * 16 + i * 8);
                ov[i] = *reinterpret_cast<const uint4*>(Op + (int64_t)q * d.o_rs + half * 16 + i * 8);
            }
        }
        float dl = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<uint4*>(Qs + q * XB_LD + half * 16 + i * 8) = qv[i];
            *reinterpret_cast<uint4*>(Gs + q * XB_LD + half * 16 + i * 8) = gv[i];
            const bf16x8 gg = *reinterpret_cast<const bf16x8*>(&gv[i]), oo = *reinterpret_cast<const bf16x8*>(&ov[i]);
#pragma unroll
            for (int e = 0; e < 8; ++e) dl += (float)gg[e] * (float)oo[e];
        }
        dl += wave::dpp<wave::QUAD_XOR1>(dl);
        if (half == 0) {
            Ds[q] = dl;
            Ls[q] = q < d.Sq ? d.lse[stat0 + q] * 1.4426950408889634f : 0.f;
            Rk[q] = d.p_drop > 0.f ? attn_rowkey(seed, d.call_id, ((uint64_t)b * d.H + h) * d.Sq + q) : 0u;
            if (d.delta && q < d.Sq) d.delta[stat0 + q] = dl;
        }
    }
    f32x4 dqa[2][2][2];   // [query step][16-query block][16-dim block]: this wave's partial dQ^T
#pragma unroll
    for (int i = 0; i < 8; ++i) dqa[i >> 2][(i >> 1) & 1][i & 1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const int nst = d.Sq > 32 ? 2 : 1;
    const bool drop = d.p_drop > 0.f;
    const float vscale = drop ? inv_keep : 1.f;   // (P goes into the dV product unscaled)

    for (int kw0 = wave * 32; kw0 < d.Sk; kw0 += 8 * 32) {
        // the block's key rows (for the transposed operand of dQ): requested now, stored into the wave's tile after the two key halves
        uint4 krows[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = lane + 64 * i, r = e >> 2, c = e & 3;
            krows[i] = *reinterpret_cast<const uint4*>(Kp + (int64_t)min(kw0 + r, d.Sk - 1) * d.k_rs + c * 8);   // (rows beyond Sk: their dS is zero)
        }
#pragma unroll
        for (int hk = 0; hk < 2; ++hk) {
            const int kj = kw0 + hk * 16 + Ln;
            const bf16x8 kf = kf_n, vf = vf_n;
            {   // the next 16 keys' fragments fly under this half's work: the block's second half, or the first half of the wave's next block
                const int nk = hk == 0 ? kw0 + 16 : kw0 + 8 * 32;
                kf_n = frag_global(Kp, d.k_rs, nk, d.Sk, 0, lane);
                vf_n = frag_global(Vp, d.v_rs, nk, d.Sk, 0, lane);
            }
            const bool kmasked = kj >= d.Sk || (kpm && kpm[kj < d.Sk ? kj : 0]);
            f32x4 dk[2], dv[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) dk[c] = dv[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
            bf16* sr = St + (hk * 16 + Ln) * XB_SLD;   // the lane's key row of the dS tile
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const int s0 = st * 32;
                if (st >= nst) break;
                const f32x4 L4[2] = {*reinterpret_cast<const f32x4*>(Ls + s0 + 4 * g), *reinterpret_cast<const f32x4*>(Ls + s0 + 16 + 4 * g)};
                const f32x4 D4[2] = {*reinterpret_cast<const f32x4*>(Ds + s0 + 4 * g), *reinterpret_cast<const f32x4*>(Ds + s0 + 16 + 4 * g)};
                // S[query][key], dP[query][key]: rows = 16 queries (LDS), columns = the 16 keys
                f32x4 s[2], dp[2];
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) {
                    const bf16x8 qa = *reinterpret_cast<const bf16x8*>(Qs + (s0 + blk * 16 + Ln) * XB_LD + g * 8);
                    const bf16x8 ga = *reinterpret_cast<const bf16x8*>(Gs + (s0 + blk * 16 + Ln) * XB_LD + g * 8);
                    s[blk] = mfma16(qa, kf, (f32x4){0.f, 0.f, 0.f, 0.f});
                    dp[blk] = mfma16(ga, vf, (f32x4){0.f, 0.f, 0.f, 0.f});
                }
                const attn::u32x4 R4[2] = {*reinterpret_cast<const attn::u32x4*>(Rk + s0 + 4 * g), *reinterpret_cast<const attn::u32x4*>(Rk + s0 + 16 + 4 * g)};
                bf16x8 pf, dsf;
                if (s0 + 32 > d.Sq || __builtin_amdgcn_ballot_w64(kmasked) != 0ull)   // (wave-uniform; masked probabilities are selected zeros)
                    attn::bwd_step_keys<true>(s, dp, L4, D4, R4, scale2, kmasked, false, kj, s0 + 4 * g, d.Sq, drop, thr, inv_keep, pf, dsf);
                else
                    attn::bwd_step_keys<false>(s, dp, L4, D4, R4, scale2, false, false, kj, s0 + 4 * g, d.Sq, drop, thr, inv_keep, pf, dsf);
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    // transposed fragments of the query tiles: matrix rows = 16 dims, k-slots = the step's 32 queries (the order pf / dsf hold them in)
                    const bf16* qg = Gs + (s0 + 4 * g + (Ln >> 2)) * XB_LD + c * 16 + (Ln & 3) * 4;
                    const bf16* qq = Qs + (s0 + 4 * g + (Ln >> 2)) * XB_LD + c * 16 + (Ln & 3) * 4;
                    const bf16x4 glo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, qg)), ghi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, qg + 16 * XB_LD));
                    const bf16x4 qlo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, qq)), qhi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, qq + 16 * XB_LD));
                    dv[c] = mfma16(__builtin_shufflevector(glo, ghi, 0, 1, 2, 3, 4, 5, 6, 7), pf, dv[c]);
                    dk[c] = mfma16(__builtin_shufflevector(qlo, qhi, 0, 1, 2, 3, 4, 5, 6, 7), dsf, dk[c]);
                }
                // dS -> the wave's tile [key][query]: the lane's key row, queries s0 + 4g .. +3 and s0 + 16 + 4g .. +3
                *reinterpret_cast<bf16x4*>(sr + s0 + 4 * g) = __builtin_shufflevector(dsf, dsf, 0, 1, 2, 3);
                *reinterpret_cast<bf16x4*>(sr + s0 + 16 + 4 * g) = __builtin_shufflevector(dsf, dsf, 4, 5, 6, 7);
            }
            if (kj < d.Sk) {
                bf16* dKp = (bf16*)d.dk + b * d.dk_bs + (int64_t)kj * d.dk_rs + (int64_t)h * DH;
                bf16* dVp = (bf16*)d.dv + b * d.dv_bs + (int64_t)kj * d.dv_rs + (int64_t)h * DH;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    bf16x4 tk, tv;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { tk[r] = (bf16)(dk[c][r] * d.scale); tv[r] = (bf16)(dv[c][r] * vscale); }
                    *reinterpret_cast<bf16x4*>(dKp + c * 16 + 4 * g) = tk;
                    *reinterpret_cast<bf16x4*>(dVp + c * 16 + 4 * g) = tv;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = lane + 64 * i;
            *reinterpret_cast<uint4*>(Kt + (e >> 2) * XB_LD + (e & 3) * 8) = krows[i];
        }
        // partial dQ^T[dim][query] += K^T dS^T over the block's 32 keys (both operands through the transpose read: same k-slot order)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave's own LDS writes above (no other wave touches its tiles)
        bf16x8 ktf[2];
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            const bf16* kp2 = Kt + (4 * g + (Ln >> 2)) * XB_LD + db * 16 + (Ln & 3) * 4;
            const bf16x4 klo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, kp2)), khi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, kp2 + 16 * XB_LD));
            ktf[db] = __builtin_shufflevector(klo, khi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            if (st >= nst) break;
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                const bf16* sp = St + (4 * g + (Ln >> 2)) * XB_SLD + st * 32 + qb * 16 + (Ln & 3) * 4;
                const bf16x4 slo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, sp)), shi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, sp + 16 * XB_SLD));
                const bf16x8 sf = __builtin_shufflevector(slo, shi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                for (int db = 0; db < 2; ++db) dqa[st][qb][db] = mfma16(ktf[db], sf, dqa[st][qb][db]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the next block rewrites the tiles)
    }
    // ---- the eight partial dQ: 32 queries per round through the tiles' space, summed in wave order ----
    float* Slot = reinterpret_cast<float*>(Tile);
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        if (st >= nst) break;
        __syncthreads();   // every wave is done with its tiles (round 0) / with reading the previous round's slots
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int db = 0; db < 2; ++db)   // lane = query qb*16 + Ln, dims db*16 + 4g .. +3
                *reinterpret_cast<f32x4*>(Slot + wave * 32 * XB_SLOT_LD + (qb * 16 + Ln) * XB_SLOT_LD + db * 16 + 4 * g) = dqa[st][qb][db];
        __syncthreads();
        const int q = tid >> 4, dp2 = (tid & 15) * 2;
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            const float2 v = *reinterpret_cast<const float2*>(Slot + w * 32 * XB_SLOT_LD + q * XB_SLOT_LD + dp2);
            a0 += v.x; a1 += v.y;
        }
        if (st * 32 + q < d.Sq) {
            typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
            bf16x2 t;
            t[0] = (bf16)(a0 * d.scale); t[1] = (bf16)(a1 * d.scale);
            *reinterpret_cast<bf16x2*>((bf16*)d.dq + b * d.dq_bs + (int64_t)(st * 32 + q) * d.dq_rs + (int64_t)h * DH + dp2) = t;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// dQ (+ delta): same geometry as the forward kernel
// ------------------------------------------------------------------------------------------------
template <int DH>
__global__ __launch_bounds__(256) void attn_bwd_dq_mfma(const RalfAttnDesc d) {
    __shared__ __attribute__((aligned(16))) bf16 Ks[KT * L<DH>::LD];
    __shared__ __attribute__((aligned(16))) bf16 Vs[KT * L<DH>::LD];
    __shared__ __attribute__((aligned(16))) uint8_t Ms[KT];
    __shared__ int tile_any;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
    const WgId wg = wg_id((d.Sq + 63) / 64, d.H);
    const int b = wg.b, h = wg.h;
    const int q0 = wg.tile * 64 + wave * 16, qi = q0 + (lane & 15);
    const bool qok = qi < d.Sq;
    const bf16* Qp = (const bf16*)d.q + b * d.q_bs + (int64_t)h * DH;
    const bf16* Kp = (const bf16*)d.k + b * d.k_bs + (int64_t)h * DH;
    const bf16* Vp = (const bf16*)d.v + b * d.v_bs + (int64_t)h * DH;
    const bf16* Op = (const bf16*)d.o + b * d.o_bs + (int64_t)h * DH;
    const bf16* Gp = (const bf16*)d.dout + b * d.do_bs + (int64_t)h * DH;
    const uint8_t* kpm = d.kpm ? d.kpm + (int64_t)b * (d.kpm_bs ? d.kpm_bs : (int64_t)d.Sk) : nullptr;
    const uint64_t seed = d.p_drop > 0.f ? (uint64_t)d.seed[0] : 0;
    const uint32_t thr = attn_thr16(d.p_drop);
    const float inv_keep = 1.f / (1.f - d.p_drop);
    const uint32_t rowkey = d.p_drop > 0.f ? attn_rowkey(seed, d.call_id, ((uint64_t)b * d.H + h) * d.Sq + qi) : 0u;   // once per query row
    const bool active = q0 < d.Sq;
    const int64_t stat = ((int64_t)b * d.H + h) * d.Sq + qi;

    bf16x8 qf[DH / 32], gf[DH / 32];
    float delta = 0.f;
#pragma unroll
    for (int c = 0; c < DH / 32; ++c) {
        qf[c] = frag_global(Qp, d.q_rs, q0, d.Sq, c * 32, lane);
        gf[c] = frag_global(Gp, d.do_rs, q0, d.Sq, c * 32, lane);
        const bf16x8 of = frag_global(Op, d.o_rs, q0, d.Sq, c * 32, lane);
#pragma unroll
        for (int i = 0; i < 8; ++i) delta += (float)gf[c][i] * (float)of[i];
    }
    delta = xor_sum(delta);   // the 4 lane groups hold disjoint d slices of the same query
    const float lse2 = qok ? d.lse[stat] * 1.4426950408889634f : 0.f, scale2 = d.scale * 1.4426950408889634f;
    if (qok && g == 0) d.delta[stat] = delta;
    f32x4 dq[DH / 16];
#pragma unroll
    for (int c = 0; c < DH / 16; ++c) dq[c] = (f32x4){0.f, 0.f, 0.f, 0.f};

    TileRegs<DH> kreg, vreg;
    tile_load<DH>(kreg, Kp, d.k_rs, 0, d.Sk);
    tile_load<DH>(vreg, Vp, d.v_rs, 0, d.Sk);
    for (int t0 = 0; t0 < d.Sk; t0 += KT) {
        tile_store<DH>(Ks, kreg);
        tile_store<DH>(Vs, vreg);
        if (threadIdx.x < KT) {   // wave 0
            const bool mk = t0 + (int)threadIdx.x >= d.Sk || (kpm && kpm[t0 + threadIdx.x]);
            Ms[threadIdx.x] = mk ? 1 : 0;
            const unsigned long long any = __ballot(mk);
            if (threadIdx.x == 0) tile_any = any != 0ull ? 1 : 0;
        }
        __syncthreads();
        if (t0 + KT < d.Sk) {   // the next tile's rows start their way now and land under this tile's matrix work
            tile_load<DH>(kreg, Kp, d.k_rs, t0 + KT, d.Sk);
            tile_load<DH>(vreg, Vp, d.v_rs, t0 + KT, d.Sk);
        }
        const bool tile_masked = tile_any != 0;
        if (active && !(d.causal && t0 > q0 + 15)) {
#pragma unroll
            for (int s0 = 0; s0 < KT; s0 += 32) {
                if (t0 + s0 >= d.Sk) break;
                f32x4 s[2], dp[2];
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) {
                    s[blk] = dp[blk] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int c = 0; c < DH / 32; ++c) {
                        s[blk] = mfma16(frag_rows<DH>(Ks, s0 + blk * 16, c * 32, lane), qf[c], s[blk]);
                        dp[blk] = mfma16(frag_rows<DH>(Vs, s0 + blk * 16, c * 32, lane), gf[c], dp[blk]);
                    }
                }
                bf16x8 dsf;
                if (tile_masked || d.causal || !qok) {   // (wave-uniform ballot not needed: !qok only makes the slow path run for a whole wave's step)
                    const uint32_t mw0 = *reinterpret_cast<const uint32_t*>(Ms + s0 + 4 * g), mw1 = *reinterpret_cast<const uint32_t*>(Ms + s0 + 16 + 4 * g);
                    dsf = attn::bwd_step_queries<true>(s, dp, lse2, delta, scale2, mw0, mw1, d.causal != 0, t0 + s0 + 4 * g, qi, qok, d.p_drop > 0.f, rowkey, thr, inv_keep);
                } else {
                    dsf = attn::bwd_step_queries<false>(s, dp, lse2, delta, scale2, 0u, 0u, false, t0 + s0 + 4 * g, qi, true, d.p_drop > 0.f, rowkey, thr, inv_keep);
                }
#pragma unroll
                for (int c = 0; c < DH / 16; ++c) dq[c] = mfma16(frag_cols_T<DH>(Ks, s0, c * 16, lane), dsf, dq[c]);
            }
        }
        __syncthreads();
    }
    if (qok) {
        bf16* dQp = (bf16*)d.dq + b * d.dq_bs + (int64_t)qi * d.dq_rs + (int64_t)h * DH;
#pragma unroll
        for (int c = 0; c < DH / 16; ++c) {
            bf16x4 t;
#pragma unroll
            for (int r = 0; r < 4; ++r) t[r] = (bf16)(dq[c][r] * d.scale);
            *reinterpret_cast<bf16x4*>(dQp + c * 16 + 4 * g) = t;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// dK, dV: grid (ceil(Sk/64), H, B), 4 waves x 16 keys; queries streamed in tiles of 64
// ------------------------------------------------------------------------------------------------
template <int DH>
__global__ __launch_bounds__(256) void attn_bwd_dkv_mfma(const RalfAttnDesc d) {
    __shared__ __attribute__((aligned(16))) bf16 Qs[KT * L<DH>::LD];
    __shared__ __attribute__((aligned(16))) bf16 Gs[KT * L<DH>::LD];
    __shared__ __attribute__((aligned(16))) float Ls[KT], Ds[KT];
    __shared__ __attribute__((aligned(16))) uint32_t Rk[KT];   // per-query dropout row keys of the staged query tile
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
    const WgId wg = wg_id((d.Sk + 63) / 64, d.H);
    const int b = wg.b, h = wg.h;
    const int k0 = wg.tile * 64 + wave * 16, kj = k0 + (lane & 15);
    const bool kok = kj < d.Sk;
    const bf16* Qp = (const bf16*)d.q + b * d.q_bs + (int64_t)h * DH;
    const bf16* Kp = (const bf16*)d.k + b * d.k_bs + (int64_t)h * DH;
    const bf16* Vp = (const bf16*)d.v + b * d.v_bs + (int64_t)h * DH;
    const bf16* Gp = (const bf16*)d.dout + b * d.do_bs + (int64_t)h * DH;
    const bool kmasked = !kok || (d.kpm && d.kpm[(int64_t)b * (d.kpm_bs ? d.kpm_bs : (int64_t)d.Sk) + kj]);
    const uint64_t seed = d.p_drop > 0.f ? (uint64_t)d.seed[0] : 0;
    const uint32_t thr = attn_thr16(d.p_drop);
    const float inv_keep = 1.f / (1.f - d.p_drop);
    const bool active = k0 < d.Sk;
    const float scale2 = d.scale * 1.4426950408889634f;
    const int64_t stat0 = ((int64_t)b * d.H + h) * d.Sq;

    bf16x8 kf[DH / 32], vf[DH / 32];
#pragma unroll
    for (int c = 0; c < DH / 32; ++c) {
        kf[c] = frag_global(Kp, d.k_rs, k0, d.Sk, c * 32, lane);
        vf[c] = frag_global(Vp, d.v_rs, k0, d.Sk, c * 32, lane);
    }
    f32x4 dk[DH / 16], dv[DH / 16];
#pragma unroll
    for (int c = 0; c < DH / 16; ++c) dk[c] = dv[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool drop = d.p_drop > 0.f;
    const bool any_kmasked = __builtin_amdgcn_ballot_w64(kmasked) != 0ull;   // (wave-uniform)
    const float vscale = drop ? inv_keep : 1.f;                               // (P goes into the dV product unscaled)

    TileRegs<DH> qreg, greg;
    tile_load<DH>(qreg, Qp, d.q_rs, 0, d.Sq);
    tile_load<DH>(greg, Gp, d.do_rs, 0, d.Sq);
    for (int t0 = 0; t0 < d.Sq; t0 += KT) {
        tile_store<DH>(Qs, qreg);
        tile_store<DH>(Gs, greg);
        if (threadIdx.x < KT) {
            const int qi = t0 + threadIdx.x;
            Ls[threadIdx.x] = qi < d.Sq ? d.lse[stat0 + qi] * 1.4426950408889634f : 0.f;   // log2 domain
            Ds[threadIdx.x] = qi < d.Sq ? d.delta[stat0 + qi] : 0.f;
            if (d.p_drop > 0.f) Rk[threadIdx.x] = attn_rowkey(seed, d.call_id, ((uint64_t)b * d.H + h) * d.Sq + qi);
        }
        __syncthreads();
        if (t0 + KT < d.Sq) {   // next query tile: loads fly under this tile's matrix work
            tile_load<DH>(qreg, Qp, d.q_rs, t0 + KT, d.Sq);
            tile_load<DH>(greg, Gp, d.do_rs, t0 + KT, d.Sq);
        }
        // causal: a query tile entirely before this wave's first key sees none of its keys
        if (active && !(d.causal && t0 + KT - 1 < k0)) {
#pragma unroll
            for (int s0 = 0; s0 < KT; s0 += 32) {
                if (t0 + s0 >= d.Sq) break;
                // S[query][key] and dP[query][key]: rows = 16 queries (LDS), columns = this wave's 16 keys
                f32x4 s[2], dp[2];
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) {
                    s[blk] = dp[blk] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int c = 0; c < DH / 32; ++c) {
                        s[blk] = mfma16(frag_rows<DH>(Qs, s0 + blk * 16, c * 32, lane), kf[c], s[blk]);
                        dp[blk] = mfma16(frag_rows<DH>(Gs, s0 + blk * 16, c * 32, lane), vf[c], dp[blk]);
                    }
                }
                bf16x8 pf, dsf;
                // the lane's 8 queries are two runs of 4: their statistics / row keys come as 16-byte LDS reads
                const f32x4 L4[2] = {*reinterpret_cast<const f32x4*>(Ls + s0 + 4 * g), *reinterpret_cast<const f32x4*>(Ls + s0 + 16 + 4 * g)};
                const f32x4 D4[2] = {*reinterpret_cast<const f32x4*>(Ds + s0 + 4 * g), *reinterpret_cast<const f32x4*>(Ds + s0 + 16 + 4 * g)};
                attn::u32x4 R4[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
                if (drop) { R4[0] = *reinterpret_cast<const attn::u32x4*>(Rk + s0 + 4 * g); R4[1] = *reinterpret_cast<const attn::u32x4*>(Rk + s0 + 16 + 4 * g); }
                if (d.causal || t0 + KT > d.Sq || any_kmasked)   // (wave-uniform; masked probabilities are selected zeros)
                    attn::bwd_step_keys<true>(s, dp, L4, D4, R4, scale2, kmasked, d.causal != 0, kj, t0 + s0 + 4 * g, d.Sq, drop, thr, inv_keep, pf, dsf);
                else
                    attn::bwd_step_keys<false>(s, dp, L4, D4, R4, scale2, false, false, kj, t0 + s0 + 4 * g, d.Sq, drop, thr, inv_keep, pf, dsf);
#pragma unroll
                for (int c = 0; c < DH / 16; ++c) {
                    dv[c] = mfma16(frag_cols_T<DH>(Gs, s0, c * 16, lane), pf, dv[c]);
                    dk[c] = mfma16(frag_cols_T<DH>(Qs, s0, c * 16, lane), dsf, dk[c]);
                }
            }
        }
        __syncthreads();
    }
    if (kok) {
        bf16* dKp = (bf16*)d.dk + b * d.dk_bs + (int64_t)kj * d.dk_rs + (int64_t)h * DH;
        bf16* dVp = (bf16*)d.dv + b * d.dv_bs + (int64_t)kj * d.dv_rs + (int64_t)h * DH;
#pragma unroll
        for (int c = 0; c < DH / 16; ++c) {
            bf16x4 tk, tv;
#pragma unroll
            for (int r = 0; r < 4; ++r) { tk[r] = (bf16)(dk[c][r] * d.scale); tv[r] = (bf16)(dv[c][r] * vscale); }
            *reinterpret_cast<bf16x4*>(dKp + c * 16 + 4 * g) = tk;
            *reinterpret_cast<bf16x4*>(dVp + c * 16 + 4 * g) = tv;
        }
    }
}
// ------------------------------------------------------------------------------------------------
// Decode step (Sq = 1, head dim 32, no dropout): a streaming kernel instead of the tiled one.  The work is reading the
// K/V cache once (B x Sk x 2 x d bytes: 140 MB per cross-attention call at B = 256, Sk = 532) -- there is one query per
// head, so nothing for the matrix cores to do.  Workgroup = (batch element, head PAIR): per key the pair's K (or V) slice is
// one 128-byte line; lane = (key slot = lane / 8, 16-byte chunk = lane % 8), 4 waves x 8 keys per wave-load.
// Pass 1 streams K -> scores in LDS, block softmax, pass 2 streams V weighted by the probabilities.
// ------------------------------------------------------------------------------------------------
constexpr int DEC_MAXK = 2048;
__global__ __launch_bounds__(256) void attn_decode_kernel(const RalfAttnDesc d) {
    __shared__ float sc[DEC_MAXK][2];
    __shared__ float red[2][4][2];
    __shared__ float part[4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hpairs = d.H / 2, hp = blockIdx.x % hpairs, b = blockIdx.x / hpairs;
    const int chunk = lane & 7, slot = lane >> 3, head = chunk >> 2;
    const bf16* Qp = (const bf16*)d.q + b * d.q_bs + hp * 64 + chunk * 8;
    const bf16* Kp = (const bf16*)d.k + b * d.k_bs + hp * 64 + chunk * 8;
    const bf16* Vp = (const bf16*)d.v + b * d.v_bs + hp * 64 + chunk * 8;
    const uint8_t* kpm = d.kpm ? d.kpm + (int64_t)b * (d.kpm_bs ? d.kpm_bs : (int64_t)d.Sk) : nullptr;
    float qv[8];
    {
        const bf16x8 t = *reinterpret_cast<const bf16x8*>(Qp);
#pragma unroll
        for (int i = 0; i < 8; ++i) qv[i] = (float)t[i] * d.scale;
    }
    // ---- pass 1: scores ----
    for (int key0 = wave * 8; key0 < d.Sk; key0 += 32 * 4) {
        bf16x8 kv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            // (unconditional load of a clamped row: a load under `if (key < Sk)` is followed by s_waitcnt vmcnt(0), one round trip each)
            const int key = min(key0 + u * 32 + slot, d.Sk - 1);
            kv[u] = *reinterpret_cast<const bf16x8*>(Kp + (int64_t)key * d.k_rs);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int key = key0 + u * 32 + slot;
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) s += qv[i] * (float)kv[u][i];
            s += __shfl_xor(s, 1);
            s += __shfl_xor(s, 2);
            if ((chunk & 3) == 0 && key < d.Sk) sc[key][head] = (kpm && kpm[key]) ? -__builtin_inff() : s;
        }
    }
    __syncthreads();
    // ---- softmax over the keys, per head (thread parity = head) ----
    const int h2 = tid & 1;
    float m = -__builtin_inff();
    for (int key = tid >> 1; key < d.Sk; key += 128) m = fmaxf(m, sc[key][h2]);
#pragma unroll
    for (int o = 2; o < 64; o <<= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane < 2) red[0][wave][lane] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0][0][h2], red[0][1][h2]), fmaxf(red[0][2][h2], red[0][3][h2]));
    const float mref = m > -__builtin_inff() ? m : 0.f;
    float l = 0.f;
    for (int key = tid >> 1; key < d.Sk; key += 128) {
        const float p = __expf(sc[key][h2] - mref);
        sc[key][h2] = p;
        l += p;
    }
#pragma unroll
    for (int o = 2; o < 64; o <<= 1) l += __shfl_xor(l, o);
    if (lane < 2) red[1][wave][lane] = l;
    __syncthreads();
    // ---- pass 2: weighted sum of V ----
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    for (int key0 = wave * 8; key0 < d.Sk; key0 += 32 * 4) {
        bf16x8 vv[4];
        float p[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int key = key0 + u * 32 + slot, kc = min(key, d.Sk - 1);
            vv[u] = *reinterpret_cast<const bf16x8*>(Vp + (int64_t)kc * d.v_rs);
            p[u] = key < d.Sk ? sc[kc][head] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += p[u] * (float)vv[u][i];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        acc[i] += __shfl_xor(acc[i], 8);
        acc[i] += __shfl_xor(acc[i], 16);
        acc[i] += __shfl_xor(acc[i], 32);
    }
    if (slot == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) part[wave][chunk * 8 + i] = acc[i];
    }
    __syncthreads();
    if (tid < 64) {
        const int hh = tid >> 5;
        const float lsum = red[1][0][hh] + red[1][1][hh] + red[1][2][hh] + red[1][3][hh];
        const float o = (part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid]) / lsum;
        ((bf16*)d.o)[b * d.o_bs + hp * 64 + tid] = (bf16)o;
    }
}

// ------------------------------------------------------------------------------------------------
// Decode step of one attention block with its projections inside: LayerNorm(x) -> q (and, self-attention, k / v of the new
// token, appended to the cache) -> attention over the cache -> o.  Replaces the layer-norm, q-projection, k/v-projection and
// attention launches of a KV-cached decoder step (4 launches of ~5-7 us of mostly launch latency each at B = 256) by one.
// Workgroup = (batch element, head pair) like attn_decode_kernel; every workgroup normalises its row itself (256 elements) and
// multiplies it with the 64 (x3) weight rows of its head pair: 32 (96) KB of weights per workgroup, L2 hits after the first.
// Rounding points are those of the separate kernels (LayerNorm output, q / k / v and o in bf16; fp32 accumulation).
// d = 256, H = 8 (head dim 32).
// ------------------------------------------------------------------------------------------------
// a K / V row of the cross-attention cache: read once per decode step and 139 MB per layer -- a streaming (non-temporal) load, so that the
// stream does not push the step's 10 MB of weights out of the L2 / infinity cache on its way through
template <bool STREAM>
__device__ __forceinline__ bf16x8 kv_load(const bf16* p) {
    if constexpr (STREAM) return __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(p));
    else return *reinterpret_cast<const bf16x8*>(p);
}

template <bool SELF, int NS>
__global__ __launch_bounds__(256) void attn_decode_fused_kernel(const RalfDecodeAttnDesc d) {
    __shared__ float sc[DEC_MAXK + 1][2];
    __shared__ float red[2][4][2];
    __shared__ float part[4][64];
    __shared__ float hrow[256];
    __shared__ float qkv[3][64];
    __shared__ float lnred[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hpairs = d.H / 2, hp = blockIdx.x % hpairs, b0 = (blockIdx.x / hpairs) * NS;
    const int chunk = lane & 7, slot = lane >> 3, head = chunk >> 2;
    constexpr int D = 256;
    constexpr int NP = SELF ? 3 : 1;
    // ---- the head pair's weight rows, once per workgroup: 8 lanes per row (one 128-byte line per load instruction and row),
    //      8 rows per wave-load.  (NS batch elements share them: with one element per workgroup the 1024 workgroups pulled
    //      100 MB of the same 384 KB through the L2 per call and the self-attention block took 21 us.)
    const int sub = lane >> 3;
    // cross-attention (one element per workgroup): the first 128 cached keys start their way BEFORE the weights, the LayerNorm and the
    // projection (all 1024 workgroups of a launch run that prologue at the same time, with nothing in flight from the K / V stream)
    constexpr bool PRE = !SELF && NS == 1;
    bf16x8 kpre[4];
    if (PRE) {
        const bf16* Kp0 = (const bf16*)d.kv + (int64_t)min(b0, d.B - 1) * d.kv_bs + (int64_t)hp * (d.kv_hs ? d.kv_hs : 64) + chunk * 8;
#pragma unroll
        for (int u = 0; u < 4; ++u) kpre[u] = kv_load<PRE>(Kp0 + (int64_t)min(wave * 8 + u * 32 + slot, d.Sk - 1) * d.kv_rs);
    }
    bf16x8 w[NP][2][4];
    float bias_r[NP][2];
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int row = ps * 32 + wave * 8 + sub;
            const bf16* wr = (const bf16*)d.W + (int64_t)(p * D + hp * 64 + row) * D + chunk * 8;
#pragma unroll
            for (int i = 0; i < 4; ++i) w[p][ps][i] = *reinterpret_cast<const bf16x8*>(wr + i * 64);
            bias_r[p][ps] = d.bias[p * D + hp * 64 + row];
        }
    const float ln_g = d.ln_g[tid], ln_b = d.ln_b[tid];
    for (int si = 0; si < NS; ++si) {
    const int b = b0 + si;
    if (b >= d.B) break;
    // ---- LayerNorm of the row ----
    {
        const float xv = (float)((const bf16*)d.x)[(int64_t)b * d.x_rs + tid];
        float s = xv;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) s += __shfl_xor(s, o);
        if (lane == 0) lnred[0][wave] = s;
        __syncthreads();
        const float mu = (lnred[0][0] + lnred[0][1] + lnred[0][2] + lnred[0][3]) * (1.f / D);
        const float dv = xv - mu;
        float q2 = dv * dv;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) q2 += __shfl_xor(q2, o);
        if (lane == 0) lnred[1][wave] = q2;
        __syncthreads();
        const float rs = rsqrtf((lnred[1][0] + lnred[1][1] + lnred[1][2] + lnred[1][3]) * (1.f / D) + d.eps);
        hrow[tid] = (float)(bf16)(dv * rs * ln_g + ln_b);
    }
    __syncthreads();
    // ---- projections ----
    {
        float hv[4][8];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) hv[i][e] = hrow[i * 64 + chunk * 8 + e];
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                float a = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int e = 0; e < 8; ++e) a += hv[i][e] * (float)w[p][ps][i][e];
                a += __shfl_xor(a, 1);
                a += __shfl_xor(a, 2);
                a += __shfl_xor(a, 4);
                if (chunk == 0) qkv[p][ps * 32 + wave * 8 + sub] = (float)(bf16)(a + bias_r[p][ps]);
            }
    }
    __syncthreads();
    bf16* KV = (bf16*)d.kv + (int64_t)b * d.kv_bs + (int64_t)hp * (d.kv_hs ? d.kv_hs : 64);
    const int64_t v_off = d.kv_hs ? d.kv_vo : D;   // head-pair-major cache: this pair's values follow its keys
    if (SELF && tid < 128) {   // the new token's k / v: row Sk of the cache (k at column 0, v at column d)
        const int which = tid >> 6, c = tid & 63;
        KV[(int64_t)d.Sk * d.kv_rs + which * D + c] = (bf16)qkv[1 + which][c];
    }
    const bf16* Kp = KV + chunk * 8;
    const bf16* Vp = KV + v_off + chunk * 8;
    const uint8_t* kpm = d.kpm ? d.kpm + (int64_t)b * d.kpm_bs : nullptr;
    float qv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) qv[i] = qkv[0][chunk * 8 + i] * d.scale;
    const int nk = SELF ? d.Sk + 1 : d.Sk;   // keys incl. the new one (whose k / v are still in LDS)
    // ---- pass 1: scores of the cached keys ----
    if (d.Sk > 0) {
        for (int key0 = wave * 8; key0 < d.Sk; key0 += 32 * 4) {
            bf16x8 kv[4];
            if (PRE && key0 == wave * 8) {
#pragma unroll
                for (int u = 0; u < 4; ++u) kv[u] = kpre[u];
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int key = min(key0 + u * 32 + slot, d.Sk - 1);
                    kv[u] = kv_load<PRE>(Kp + (int64_t)key * d.kv_rs);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int key = key0 + u * 32 + slot;
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) s += qv[i] * (float)kv[u][i];
                s += __shfl_xor(s, 1);
                s += __shfl_xor(s, 2);
                if ((chunk & 3) == 0 && key < d.Sk) sc[key][head] = (kpm && kpm[key]) ? -__builtin_inff() : s;
            }
        }
    }
    if (SELF && tid < 8) {   // the new key (lanes 0..7 of wave 0: slot 0)
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += qv[i] * qkv[1][chunk * 8 + i];
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        if ((chunk & 3) == 0) sc[d.Sk][head] = (kpm && kpm[d.Sk]) ? -__builtin_inff() : s;
    }
    // (cross-attention) the first 128 values start their way before the softmax: they do not depend on it
    bf16x8 vpre[4];
    if (PRE) {
#pragma unroll
        for (int u = 0; u < 4; ++u) vpre[u] = kv_load<PRE>(Vp + (int64_t)min(wave * 8 + u * 32 + slot, d.Sk - 1) * d.kv_rs);
    }
    __syncthreads();
    // ---- softmax over the keys, per head (thread parity = head) ----
    const int h2 = tid & 1;
    float m = -__builtin_inff();
    for (int key = tid >> 1; key < nk; key += 128) m = fmaxf(m, sc[key][h2]);
#pragma unroll
    for (int o = 2; o < 64; o <<= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane < 2) red[0][wave][lane] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0][0][h2], red[0][1][h2]), fmaxf(red[0][2][h2], red[0][3][h2]));
    const float mref = m > -__builtin_inff() ? m : 0.f;
    float l = 0.f;
    for (int key = tid >> 1; key < nk; key += 128) {
        const float p = __expf(sc[key][h2] - mref);
        sc[key][h2] = p;
        l += p;
    }
#pragma unroll
    for (int o = 2; o < 64; o <<= 1) l += __shfl_xor(l, o);
    if (lane < 2) red[1][wave][lane] = l;
    __syncthreads();
    // ---- pass 2: weighted sum of V ----
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    if (d.Sk > 0) {
        for (int key0 = wave * 8; key0 < d.Sk; key0 += 32 * 4) {
            bf16x8 vv[4];
            float p[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int key = key0 + u * 32 + slot, kc = min(key, d.Sk - 1);
                if (PRE && key0 == wave * 8) vv[u] = vpre[u];
                else vv[u] = kv_load<PRE>(Vp + (int64_t)kc * d.kv_rs);
                p[u] = key < d.Sk ? sc[kc][head] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] += p[u] * (float)vv[u][i];
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        acc[i] += __shfl_xor(acc[i], 8);
        acc[i] += __shfl_xor(acc[i], 16);
        acc[i] += __shfl_xor(acc[i], 32);
    }
    if (slot == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) part[wave][chunk * 8 + i] = acc[i];
    }
    __syncthreads();
    if (tid < 64) {
        const int hh = tid >> 5;
        const float lsum = red[1][0][hh] + red[1][1][hh] + red[1][2][hh] + red[1][3][hh];
        float o = part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid];
        if (SELF) o += sc[d.Sk][hh] * qkv[2][tid];
        ((bf16*)d.o)[(int64_t)b * d.o_rs + hp * 64 + tid] = (bf16)(o / lsum);
    }
    if (NS > 1) __syncthreads();   // the LDS arrays are reused by the next batch element
    }
}

// Self-attention block of a decode step, 4 batch elements per workgroup, one WAVE per element: the head pair's q / k / v weight rows
// (96 KB) are loaded once per workgroup, spread over the registers of its 4 waves; every wave multiplies ITS rows with the
// normalised rows of all 4 elements (LDS), and after one barrier each wave runs the whole attention of its own element (a few dozen
// cached keys) with wave-local reductions only.  (One element per workgroup: 1024 workgroups pulled 100 MB of the same 384 KB of
// weights through the L2, 21 us; four elements one after the other in a workgroup: 25 us of serial latency chains.)
__global__ __launch_bounds__(256) void attn_decode_self4_kernel(const RalfDecodeAttnDesc d) {
    constexpr int D = 256, NS = 4;
    __shared__ float hrow[NS][D];
    __shared__ float qkv[NS][3][64];
    __shared__ float sc[NS][DEC_MAXK + 1][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hpairs = d.H / 2, hp = blockIdx.x % hpairs, b0 = (blockIdx.x / hpairs) * NS;
    const int chunk = lane & 7, slot = lane >> 3, head = chunk >> 2, sub = slot;
    const int b = b0 + wave;                 // this wave's batch element
    const bool live = b < d.B;
    // keys already cached for THIS element: one count for the batch, or per element (d.pos: samples of a lock-step loop that rewind their
    // prefixes independently -- the relation task's back-tracking) -- wave-uniform either way
    const int Sk = d.pos ? d.pos[live ? b : d.B - 1] : d.Sk;
    // ---- weight rows of this wave: rows {wave*8 + sub, 32 + wave*8 + sub} of each of q, k, v ----
    bf16x8 w[3][2][4];
    float bias_r[3][2];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int row = ps * 32 + wave * 8 + sub;
            const bf16* wr = (const bf16*)d.W + (int64_t)(p * D + hp * 64 + row) * D + chunk * 8;
#pragma unroll
            for (int i = 0; i < 4; ++i) w[p][ps][i] = *reinterpret_cast<const bf16x8*>(wr + i * 64);
            bias_r[p][ps] = d.bias[p * D + hp * 64 + row];
        }
    // ---- the first 64 cached keys / values of this wave's element start their way now (rows < Sk were written by earlier steps):
    //      their latency hides behind the LayerNorm and the projections ----
    constexpr int NPRE = 8;                 // wave-loads of 8 keys
    bf16x8 kpre[NPRE], vpre[NPRE];
    {
        const int bb = live ? b : d.B - 1;
        const bf16* KVr = (const bf16*)d.kv + (int64_t)bb * d.kv_bs + hp * 64 + chunk * 8;
        const int last = Sk > 0 ? Sk - 1 : 0;
#pragma unroll
        for (int u = 0; u < NPRE; ++u) {
            const int key = min(u * 8 + slot, last);
            kpre[u] = *reinterpret_cast<const bf16x8*>(KVr + (int64_t)key * d.kv_rs);
            vpre[u] = *reinterpret_cast<const bf16x8*>(KVr + D + (int64_t)key * d.kv_rs);
        }
    }
    // ---- LayerNorm of this wave's row (lane = 4 consecutive elements) ----
    {
        const int bb = live ? b : d.B - 1;
        const bf16x4 xr = *reinterpret_cast<const bf16x4*>((const bf16*)d.x + (int64_t)bb * d.x_rs + lane * 4);
        float xv[4], s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { xv[i] = (float)xr[i]; s += xv[i]; }
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) s += __shfl_xor(s, o);
        const float mu = s * (1.f / D);
        float q2 = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { xv[i] -= mu; q2 += xv[i] * xv[i]; }
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) q2 += __shfl_xor(q2, o);
        const float rs = rsqrtf(q2 * (1.f / D) + d.eps);
        const float4 g = *reinterpret_cast<const float4*>(d.ln_g + lane * 4), be = *reinterpret_cast<const float4*>(d.ln_b + lane * 4);
        const float gg[4] = {g.x, g.y, g.z, g.w}, bb4[4] = {be.x, be.y, be.z, be.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) hrow[wave][lane * 4 + i] = (float)(bf16)(xv[i] * rs * gg[i] + bb4[i]);
    }
    __syncthreads();
    // ---- projections: this wave's 6 x 8 rows against all NS normalised rows ----
#pragma unroll 1
    for (int si = 0; si < NS; ++si) {
        float hv[4][8];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) hv[i][e] = hrow[si][i * 64 + chunk * 8 + e];
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                float a = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int e = 0; e < 8; ++e) a += hv[i][e] * (float)w[p][ps][i][e];
                a += __shfl_xor(a, 1);
                a += __shfl_xor(a, 2);
                a += __shfl_xor(a, 4);
                if (chunk == 0) qkv[si][p][ps * 32 + wave * 8 + sub] = (float)(bf16)(a + bias_r[p][ps]);
            }
    }
    __syncthreads();
    if (!live) return;
    // ---- from here on: one wave = one batch element, wave-local ----
    bf16* KV = (bf16*)d.kv + (int64_t)b * d.kv_bs + hp * 64;
    KV[(int64_t)Sk * d.kv_rs + lane] = (bf16)qkv[wave][1][lane];            // the new token's k / v: row Sk of the cache
    KV[(int64_t)Sk * d.kv_rs + D + lane] = (bf16)qkv[wave][2][lane];
    const bf16* Kp = KV + chunk * 8;
    const bf16* Vp = KV + D + chunk * 8;
    const uint8_t* kpm = d.kpm ? d.kpm + (int64_t)b * d.kpm_bs : nullptr;
    float qv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) qv[i] = qkv[wave][0][chunk * 8 + i] * d.scale;
    float (*scw)[2] = sc[wave];
    // scores of the cached keys: 8 keys per wave-load; the first 64 keys are already in registers
#pragma unroll
    for (int u = 0; u < NPRE; ++u) {
        const int key = u * 8 + slot;
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += qv[i] * (float)kpre[u][i];
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        if ((chunk & 3) == 0 && key < Sk) scw[key][head] = (kpm && kpm[key]) ? -__builtin_inff() : s;
    }
    for (int key0 = NPRE * 8; key0 < Sk; key0 += 32) {
        bf16x8 kv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int key = min(key0 + u * 8 + slot, Sk - 1);
            kv[u] = *reinterpret_cast<const bf16x8*>(Kp + (int64_t)key * d.kv_rs);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int key = key0 + u * 8 + slot;
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) s += qv[i] * (float)kv[u][i];
            s += __shfl_xor(s, 1);
            s += __shfl_xor(s, 2);
            if ((chunk & 3) == 0 && key < Sk) scw[key][head] = (kpm && kpm[key]) ? -__builtin_inff() : s;
        }
    }
    if (lane < 8) {   // the new key (slot 0)
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += qv[i] * qkv[wave][1][chunk * 8 + i];
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        if ((chunk & 3) == 0) scw[Sk][head] = (kpm && kpm[Sk]) ? -__builtin_inff() : s;
    }
    __builtin_amdgcn_wave_barrier();   // (scores written by other lanes of this wave: LDS accesses of a wave complete in order)
    const int nk = Sk + 1;
    // softmax per head: lane parity = head, 32 lanes per head
    const int h2 = lane & 1;
    float m = -__builtin_inff();
    for (int key = lane >> 1; key < nk; key += 32) m = fmaxf(m, scw[key][h2]);
#pragma unroll
    for (int o = 2; o < 64; o <<= 1) m = fmaxf(m, __shfl_xor(m, o));
    const float mref = m > -__builtin_inff() ? m : 0.f;
    float l = 0.f;
    for (int key = lane >> 1; key < nk; key += 32) {
        const float p = __expf(scw[key][h2] - mref);
        scw[key][h2] = p;
        l += p;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int o = 2; o < 64; o <<= 1) l += __shfl_xor(l, o);      // lane parity h2 holds the sum of head h2
    // weighted sum of V
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
#pragma unroll
    for (int u = 0; u < NPRE; ++u) {
        const int key = u * 8 + slot;
        const bool in = key < Sk;          // (rows beyond the prefix may hold anything: select, never multiply by zero)
        const float p = in ? scw[key][head] : 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += in ? p * (float)vpre[u][i] : 0.f;
    }
    for (int key0 = NPRE * 8; key0 < Sk; key0 += 32) {
        bf16x8 vv[4];
        float p[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int key = key0 + u * 8 + slot, kc = min(key, Sk - 1);
            vv[u] = *reinterpret_cast<const bf16x8*>(Vp + (int64_t)kc * d.kv_rs);
            p[u] = key < Sk ? scw[kc][head] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += p[u] * (float)vv[u][i];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        acc[i] += __shfl_xor(acc[i], 8);
        acc[i] += __shfl_xor(acc[i], 16);
        acc[i] += __shfl_xor(acc[i], 32);
    }
    // lanes 0..7 (slot 0) hold the 64 sums: chunk c -> columns 8c .. 8c+7 (head = c >> 2)
    const float l0 = __shfl(l, 0), l1 = __shfl(l, 1);
    if (slot == 0) {
        const float lsum = head ? l1 : l0, pn = scw[Sk][head];
        bf16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (bf16)((acc[i] + pn * qkv[wave][2][chunk * 8 + i]) / lsum);
        *reinterpret_cast<bf16x8*>((bf16*)d.o + (int64_t)b * d.o_rs + hp * 64 + chunk * 8) = o;
    }
}
}  // namespace

// called from attention.hip for dtype == RALF_BF16 (descriptor already validated)
int ralf_attention_fwd_mfma(const RalfAttnDesc& d, hipStream_t st) {
    if (d.Sq == 1 && d.dh == 32 && d.H % 2 == 0 && !d.causal && d.p_drop == 0.f && !d.lse && d.Sk <= DEC_MAXK &&
        d.k_rs % 8 == 0 && d.v_rs % 8 == 0 && d.k_bs % 8 == 0 && d.v_bs % 8 == 0 && d.q_bs % 8 == 0) {
        hipLaunchKernelGGL(attn_decode_kernel, dim3(d.B * (d.H / 2)), dim3(256), 0, st, d);
        return ralf::check_launch("attention_decode");
    }
    const dim3 grid(ceil_div(d.Sq, 64) * d.H * d.B);
    if (d.dh == 32) hipLaunchKernelGGL((attn_fwd_mfma<32>), grid, dim3(256), 0, st, d);
    else hipLaunchKernelGGL((attn_fwd_mfma<64>), grid, dim3(256), 0, st, d);
    return ralf::check_launch("attention_fwd_mfma");
}
int ralf_attention_bwd_mfma(const RalfAttnDesc& d, hipStream_t st) {
    // 128 .. 256 queries x keys per head (the image encoder's 16 x 16 tokens): dQ, dK, dV from one pass over the scores.  Measured inside the
    // train step (windowed rocprofv3 A/B): 61 us per call against 42 + 29 us for the per-key + per-query kernels at S = 256; on the decoder's
    // 50-row sequences the one-pass kernel is SLOWER (11.5 us against 4.0 + 4.3: one 8-wave workgroup per head, 158 KB of LDS = one
    // workgroup per CU, against two kernels whose 4-wave workgroups fill the CUs), so short sequences keep the two kernels.
    // RALF_ATTN_BWD_FUSED: 0 = never, 2 = wherever it fits (tests).
    static const int fused = [] { const char* e = getenv("RALF_ATTN_BWD_FUSED"); return e ? atoi(e) : 1; }();
    if (fused && d.dh == 32 && d.Sq <= FB_S && d.Sk <= FB_S && (fused == 2 || (d.Sq >= 128 && d.Sk >= 128)) && d.q_rs % 8 == 0 && d.k_rs % 8 == 0 && d.v_rs % 8 == 0 && d.o_rs % 8 == 0 && d.do_rs % 8 == 0 &&
        d.dq_rs % 2 == 0) {
        // RALF_ATTN_BWD_FUSED16=0: the 8-wave form (32 keys per wave, fp32 partial-dQ slots)
        static const int w16 = [] { const char* e = getenv("RALF_ATTN_BWD_FUSED16"); return e ? atoi(e) : 1; }();
        if (w16 && d.dq_rs % 4 == 0) hipLaunchKernelGGL(attn_bwd_fused16_mfma, dim3(d.H * d.B), dim3(1024), 0, st, d);
        else hipLaunchKernelGGL(attn_bwd_fused_mfma, dim3(d.H * d.B), dim3(512), 0, st, d);
        return ralf::check_launch("attention_bwd_fused");
    }
    // a few dozen queries over a long memory (the decoder's cross-attention): one pass, one workgroup per (batch, head).  RALF_ATTN_BWD_CROSS=0: the pair below
    static const int cross = [] { const char* e = getenv("RALF_ATTN_BWD_CROSS"); return e ? atoi(e) : 1; }();
    if (cross && d.dh == 32 && !d.causal && d.Sq <= XB_Q && (cross == 2 || d.Sk >= 128) && d.q_rs % 8 == 0 && d.k_rs % 8 == 0 && d.v_rs % 8 == 0 && d.o_rs % 8 == 0 &&
        d.do_rs % 8 == 0 && d.dq_rs % 2 == 0 && d.dk_rs % 4 == 0 && d.dv_rs % 4 == 0) {
        static const int occ = [] { const char* e = getenv("RALF_ATTN_CROSS_OCC"); return e ? atoi(e) : 2; }();   // (4: 128 registers, 41 of them spilled: 53 us against 40)
        if (occ == 4) hipLaunchKernelGGL(attn_bwd_cross_mfma<4>, dim3(d.H * d.B), dim3(512), 0, st, d);
        else hipLaunchKernelGGL(attn_bwd_cross_mfma<2>, dim3(d.H * d.B), dim3(512), 0, st, d);
        return ralf::check_launch("attention_bwd_cross");
    }
    const dim3 gq(ceil_div(d.Sq, 64) * d.H * d.B), gk(ceil_div(d.Sk, 64) * d.H * d.B);
    if (d.dh == 32) {
        hipLaunchKernelGGL((attn_bwd_dq_mfma<32>), gq, dim3(256), 0, st, d);
        hipLaunchKernelGGL((attn_bwd_dkv_mfma<32>), gk, dim3(256), 0, st, d);
    } else {
        hipLaunchKernelGGL((attn_bwd_dq_mfma<64>), gq, dim3(256), 0, st, d);
        hipLaunchKernelGGL((attn_bwd_dkv_mfma<64>), gk, dim3(256), 0, st, d);
    }
    return ralf::check_launch("attention_bwd_mfma");
}

extern "C" int ralf_decode_attn_max_keys(void) { return DEC_MAXK; }

extern "C" int ralf_decode_attn(const RalfDecodeAttnDesc* dp, void* stream) {
    RALF_REQUIRE(dp, "decode_attn: null descriptor");
    const RalfDecodeAttnDesc& d = *dp;
    RALF_REQUIRE(d.x && d.ln_g && d.ln_b && d.W && d.bias && d.kv && d.o, "decode_attn: null pointer");
    RALF_REQUIRE(d.d == 256 && d.H == 8 && d.B > 0 && d.Sk >= 0 && (d.self_ || d.Sk > 0) && d.Sk + (d.self_ ? 1 : 0) <= DEC_MAXK,
                 "decode_attn: needs d = 256, H = 8 and at most %d keys (got d=%d H=%d Sk=%d)", DEC_MAXK, d.d, d.H, d.Sk);
    RALF_REQUIRE(d.kv_rs % 8 == 0 && d.kv_bs % 8 == 0 && ((uintptr_t)d.kv % 16) == 0 && ((uintptr_t)d.W % 16) == 0 && (!d.kpm || d.kpm_bs > 0),
                 "decode_attn: cache rows and weights must be 16-byte aligned; kpm needs its row stride");
    RALF_REQUIRE(!d.pos || d.self_, "decode_attn: per-element positions belong to the self-attention block");
    RALF_REQUIRE(!d.kv_hs || (!d.self_ && d.kv_hs % 8 == 0 && d.kv_vo % 8 == 0), "decode_attn: the head-pair-major cache layout belongs to the cross-attention block (16-byte aligned strides)");
    hipStream_t st = (hipStream_t)stream;
    // self-attention (a few dozen keys): 4 batch elements share a workgroup's weight rows; cross-attention streams its K/V cache
    // from HBM and wants every workgroup it can get
    if (d.self_) hipLaunchKernelGGL(attn_decode_self4_kernel, dim3(ceil_div(d.B, 4) * 4), dim3(256), 0, st, d);
    else hipLaunchKernelGGL((attn_decode_fused_kernel<false, 1>), dim3(d.B * 4), dim3(256), 0, st, d);
    return ralf::check_launch("decode_attn");
}
