// Row-strip GEMM for the transformer's linear layers on gfx950 (bf16 operands, fp32 accumulate):
//
//     y[m, :] = epi( LN?(x)[m, :] @ W^T  (forward, W = nn.Linear.weight [N, K])      or      x[m, :] @ W  (data gradient, W [K, N]) )
//
// Replaces the F.linear / F.layer_norm pairs of nn.TransformerEncoderLayer / DecoderLayer / MultiheadAttention
// (image2layout/train/models/retrieval_augmented_autoreg.py:116-126, common/common.py:25-34), FeedForward and Attention
// (common/attention.py:15-71) and their data-gradient products.
//
// Why not the tiled kernel of gemm_impl.h: with d_model = 256 these products are HBM- and latency-bound, not MFMA-bound (194 flop
// per byte of minimal traffic against a machine balance of ~500): a 128x128 tile of a K = 256 product is four k-steps between a
// cold start and a store tail, three such workgroups per CU.  Here ONE workgroup owns a strip of 64 rows for the whole
// product:
//   * the strip of x (64 x K, K <= 512) is loaded ONCE and stays in LDS -- optionally LayerNorm-ed in place first (the norm's
//     statistics and normalised rows are written out for the backward pass), so LN -> Linear is one launch and x is read once;
//     longer reductions (K = 768 / 1024: data gradients, FFN second layer) stream 64x64 pieces of x beside the weights;
//   * the weights stream through a double-buffered LDS tile in ONE flat loop over (column tile, k-step) with one barrier per
//     step: the loads of step s+1 fly under the MFMAs of step s across column-tile boundaries, and the epilogue of a column
//     tile runs while the next tile's first weights are already on their way;
//   * 8 waves (2 x 4, 32 x 32 or 32 x 64 per wave), v_mfma_f32_32x32x16_bf16 on the TRANSPOSED tile (weights as the row
//     operand) so a lane ends up with 4 consecutive output columns; the epilogue goes through an fp32 LDS tile: 16-byte
//     coalesced stores, bias / ReLU / GELU / dropout / residual / pre-activation copy fused.
//   * [K][N]-stored weights (data gradient) are fed with ds_read_b64_tr_b16: no transposed copy of W exists.
#include "common.h"

namespace {
typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

constexpr int BM = 64, BK = 64, NT = 512;
constexpr int LDK = BK + 8;   // k-contiguous LDS rows: 64 + 8 pad

__device__ __forceinline__ uint32_t rs_rng24(uint64_t seed, uint64_t call, uint64_t idx) {   // = rng24 of gemm_impl.h / pointwise.hip
    uint64_t z = seed + call * 0x9E3779B97F4A7C15ull + idx * 0xD1B54A32D192ED03ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (uint32_t)(z >> 40);
}
__device__ __forceinline__ float rs_gelu(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float rs_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <bool BKC, int BN, int KRES>
constexpr int rs_lds_bytes() {
    constexpr int a = KRES ? BM * (KRES + 8) * 2 : 2 * BM * LDK * 2;
    constexpr int b = 2 * (BKC ? BN * LDK : BK * (BN + 8)) * 2;
    constexpr int c = BM * (BN + 4) * 4;
    return BN == 256 ? a + (b > c ? b : c) : a + b + c;   // BN = 256 (a single column tile): the C staging tile re-uses the weight buffers
}

// BKC: W stored [N][K] (k contiguous: forward) else [K][N] (data gradient).  BN: columns per tile (256 only for N <= 256).
// KRES: 0 = x streamed in 64x64 pieces, else K (256 / 512): x strip resident, LayerNorm prologue possible (KRES = 256 = the row).
template <bool BKC, int BN, int KRES>
__global__ __launch_bounds__(NT) void rs_gemm_kernel(const RalfRsDesc d) {
    constexpr int WFN = BN / 128;                 // 32x32 fragments per wave along n
    constexpr int LDA = KRES ? KRES + 8 : LDK;
    constexpr int LDB = BKC ? LDK : BN + 8;
    constexpr int A_ELEMS = KRES ? BM * LDA : 2 * BM * LDK;
    constexpr int B_ELEMS = BKC ? BN * LDK : BK * (BN + 8);
    constexpr int CP = BN + 4;
    constexpr int NVB = BN * BK / 8 / NT;         // 16-byte vectors of a weight piece per thread
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[rs_lds_bytes<BKC, BN, KRES>()];
    bf16* la = reinterpret_cast<bf16*>(lds_raw);
    bf16* lb = la + A_ELEMS;
    float* cs = BN == 256 ? reinterpret_cast<float*>(lb) : reinterpret_cast<float*>(lb + 2 * B_ELEMS);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave & 1, wn = wave >> 1;
    const int l31 = lane & 31, lh = lane >> 5;
    const int trL = lane & 15, tr_rowblk = ((lane >> 4) & 1) * 16, tr_k = lh * 8 + (trL >> 2), tr_c = (trL & 3) * 4;
    const int m0 = blockIdx.x * BM;
    const int M = d.M, N = d.N, K = d.K;
    const int KT = K / BK, NTILES = (N + BN - 1) / BN, S = NTILES * KT;
    const bf16* X = (const bf16*)d.x;
    const bf16* W = (const bf16*)d.w;

    // ---- x strip (resident) + LayerNorm prologue ----
    if constexpr (KRES != 0) {
        constexpr int VPR = KRES / 8;
        for (int v = tid; v < BM * VPR; v += NT) {
            const int r = v / VPR, c = v % VPR;
            u32x4 t = {0u, 0u, 0u, 0u};
            if (m0 + r < M) t = *reinterpret_cast<const u32x4*>(X + (int64_t)(m0 + r) * d.ldx + c * 8);
            *reinterpret_cast<u32x4*>(la + r * LDA + c * 8) = t;
        }
        if (KRES == 256 && d.ln_gamma) {
            __syncthreads();
            float g[4], b[4];
            *reinterpret_cast<float4*>(g) = *reinterpret_cast<const float4*>(d.ln_gamma + lane * 4);
            *reinterpret_cast<float4*>(b) = *reinterpret_cast<const float4*>(d.ln_beta + lane * 4);
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {   // a wave normalises 8 rows, a lane 4 columns
                const int r = wave * 8 + rr;
                bf16* row = la + r * LDA + lane * 4;
                const bf16x4 t = *reinterpret_cast<const bf16x4*>(row);
                float v[4] = {(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
                const float mu = rs_wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.f / 256.f);
                float q = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) { v[i] -= mu; q += v[i] * v[i]; }
                const float rs = rsqrtf(rs_wave_sum(q) * (1.f / 256.f) + d.ln_eps);
                bf16x4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (bf16)(v[i] * rs * g[i] + b[i]);
                *reinterpret_cast<bf16x4*>(row) = o;
                if (m0 + r < M) {
                    if (d.xln) *reinterpret_cast<bf16x4*>((bf16*)d.xln + (int64_t)(m0 + r) * 256 + lane * 4) = o;
                    if (lane == 0 && d.ln_mean) { d.ln_mean[m0 + r] = mu; d.ln_rstd[m0 + r] = rs; }
                }
            }
        }
    }

    // ---- streaming loaders: a ring of 3 register stages in front of the 2 LDS buffers.  One workgroup per CU means no other
    // workgroup hides a load's latency: with the loads of step s+1 issued at the top of step s, a step cost a full L2 round trip
    // (~1 us measured).  The data of step s+1 is now requested two steps earlier. ----
    constexpr int RING = 3;
    u32x4 rb[RING][NVB], ra[RING];
    auto gload = [&](int stage, int nt, int kt) {
        const int n0 = nt * BN, k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < NVB; ++i) {
            const int v = tid + NT * i;
            if constexpr (BKC) {
                const int r = v >> 3, c = v & 7;
                const int n = min(n0 + r, N - 1);
                rb[stage][i] = *reinterpret_cast<const u32x4*>(W + (int64_t)n * d.ldw + k0 + c * 8);
            } else {
                constexpr int VR = BN / 8;
                const int r = v / VR, c = v % VR;
                const int n = min(n0 + c * 8, N - 8);
                rb[stage][i] = *reinterpret_cast<const u32x4*>(W + (int64_t)(k0 + r) * d.ldw + n);
            }
        }
        if constexpr (KRES == 0) {
            const int r = tid >> 3, c = tid & 7;
            const int m = min(m0 + r, M - 1);
            ra[stage] = *reinterpret_cast<const u32x4*>(X + (int64_t)m * d.ldx + k0 + c * 8);
        }
    };
    auto lstore = [&](int stage, int buf) {
        bf16* b = lb + buf * B_ELEMS;
#pragma unroll
        for (int i = 0; i < NVB; ++i) {
            const int v = tid + NT * i;
            if constexpr (BKC) *reinterpret_cast<u32x4*>(b + (v >> 3) * LDK + (v & 7) * 8) = rb[stage][i];
            else { constexpr int VR = BN / 8; *reinterpret_cast<u32x4*>(b + (v / VR) * LDB + (v % VR) * 8) = rb[stage][i]; }
        }
        if constexpr (KRES == 0) *reinterpret_cast<u32x4*>(la + buf * BM * LDK + (tid >> 3) * LDK + (tid & 7) * 8) = ra[stage];
    };

    f32x16 acc[WFN];
#pragma unroll
    for (int j = 0; j < WFN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    auto compute = [&](const int buf, const int kt) {
        const bf16* b = lb + buf * B_ELEMS;
        const bf16* a = KRES ? la + kt * BK : la + buf * BM * LDK;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const bf16x8 af = *reinterpret_cast<const bf16x8*>(a + (wm * 32 + l31) * LDA + ks * 16 + lh * 8);
#pragma unroll
            for (int j = 0; j < WFN; ++j) {
                bf16x8 bfr;
                if constexpr (BKC) {
                    bfr = *reinterpret_cast<const bf16x8*>(b + (wn * 32 * WFN + j * 32 + l31) * LDK + ks * 16 + lh * 8);
                } else {
                    const bf16* q = b + (ks * 16 + tr_k) * LDB + wn * 32 * WFN + j * 32 + tr_rowblk + tr_c;
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, q));
                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, q + 4 * LDB));
                    bfr = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr, af, acc[j], 0, 0, 0);
            }
        }
    };

    // epilogue of column tile nt: accumulators -> fp32 LDS tile -> 8 consecutive columns of one row per thread
    auto epilogue = [&](int nt) {
        const int n0 = nt * BN;
#pragma unroll
        for (int j = 0; j < WFN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(cs + (wm * 32 + l31) * CP + wn * 32 * WFN + j * 32 + 8 * g + 4 * lh) =
                    make_float4(acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]);
#pragma unroll
        for (int j = 0; j < WFN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        __syncthreads();
        constexpr int CG = BN / 8, RPP = NT / CG;
        const uint32_t thr = (uint32_t)(d.drop_p * 16777216.f);
        const float inv = 1.f / (1.f - d.drop_p);
        const uint64_t sd = d.drop_p > 0.f ? (uint64_t)d.seed[0] : 0;
#pragma unroll
        for (int p = 0; p < BM / RPP; ++p) {
            const int lr = p * RPP + tid / CG, c = (tid % CG) * 8;
            const int m = m0 + lr, n = n0 + c;
            if (m < M && n < N) {
                const float4 lo = *reinterpret_cast<const float4*>(cs + lr * CP + c), hi = *reinterpret_cast<const float4*>(cs + lr * CP + c + 4);
                float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                if (d.bias) {
                    const float4 b0 = *reinterpret_cast<const float4*>(d.bias + n), b1 = *reinterpret_cast<const float4*>(d.bias + n + 4);
                    v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
                }
                const int64_t off = (int64_t)m * d.ldy + n;
                if (d.y2) {   // pre-activation copy (GELU gradient)
                    bf16x8 t;
#pragma unroll
                    for (int q = 0; q < 8; ++q) t[q] = (bf16)v[q];
                    *reinterpret_cast<bf16x8*>((bf16*)d.y2 + off) = t;
                }
                if (d.act == RALF_ACT_RELU) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] = fmaxf(v[q], 0.f);
                } else if (d.act == RALF_ACT_GELU) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] = rs_gelu(v[q]);
                }
                if (d.drop_p > 0.f) {
                    const uint64_t e0 = (uint64_t)m * N + n;
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] = rs_rng24(sd, d.call_id, e0 + q) >= thr ? v[q] * inv : 0.f;
                }
                if (d.aux) {   // ReLU gradient mask from the saved activation: v = aux > 0 ? v * aux_scale : 0
                    const bf16x8 a = *reinterpret_cast<const bf16x8*>((const bf16*)d.aux + off);
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] = (float)a[q] > 0.f ? v[q] * d.aux_scale : 0.f;
                }
                if (d.res) {
                    const bf16x8 r8 = *reinterpret_cast<const bf16x8*>((const bf16*)d.res + (int64_t)m * d.ldr + n);
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] += (float)r8[q];
                }
                if (d.out_f32) {
                    *reinterpret_cast<float4*>((float*)d.y + off) = make_float4(v[0], v[1], v[2], v[3]);
                    *reinterpret_cast<float4*>((float*)d.y + off + 4) = make_float4(v[4], v[5], v[6], v[7]);
                } else {
                    bf16x8 t;
#pragma unroll
                    for (int q = 0; q < 8; ++q) t[q] = (bf16)v[q];
                    *reinterpret_cast<bf16x8*>((bf16*)d.y + off) = t;
                }
            }
        }
    };

    // ---- one flat loop over (column tile, k-step), one barrier per step.  At the top of step s: LDS buffer s&1 holds step s,
    // register stage (s+1)%3 holds step s+1 (requested two steps ago), stage (s+2)%3 step s+2; step s+3 is requested now.
    // Unrolled by 6 so that ring stage and buffer parity are compile-time constants (no runtime-indexed register arrays). ----
    auto step_of = [&](int s, int& nt, int& kt) { nt = s / KT; kt = s - nt * KT; };
    {
        int nt_, kt_;
        gload(0, 0, 0);
        if (S > 1) { step_of(1, nt_, kt_); gload(1, nt_, kt_); }
        if (S > 2) { step_of(2, nt_, kt_); gload(2, nt_, kt_); }
        lstore(0, 0);
        __syncthreads();
    }
    for (int sb = 0; sb < S; sb += 6) {
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int s = sb + u;
            if (s < S) {   // (uniform)
                int nt, kt;
                step_of(s, nt, kt);
                if (s + 3 < S) { int n3, k3; step_of(s + 3, n3, k3); gload(u % 3, n3, k3); }   // stage (s+3)%3 = s%3 was stored a step ago
                compute(u & 1, kt);
                if (s + 1 < S) lstore((u + 1) % 3, (u + 1) & 1);
                __syncthreads();
                if (kt == KT - 1) epilogue(nt);   // (ends with reads of cs only: the next write of cs is at least one barrier away)
            }
        }
    }
}

template <bool BKC, int BN, int KRES>
int rs_launch(const RalfRsDesc& d, hipStream_t st) {
    hipLaunchKernelGGL((rs_gemm_kernel<BKC, BN, KRES>), dim3(ceil_div(d.M, BM)), dim3(NT), 0, st, d);
    return ralf::check_launch("rs_gemm");
}
template <bool BKC>
int rs_dispatch(const RalfRsDesc& d, hipStream_t st) {
    const bool wide = d.N <= 256 && d.N > 128;
    if (d.K == 256) return wide ? rs_launch<BKC, 256, 256>(d, st) : rs_launch<BKC, 128, 256>(d, st);
    if (d.K == 512) return wide ? rs_launch<BKC, 256, 0>(d, st) : rs_launch<BKC, 128, 512>(d, st);
    return wide ? rs_launch<BKC, 256, 0>(d, st) : rs_launch<BKC, 128, 0>(d, st);
}
}  // namespace

extern "C" int ralf_rs_gemm(const RalfRsDesc* dp, void* stream) {
    RALF_REQUIRE(dp, "rs_gemm: null descriptor");
    const RalfRsDesc& d = *dp;
    RALF_REQUIRE(d.x && d.w && d.y && d.M > 0 && d.N >= 8 && d.K >= 64, "rs_gemm: bad operands (M=%d N=%d K=%d)", d.M, d.N, d.K);
    RALF_REQUIRE(d.K % 64 == 0 && d.N % 8 == 0, "rs_gemm: K %% 64 and N %% 8 (K=%d N=%d)", d.K, d.N);
    RALF_REQUIRE(d.ldx % 8 == 0 && d.ldw % 8 == 0 && d.ldy % 8 == 0 && (!d.res || d.ldr % 8 == 0), "rs_gemm: leading dimensions must be multiples of 8");
    auto al = [](const void* p) { return (((uintptr_t)p) & 15) == 0; };
    RALF_REQUIRE(al(d.x) && al(d.w) && al(d.y) && al(d.y2) && al(d.res) && al(d.aux) && al(d.bias) && al(d.xln) && al(d.ln_gamma) && al(d.ln_beta),
                 "rs_gemm: operands must be 16-byte aligned");
    RALF_REQUIRE(!d.ln_gamma || (d.K == 256 && d.ln_beta && d.w_kcontig), "rs_gemm: the LayerNorm prologue needs K = 256 (the normalised row) and a forward product");
    RALF_REQUIRE(d.drop_p >= 0.f && d.drop_p < 1.f && (d.drop_p == 0.f || (d.seed && d.ldy == d.N)), "rs_gemm: fused dropout needs a seed and a contiguous [M,N] output");
    RALF_REQUIRE(!d.out_f32 || !d.y2, "rs_gemm: the pre-activation copy is bf16 only");
    hipStream_t st = (hipStream_t)stream;
    return d.w_kcontig ? rs_dispatch<true>(d, st) : rs_dispatch<false>(d, st);
}
