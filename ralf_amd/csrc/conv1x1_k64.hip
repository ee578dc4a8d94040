// 1x1 convolutions with 64 INPUT channels on NHWC bf16 activations (gfx950): y[M, N] = x[M, 64] w[N, 64]^T, N % 64 == 0, M % 64 == 0,
// optionally with the per-64-row column statistics of the BatchNorm that follows (training) or that BatchNorm's eval-mode scale / shift,
// residual and ReLU (inference).
//
// Replaces torch's conv2d for timm Bottleneck.conv3 / downsample.0 / the first conv1 of ResNet-50's layer1 (image2layout/train/models/common/
// image.py:39-48: 64 -> 256 and 64 -> 64 channels on the 64 x 64 maps): at B = 64 these are M = 262 144 rows -- one k-tile of matrix work per
// output tile in front of a store four times the size of the operand.  The general GEMM (gemm_impl.h) spends such a tile on its latency chain
// (operands -> LDS -> MFMA -> staging tile -> store, two workgroups per CU): 40 us plain, 54 us with the statistics epilogue in isolation
// (4.2 / 3.1 TB/s), up to 111 us inside the train step, where a plain stream gets 6 TB/s on this chip.
//
// Here every WAVE is an independent worker on 64 x 64 output tiles: its 64 weight rows live in registers for the whole launch (8 fragments),
// the A fragments of a tile come straight from global memory (each lane loads the 16-byte k-slices of its own rows: no LDS, no barrier), the
// next tile's are requested before the current tile is stored, and the accumulators pass through a wave-private LDS tile so that stores are
// whole 128-byte row segments.  The 4 waves of a workgroup take consecutive tiles (one row block's four column groups at N = 256: the A rows
// are shared through the L1).  The MFMA chain of a tile (four k-steps in order) is the tiled kernel's: the same output bits.
// Measured and dropped: a 32-column half of the tile at a time (32 accumulator registers less: 151 VGPRs) with three workgroups per CU --
// 32.4 / 36.0 us against 30.7 / 33.0 (plain / statistics), and the inference form 367 against 323 us: two workgroups of four waves per CU it is.
#include <algorithm>

#include "common.h"

namespace {
typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// staging row stride in elements: bf16 rows of 64 + 8 (144 bytes) for the plain / statistics form (the statistics are taken on the values AS
// STORED); fp32 rows of 64 + 4 (272 bytes) for the inference form, whose residual is added before the one rounding (the tiled kernel's order)
template <int MODE> struct Stage { typedef bf16 T; static constexpr int LD = 72; };
template <> struct Stage<1> { typedef float T; static constexpr int LD = 68; };

struct CParams {
    const bf16* x; const bf16* w; bf16* y;
    float* colstats;                 // [M / 64, 2, N] or NULL
    const float* scale; const float* shift; const bf16* res;   // inference epilogue (scale NULL = off): relu?(acc * scale + shift (+ res))
    int M, N, relu, relu_post;
    int ntiles, ncg;                 // (M / 64) * (N / 64) tiles, column groups per row block
};

// KS = K / 16 k-steps: 4 (64 input channels: the next tile's rows are prefetched) or 8 (128: weights + one tile's rows + accumulators fill the
// 256 registers of two waves per SIMD, no prefetch -- the eight waves of a CU cover for each other)
template <int MODE, int KS>   // MODE 0: plain / statistics, 1: scale + shift (+ residual) (+ ReLU)
__global__ __launch_bounds__(256, 2) void conv1x1_k64_kernel(const CParams P) {
    constexpr int K = 16 * KS;
    constexpr bool PREF = KS == 4;
    typedef typename Stage<MODE>::T ST;
    constexpr int SLD = Stage<MODE>::LD;
    __shared__ __attribute__((aligned(16))) ST stage[4][64 * SLD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, lh = lane >> 5;
    ST* st = stage[wave];
    const int stride = gridDim.x * 4;
    int t = blockIdx.x * 4 + wave;
    if (t >= P.ntiles) return;
    const int cg = t % P.ncg, n0 = cg * 64;   // (stride % ncg == 0: a wave keeps its column group)
    // ---- this wave's weight rows: fragments j (32 rows each) x k-step s ----
    bf16x8 wf[2][KS];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int s = 0; s < KS; ++s) wf[j][s] = *reinterpret_cast<const bf16x8*>(P.w + (int64_t)(n0 + 32 * j + l31) * K + 16 * s + 8 * lh);
    bf16x8 af[2][KS], an[PREF ? 2 : 1][PREF ? KS : 1];
    auto load = [&](bf16x8 (&a)[2][KS], int tile) {
        const int64_t m0 = (int64_t)(tile / P.ncg) * 64;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int s = 0; s < KS; ++s) a[i][s] = *reinterpret_cast<const bf16x8*>(P.x + (m0 + 32 * i + l31) * K + 16 * s + 8 * lh);
    };
    load(af, t);
    for (; t < P.ntiles; t += stride) {
        const int tn = t + stride;
        if constexpr (PREF) {
            if (tn < P.ntiles) load(an, tn);   // the next tile's rows fly under this tile's MFMAs, staging and stores
        }
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
                for (int s = 0; s < KS; ++s) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j][s], af[i][s], acc[i][j], 0, 0, 0);
            }
        const int64_t m0 = (int64_t)(t / P.ncg) * 64;
        // ---- accumulators -> wave-private staging tile (register r of a lane: column (r & 3) + 8 (r >> 2) + 4 lh of the fragment, row l31) ----
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = 32 * j + 8 * g + 4 * lh;
                    float v[4] = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                    if (MODE == 1) {
                        const float4 sc = *reinterpret_cast<const float4*>(P.scale + n0 + c), sh = *reinterpret_cast<const float4*>(P.shift + n0 + c);
                        v[0] = v[0] * sc.x + sh.x; v[1] = v[1] * sc.y + sh.y; v[2] = v[2] * sc.z + sh.z; v[3] = v[3] * sc.w + sh.w;
                        if (P.relu && !P.relu_post) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                    }
                    if constexpr (MODE == 1) {
                        *reinterpret_cast<float4*>(st + (32 * i + l31) * SLD + c) = make_float4(v[0], v[1], v[2], v[3]);
                    } else {
                        bf16x4 o;
                        o[0] = (bf16)v[0]; o[1] = (bf16)v[1]; o[2] = (bf16)v[2]; o[3] = (bf16)v[3];
                        *reinterpret_cast<bf16x4*>(st + (32 * i + l31) * SLD + c) = o;
                    }
                }
        __builtin_amdgcn_wave_barrier();
        // ---- whole 128-byte row segments out: 8 lanes per row, 8 rows per store instruction ----
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int row = p * 8 + (lane >> 3), col = (lane & 7) * 8;
            const int64_t off = (m0 + row) * P.N + n0 + col;
            bf16x8 o;
            if constexpr (MODE == 1) {   // the tiled kernel's order: scale / shift (+ ReLU), + residual in fp32, (ReLU), ONE rounding
                const float4 lo = *reinterpret_cast<const float4*>(st + row * SLD + col), hi = *reinterpret_cast<const float4*>(st + row * SLD + col + 4);
                float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                if (P.res) {
                    const bf16x8 rv = *reinterpret_cast<const bf16x8*>(P.res + off);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += (float)rv[e];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (bf16)((P.relu && P.relu_post) ? fmaxf(v[e], 0.f) : v[e]);
            } else {
                o = *reinterpret_cast<const bf16x8*>(st + row * SLD + col);
            }
            *reinterpret_cast<bf16x8*>(P.y + off) = o;
        }
        if (MODE == 0 && P.colstats) {   // column sums of this 64-row block on the values as stored (lane = column)
            float s1 = 0.f, s2 = 0.f;
#pragma unroll 16
            for (int r = 0; r < 64; ++r) {
                const float v = (float)st[r * SLD + lane];
                s1 += v;
                s2 = __fmaf_rn(v, v, s2);
            }
            float* pr = P.colstats + (m0 / 64) * 2 * P.N + n0 + lane;
            pr[0] = s1;
            pr[P.N] = s2;
        }
        __builtin_amdgcn_wave_barrier();   // the staging tile is rewritten by the next tile
        if constexpr (PREF) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int s = 0; s < KS; ++s) af[i][s] = an[i][s];
        } else {
            if (tn < P.ntiles) load(af, tn);
        }
    }
}
}  // namespace

/* y [M, N] = x [M, K] w [N, K]^T on bf16 (1x1 convolution with K = 64 or 128 input channels, NHWC rows); M % 64 == 0, N % 64 == 0, N / 64 a power of two <= 32.
 * colstats (may be NULL): fp32 [M / 64, 2, N] -- per 64-row block the column sums and sums of squares of y AS STORED (the BatchNorm
 *   statistics partials ralf_gemm's colstats epilogue writes; consumed by ralf_bn_* the same way).
 * scale / shift (may be NULL together; then res / relu are ignored): y = act(acc * scale[n] + shift[n]) (+ res, ReLU after it when relu == 2):
 *   eval-mode BatchNorm (+ residual + ReLU) of the inference backbone, RalfGemmDesc.colscale / bias / res / RALF_ACT_RELU(_POST). */
extern "C" int ralf_conv1x1_k64(const void* x, const void* w, void* y, float* colstats, const float* scale, const float* shift, const void* res,
                                int relu, int64_t M, int N, int K, void* stream) {
    RALF_REQUIRE(x && w && y && M > 0 && N > 0, "conv1x1_k64: bad arguments");
    RALF_REQUIRE(K == 64 || K == 128, "conv1x1_k64: 64 or 128 input channels (K=%d)", K);
    RALF_REQUIRE(M % 64 == 0 && N % 64 == 0 && N <= 2048 && ((N / 64) & (N / 64 - 1)) == 0, "conv1x1_k64: M %% 64 == 0, N = 64 * 2^k <= 2048 (M=%lld N=%d)", (long long)M, N);
    RALF_REQUIRE(M * (int64_t)N < (1ll << 40) && (M / 64) * (int64_t)(N / 64) < (1ll << 31), "conv1x1_k64: too many tiles");
    RALF_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)w % 16) == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)res % 16) == 0 &&
                 ((uintptr_t)scale % 16) == 0 && ((uintptr_t)shift % 16) == 0, "conv1x1_k64: operands must be 16-byte aligned");
    RALF_REQUIRE(!scale == !shift && !(scale && colstats) && relu >= 0 && relu <= 2, "conv1x1_k64: scale and shift go together, without statistics; relu 0, 1 or 2 (after the residual)");
    CParams P;
    P.x = (const bf16*)x; P.w = (const bf16*)w; P.y = (bf16*)y; P.colstats = colstats;
    P.scale = scale; P.shift = shift; P.res = scale ? (const bf16*)res : nullptr;
    P.M = (int)M; P.N = N; P.relu = relu != 0; P.relu_post = relu == 2;
    P.ncg = N / 64; P.ntiles = (int)((M / 64) * P.ncg);
    // persistent waves: two workgroups of four per CU (256 CUs); a wave's tile stride (4 * grid) is a multiple of the column groups
    const int grid = (int)std::min<int64_t>(512, (P.ntiles + 3) / 4);
    int g = grid;
    while ((4 * g) % P.ncg) ++g;
    hipStream_t st = (hipStream_t)stream;
    if (K == 64) {
        if (scale) hipLaunchKernelGGL((conv1x1_k64_kernel<1, 4>), dim3(g), dim3(256), 0, st, P);
        else hipLaunchKernelGGL((conv1x1_k64_kernel<0, 4>), dim3(g), dim3(256), 0, st, P);
    } else {
        if (scale) hipLaunchKernelGGL((conv1x1_k64_kernel<1, 8>), dim3(g), dim3(256), 0, st, P);
        else hipLaunchKernelGGL((conv1x1_k64_kernel<0, 8>), dim3(g), dim3(256), 0, st, P);
    }
    return ralf::check_launch("conv1x1_k64");
}
