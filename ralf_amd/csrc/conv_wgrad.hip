// Weight gradient of a 3x3 / pad 1 convolution (stride 1 or 2) on NHWC bf16 activations, DIRECT form (gfx950):
//
//   dW[co][kh][kw][ci] = sum over (b, oy, ox) of  dy[b, oy, ox, co] * x[b, s oy + kh - 1, s ox + kw - 1, ci]
//
// Replaces the weight-gradient half of torch's conv2d backward for the 3x3 convolutions of the ResNet bottlenecks
// (image2layout/train/models/common/image.py:39-48: timm Bottleneck.conv2).  The implicit-GEMM form (gemm_impl.h, gather = 2) builds the
// im2col operand [pixels][9 Ci] on the fly: every input pixel travels global -> LDS nine times and every MFMA needs two LDS fragment reads,
// which is where that kernel sits (360 TFLOP/s, 53-58 us per layer).  Here a workgroup owns a [64 co] x [64 ci] x [9 taps] block of dW and
// walks pixel tiles of 64 output pixels (whole image rows): per tile it stages the 64 x 64 slice of dy and the (R + 2) x (W + 2) halo patch of x
// ONCE, and the nine taps are nine MFMA chains over the same dy fragments with the x fragments read at a tap-shifted patch row -- 2.4-4.5 x
// fewer operand bytes per MAC and 1.1 LDS fragment reads per MFMA.
//
//   grid   = (Co / 64) * (Ci / 64) * nsplit workgroups of 512 threads (8 waves: 2 x 2 over the 64 x 64 block, two tap groups {0..4}, {5..8})
//   split s walks pixel tiles s, s + nsplit, ... and leaves its block in partial[s][Co][9][Ci] (fp32);
//   conv3x3_wgrad_reduce_kernel sums the splits in order (deterministic) straight into the OIHW fp32 gradient.
//
// Both operands are [k = pixel][row] tiles in LDS (row = co for dy, ci for x) and reach the matrix cores through ds_read_b64_tr_b16, as the
// TN products of gemm_impl.h do; the accumulator layout (transposed tile: a lane holds 4 consecutive ci) is that kernel's too.
#include "common.h"

namespace {
typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

constexpr int LD = 72;            // LDS row stride in elements: 64 channels + 8 pad (conflict-light transpose reads, as gemm_impl.h RPAD)
// largest halo patch of a 64-output-pixel tile and its 16-byte vectors per thread: stride 1: one row of 64 pixels -> 3 x 66 (512 x 4 >= 198 x 8);
// stride 2: two rows of 32 -> 5 x 65 input pixels (512 x 6 >= 325 x 8)
template <int S> struct PatchCfg;
template <> struct PatchCfg<1> { static constexpr int PPMAX = 3 * 66, PSLOTS = 4; };
template <> struct PatchCfg<2> { static constexpr int PPMAX = 5 * 65, PSLOTS = 6; };

struct WParams {
    const bf16* dy; const bf16* x; float* partial;
    int B, H, W, Co, Ci, R, nsplit, ntiles, tiles_per_img;   // H, W: the OUTPUT (dy) grid; R = 64 / W rows per pixel tile; ntiles = B * H / R
    int IH, IW;                                              // the input (x) grid: H, W at stride 1; 2 H, 2 W (or one less) at stride 2
};

template <int S>
__global__ __launch_bounds__(512, 2) void conv3x3_wgrad_kernel(const WParams P) {
    constexpr int PPMAX = PatchCfg<S>::PPMAX, PSLOTS = PatchCfg<S>::PSLOTS;
    __shared__ __attribute__((aligned(16))) bf16 dyt[2][64 * LD];
    __shared__ __attribute__((aligned(16))) bf16 xpt[2][PPMAX * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = (wave >> 1) & 1, tg = wave >> 2;          // co half, ci half, tap group
    const int W = P.W, R = P.R, WP = (W - 1) * S + 3, PP = ((R - 1) * S + 3) * WP;   // patch: input columns s ox + kw - 1, rows s oy + kh - 1
    // block -> (split, ci tile, co tile)
    const int tiles_ci = P.Ci / 64, tiles_co = P.Co / 64;
    int bid = blockIdx.x;
    const int split = bid / (tiles_ci * tiles_co);
    bid -= split * tiles_ci * tiles_co;
    const int tci = bid / tiles_co, tco = bid - tci * tiles_co;
    const int co0 = tco * 64, ci0 = tci * 64;

    // ---- this thread's staging slots ----
    // dy: pixel tid / 8 of the tile, vector tid % 8 of its 64-channel slice
    const int dpx = tid >> 3, dvec = tid & 7;
    const int dy_rel = dpx * P.Co + co0 + dvec * 8;                          // relative to the tile's first pixel (tiles are whole image rows)
    // patch: slots v = tid + 512 * i over PP * 8 vectors; pixel (pr, pc) of the halo patch, rows pr = 0 / R + 1 and columns pc = 0 / W + 1 are halo
    int p_rel[PSLOTS], p_lds[PSLOTS];
    uint32_t p_colok = 0, p_used = 0, p_top = 0, p_bot = 0;
#pragma unroll
    for (int i = 0; i < PSLOTS; ++i) {
        const int v = tid + 512 * i, px = v >> 3, vec = v & 7;
        const bool used = px < PP;
        const int pr = used ? px / WP : 0, pc = used ? px - pr * WP : 0;
        p_rel[i] = ((pr - 1) * P.IW + (pc - 1)) * P.Ci + ci0 + vec * 8;        // relative to input pixel (s oy0, 0)
        p_lds[i] = (used ? px : 0) * LD + vec * 8;
        p_used |= (uint32_t)used << i;
        p_colok |= (uint32_t)(pc >= 1 && pc <= P.IW) << i;
        p_top |= (uint32_t)(pr == 0) << i;
        p_bot |= (uint32_t)(pr == (R - 1) * S + 2) << i;
    }
    // two staging register sets: a tile's global loads get TWO compute phases to land (with one set, and one workgroup of 8 waves per CU,
    // the loop ran at the load latency: 2.8 us per 64-pixel tile against 0.3 us of matrix work)
    struct Regs { u32x4 dy; u32x4 xp[PSLOTS]; uint32_t ok; };
    Regs r0, r1;
    auto gload = [&](Regs& r, int t) {   // pixel tile t: image b, first output row oy0
        const int b = t / P.tiles_per_img, oy0 = (t - b * P.tiles_per_img) * R;
        const int64_t pix0 = ((int64_t)b * P.H + oy0) * W;
        r.dy = *reinterpret_cast<const u32x4*>(P.dy + pix0 * P.Co + dy_rel);
        uint32_t ok = p_used & p_colok;
        if (oy0 == 0) ok &= ~p_top;
        if ((oy0 + R - 1) * S + 1 >= P.IH) ok &= ~p_bot;   // the patch's last row lies below the image (stride 1: the tile is the image's last rows)
        const bf16* xb = P.x + (((int64_t)b * P.IH + (int64_t)oy0 * S) * P.IW) * P.Ci;
#pragma unroll
        for (int i = 0; i < PSLOTS; ++i)   // unconditional loads from a clamped address, zero selected when the registers are staged
            r.xp[i] = *reinterpret_cast<const u32x4*>(xb + (((ok >> i) & 1u) ? p_rel[i] : ci0));
        r.ok = ok;
    };
    auto stage = [&](const Regs& r, int buf) {
        *reinterpret_cast<u32x4*>(&dyt[buf][dpx * LD + dvec * 8]) = r.dy;
#pragma unroll
        for (int i = 0; i < PSLOTS; ++i) {
            if ((p_used >> i) & 1u) {
                const bool ok = (r.ok >> i) & 1u;
                u32x4 v = r.xp[i];
                v.x = ok ? v.x : 0u; v.y = ok ? v.y : 0u; v.z = ok ? v.z : 0u; v.w = ok ? v.w : 0u;
                *reinterpret_cast<u32x4*>(&xpt[buf][p_lds[i]]) = v;
            }
        }
    };

    // ---- fragment geometry (gemm_impl.h: transpose reads of a [k][rows] tile) ----
    const int lh = lane >> 5, trL = lane & 15, tr_rowblk = ((lane >> 4) & 1) * 16, tr_k = lh * 8 + (trL >> 2), tr_c = (trL & 3) * 4;
    int a_off[4][2], b_off[4][2];   // element offsets of this lane's k-rows (k-slice ks, low / high half) in the dy tile / in the patch at tap (0, 0)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int kr = ks * 16 + tr_k + 4 * h;
            a_off[ks][h] = kr * LD + wm * 32 + tr_rowblk + tr_c;
            b_off[ks][h] = ((kr / W) * S * WP + (kr % W) * S) * LD + wn * 32 + tr_rowblk + tr_c;
        }
    const int tap0 = tg ? 5 : 0, ntap = tg ? 4 : 5;
    f32x16 acc[5];
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    auto compute = [&](int buf) {
        const bf16* A = dyt[buf];
        const bf16* Bp = xpt[buf];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x4 alo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, A + a_off[ks][0]));
            const bf16x4 ahi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, A + a_off[ks][1]));
            const bf16x8 a = __builtin_shufflevector(alo, ahi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
            for (int t = 0; t < 5; ++t) {
                if (t < ntap) {   // (wave-uniform)
                    const int tap = tap0 + t, shift = ((tap / 3) * WP + tap % 3) * LD;
                    const bf16x4 blo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, Bp + b_off[ks][0] + shift));
                    const bf16x4 bhi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, Bp + b_off[ks][1] + shift));
                    const bf16x8 b = __builtin_shufflevector(blo, bhi, 0, 1, 2, 3, 4, 5, 6, 7);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc[t], 0, 0, 0);
                }
            }
        }
    };

    // ---- pixel tiles split, split + nsplit, ...: two LDS buffers, two register sets (prefetch distance 2), branch-free steady state ----
    const int n = split < P.ntiles ? (P.ntiles - split + P.nsplit - 1) / P.nsplit : 0;   // tiles of this split
    auto tile_of = [&](int i) { return split + (i < n ? i : n - 1) * P.nsplit; };         // (beyond the end: the last tile again, loaded and dropped)
    if (n > 0) {
        gload(r0, tile_of(0));
        gload(r1, tile_of(1));
        stage(r0, 0);
        __syncthreads();
        int i = 0;
        for (; i + 2 < n; i += 2) {   // buffer 0 = tile i; set 1 = tile i + 1 (on its way)
            gload(r0, tile_of(i + 2));
            compute(0);
            stage(r1, 1);
            __syncthreads();
            gload(r1, tile_of(i + 3));
            compute(1);
            stage(r0, 0);
            __syncthreads();
        }
        compute(0);
        if (n - i == 2) {
            stage(r1, 1);
            __syncthreads();
            compute(1);
        }
    }
    // ---- this split's block -> partial[split][co][tap][ci] (a lane holds 4 consecutive ci per register group) ----
    const int co = co0 + wm * 32 + (lane & 31);
    float* out = P.partial + (((int64_t)split * P.Co + co) * 9) * P.Ci + ci0 + wn * 32 + 4 * lh;
#pragma unroll
    for (int t2 = 0; t2 < 5; ++t2) {
        if (t2 < ntap) {
            float* o = out + (int64_t)(tap0 + t2) * P.Ci;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(o + 8 * g) = make_float4(acc[t2][4 * g], acc[t2][4 * g + 1], acc[t2][4 * g + 2], acc[t2][4 * g + 3]);
        }
    }
}

// dW[co][ci][tap] (OIHW, fp32) = sum over the splits (in order: deterministic) of partial[s][co][tap][ci]; a thread per (co, tap, ci)
__global__ __launch_bounds__(256) void conv3x3_wgrad_reduce_kernel(const float* __restrict__ partial, int nsplit, int Co, int Ci, float* __restrict__ dW, int accumulate) {
    const int64_t per = (int64_t)Co * 9 * Ci;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;   // index into [co][tap][ci]
    if (e >= per) return;
    const int ci = (int)(e % Ci);
    const int64_t t2 = e / Ci;
    const int tap = (int)(t2 % 9), co = (int)(t2 / 9);
    float s = 0.f;
    int sp = 0;
    for (; sp + 8 <= nsplit; sp += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = partial[(int64_t)(sp + u) * per + e];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; sp < nsplit; ++sp) s += partial[(int64_t)sp * per + e];
    float* o = dW + ((int64_t)co * Ci + ci) * 9 + tap;
    *o = accumulate ? *o + s : s;
}
}  // namespace

extern "C" size_t ralf_conv3x3_wgrad_workspace_bytes(int B, int H, int W, int Ci, int Co) {
    if (B <= 0 || H <= 0 || W <= 0 || Ci <= 0 || Co <= 0 || W > 64 || 64 % W) return 0;
    const int R = 64 / W;
    if (H % R) return 0;
    const int ntiles = B * (H / R), blocks = (Co / 64) * (Ci / 64);
    int nsplit = 256 / (blocks > 0 ? blocks : 1);
    nsplit = nsplit < 1 ? 1 : (nsplit > ntiles ? ntiles : nsplit);
    return (size_t)nsplit * Co * 9 * Ci * sizeof(float);
}

/* H, W: the OUTPUT grid (dy [B,H,W,Co]); x is [B,IH,IW,Ci] with IH = H, IW = W at stride 1 and (IH + 1) / 2 == H, (IW + 1) / 2 == W at stride 2 */
extern "C" int ralf_conv3x3_wgrad(const void* dy, const void* x, float* dW, int B, int H, int W, int IH, int IW, int stride, int Ci, int Co, int accumulate,
                                  void* workspace, size_t workspace_bytes, void* stream) {
    RALF_REQUIRE(dy && x && dW && B > 0 && H > 0, "conv3x3_wgrad: bad arguments");
    RALF_REQUIRE(stride == 1 || stride == 2, "conv3x3_wgrad: stride 1 or 2");
    RALF_REQUIRE(W >= 8 && W <= 64 && (64 % W) == 0 && H % (64 / W) == 0, "conv3x3_wgrad: output width must be 8, 16, 32 or 64 and the output height a multiple of 64 / W");
    RALF_REQUIRE(stride == 1 ? (IH == H && IW == W) : ((IH + 1) / 2 == H && (IW + 1) / 2 == W && W <= 32), "conv3x3_wgrad: input grid does not match (pad 1; stride 2: output width <= 32)");
    RALF_REQUIRE(Ci % 64 == 0 && Co % 64 == 0, "conv3x3_wgrad: channel counts must be multiples of 64 (Ci=%d Co=%d)", Ci, Co);
    RALF_REQUIRE((((uintptr_t)dy | (uintptr_t)x) & 15) == 0 && ((uintptr_t)dW & 3) == 0, "conv3x3_wgrad: operands must be 16-byte aligned");
    RALF_REQUIRE((int64_t)B * IH * IW * (int64_t)Ci < (1ll << 31) && (int64_t)B * H * W * (int64_t)Co < (1ll << 31), "conv3x3_wgrad: tensor too large for 32-bit offsets");
    const size_t need = ralf_conv3x3_wgrad_workspace_bytes(B, H, W, Ci, Co);
    if (!workspace || workspace_bytes < need) { ralf::set_error("conv3x3_wgrad: workspace %zu < required %zu bytes", workspace_bytes, need); return RALF_ERR_WORKSPACE; }
    WParams P;
    P.dy = (const bf16*)dy; P.x = (const bf16*)x; P.partial = (float*)workspace;
    P.B = B; P.H = H; P.W = W; P.Co = Co; P.Ci = Ci; P.R = 64 / W; P.IH = IH; P.IW = IW;
    P.tiles_per_img = H / P.R; P.ntiles = B * P.tiles_per_img;
    const int blocks = (Co / 64) * (Ci / 64);
    P.nsplit = (int)(need / ((size_t)Co * 9 * Ci * sizeof(float)));
    RALF_REQUIRE(((P.R - 1) * stride + 3) * ((W - 1) * stride + 3) <= (stride == 1 ? PatchCfg<1>::PPMAX : PatchCfg<2>::PPMAX), "conv3x3_wgrad: halo patch too large");
    hipStream_t st = (hipStream_t)stream;
    if (stride == 1) hipLaunchKernelGGL(conv3x3_wgrad_kernel<1>, dim3(blocks * P.nsplit), dim3(512), 0, st, P);
    else hipLaunchKernelGGL(conv3x3_wgrad_kernel<2>, dim3(blocks * P.nsplit), dim3(512), 0, st, P);
    hipLaunchKernelGGL(conv3x3_wgrad_reduce_kernel, dim3((unsigned)(((int64_t)Co * 9 * Ci + 255) / 256)), dim3(256), 0, st, (const float*)workspace, P.nsplit, Co, Ci, dW, accumulate);
    return ralf::check_launch("conv3x3_wgrad");
}
