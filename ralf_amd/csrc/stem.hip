// The ResNet stem convolution -- 7x7, stride 2, pad 3, 4 input channels stored as 8 (16-byte pixels), 64 output channels -- in DIRECT form for
// gfx950 (timm resnet50 conv1 with the 4th saliency channel, image2layout/train/models/common/image.py:39-48,70-77).
//
// The implicit-GEMM form (gemm_impl.h, the general per-vector gather) spends 147 us on it at B = 64, 256 x 256: a 16-byte vector per tap and pixel,
// each with its own address arithmetic, for a product whose real contraction is 196 deep.  Here:
//   * a workgroup (8 waves) owns ONE OUTPUT ROW segment of <= 128 pixels x 64 channels per tile and walks tiles blockIdx.x, + gridDim.x, ...;
//   * the 7 x (2 * 128 + 5) input pixels the tile needs are staged once in LDS (30 KB); for a fixed kernel row kh the contraction runs over
//     (kw, c) = the 7 x 8 values that lie CONTIGUOUSLY behind input pixel 2 ox - 3 of row 2 oy - 3 + kh, so the pixel-side MFMA operand of k-step
//     (kh, kw pair) is one 16-byte LDS read per lane (conflict-free: the 64 lanes cover a dense 1 KiB), 28 k-steps of v_mfma_f32_32x32x16_bf16;
//   * the 64 x 448 weights (kw padded to 8, zero tap) live in REGISTERS for the whole kernel (28 fragments per lane, loaded once per workgroup);
//   * the tile leaves through LDS in 16-byte row-contiguous stores, with its per-channel sum / sum of squares (of the stored bf16 values) as one
//     row of BatchNorm partial statistics (the layout ralf_bn_stats_from_partials takes).
#include "common.h"

namespace {
typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int TPX = 128;                 // output pixels per tile (one output row segment)
constexpr int PW = 2 * TPX + 6;          // patch columns: input pixels 2 ox - 3 ... 2 ox + 4 for ox in [0, 128) (column 2 ox + 4 = the zero tap kw = 7) -> 262
constexpr int PROWS = 7;
constexpr int PVEC = PROWS * PW;         // 16-byte pixels per patch
constexpr int PSLOTS = (PVEC + 511) / 512;
constexpr int OLD = 64 + 8;              // output staging row stride (elements)

struct SParams {
    const bf16* x; const bf16* w; bf16* y; float* part;
    int B, IH, IW, OH, OW, segs, ntiles;   // segs = ceil(OW / 128) row segments; ntiles = B * OH * segs
};

__global__ __launch_bounds__(512, 2) void stem7x7_fwd_kernel(const SParams P) {
    // the patch keeps the FOUR real channels of a pixel (8 bytes; the image is stored with 8, the upper four zero): a 16-byte operand then spans two pixels, a
    // k-step of the matrix core four taps x four channels, and a kernel row takes 2 k-steps (taps 0 .. 7, the eighth a zero tap) instead of 4 -- 14 MFMAs per
    // 32 x 32 output tile instead of 28, half of which multiplied zeros
    __shared__ __attribute__((aligned(16))) bf16 patch[PVEC * 4 + 16];   // (+ slack: the zero tap of the last pixel reads one pixel past the row)
    __shared__ __attribute__((aligned(16))) bf16 ost[TPX * OLD];
    __shared__ float red[2][8][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pxf = wave & 3, cof = wave >> 2, l31 = lane & 31, lh = lane >> 5;

    // ---- weights -> registers: fragment (kh, j): row co = 32 cof + l31, taps kw = 4 j + 2 lh, + 1 with their 4 real channels; kw = 7 is the zero tap ----
    bf16x8 wf[PROWS][2];
    {
        const bf16* wr = P.w + (int64_t)(cof * 32 + l31) * (7 * 7 * 8);
        typedef __bf16 bf16x4w __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int kh = 0; kh < PROWS; ++kh)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int kw = 4 * j + 2 * lh;
                const bf16x4w lo = *reinterpret_cast<const bf16x4w*>(wr + (kh * 7 + kw) * 8);
                bf16x4w hi = *reinterpret_cast<const bf16x4w*>(wr + (kh * 7 + (kw + 1 < 7 ? kw + 1 : 6)) * 8);
                if (kw + 1 >= 7) hi = bf16x4w{(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
                wf[kh][j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
    }
    // ---- this thread's patch slots: vector v = tid + 512 i -> (row kh, column pc); input pixel (2 oy - 3 + kh, 2 ox0 - 3 + pc) ----
    int s_kh[PSLOTS], s_pc[PSLOTS];
    uint32_t s_used = 0;
#pragma unroll
    for (int i = 0; i < PSLOTS; ++i) {
        const int v = tid + 512 * i;
        const bool used = v < PVEC;
        s_kh[i] = used ? v / PW : 0;
        s_pc[i] = used ? v - s_kh[i] * PW : 0;
        s_used |= (uint32_t)used << i;
    }
    u32x4 rp[PSLOTS];
    uint32_t rok = 0;
    auto gload = [&](int t) {
        const int seg = t % P.segs, row = t / P.segs, oy = row % P.OH, b = row / P.OH;
        const int iy0 = 2 * oy - 3, ix0 = 2 * seg * TPX - 3;
        uint32_t ok = 0;
#pragma unroll
        for (int i = 0; i < PSLOTS; ++i) {
            const int iy = iy0 + s_kh[i], ix = ix0 + s_pc[i];
            const bool in = ((s_used >> i) & 1u) && (unsigned)iy < (unsigned)P.IH && (unsigned)ix < (unsigned)P.IW;
            const int64_t off = (((int64_t)b * P.IH + (in ? iy : 0)) * P.IW + (in ? ix : 0)) * 8;   // (clamped address, zero selected at stage time)
            rp[i] = *reinterpret_cast<const u32x4*>(P.x + off);
            ok |= (uint32_t)in << i;
        }
        rok = ok;
    };
    auto stage = [&]() {
#pragma unroll
        for (int i = 0; i < PSLOTS; ++i) {
            if ((s_used >> i) & 1u) {
                const bool ok = (rok >> i) & 1u;
                const u32x4 v = rp[i];
                uint2 lo;
                lo.x = ok ? v.x : 0u; lo.y = ok ? v.y : 0u;
                *reinterpret_cast<uint2*>(&patch[(s_kh[i] * PW + s_pc[i]) * 4]) = lo;
            }
        }
    };

    int t = blockIdx.x;
    if (t < P.ntiles) gload(t);
    if (tid < 4) *reinterpret_cast<uint2*>(&patch[PVEC * 4 + tid * 4]) = uint2{0u, 0u};   // (the slack behind the last row)
    for (; t < P.ntiles; t += gridDim.x) {
        __syncthreads();            // the previous tile's reads of patch / ost are done
        stage();
        __syncthreads();
        const int tn = t + gridDim.x;
        if (tn < P.ntiles) gload(tn);
        // ---- 14 k-steps: pixel operand = the 16 bytes (two pixels x four channels) behind patch pixel (kh, 2 px + 4 j + 2 lh) ----
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const bf16* ap = patch + (2 * (pxf * 32 + l31) + 2 * lh) * 4;
#pragma unroll
        for (int kh = 0; kh < PROWS; ++kh)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(ap + (kh * PW + 4 * j) * 4);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[kh][j], a, acc, 0, 0, 0);
            }
        // ---- tile -> LDS [128 px][64 co] (a lane holds pixel 32 pxf + l31, channels 32 cof + 8 g + 4 lh + 0..3) ----
        {
            bf16* o = ost + (pxf * 32 + l31) * OLD + cof * 32 + 4 * lh;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                bf16x4 v;
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = (bf16)acc[4 * g + q];
                *reinterpret_cast<bf16x4*>(o + 8 * g) = v;
            }
        }
        __syncthreads();
        const int seg = t % P.segs, row = t / P.segs;
        const int npx = min(TPX, P.OW - seg * TPX);
        // 16-byte row-contiguous stores: thread -> (pixel tid / 8 [+ 64], vector tid % 8)
        bf16* yrow = P.y + ((int64_t)row * P.OW + (int64_t)seg * TPX) * 64;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int px = (tid >> 3) + 64 * h, vec = tid & 7;
            if (px < npx) *reinterpret_cast<u32x4*>(yrow + (int64_t)px * 64 + vec * 8) = *reinterpret_cast<const u32x4*>(ost + px * OLD + vec * 8);
        }
        if (P.part) {   // BatchNorm partial statistics of the values as stored: channel tid % 64, pixels tid / 64 + 8 k
            const int c = tid & 63, pg = tid >> 6;
            float s1 = 0.f, s2 = 0.f;
            for (int px = pg; px < npx; px += 8) {
                const float v = (float)ost[px * OLD + c];
                s1 += v; s2 += v * v;
            }
            red[0][pg][c] = s1; red[1][pg][c] = s2;
            __syncthreads();
            if (tid < 128) {
                const int which = tid >> 6, cc = tid & 63;
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) s += red[which][k][cc];
                P.part[((int64_t)t * 2 + which) * 64 + cc] = s;
            }
        }
    }
}

// ---- weight gradient of the stem:  dW[co][kh][kw][c] = sum over (b, oy, ox) of dy[b, oy, ox, co] * x[b, 2 oy - 3 + kh, 2 ox - 3 + kw, c] ----
// Same tiles and the same patch: per output-row tile the 128 x 64 slice of dy and the 7-row input patch are staged once; for a fixed kernel row kh
// the (kw, c) axis of the weight block is the 64 contiguous values behind patch pixel (kh, 2 ox), so the x operand of the contraction over pixels is
// a [k = pixel][n = 64] tile whose k-rows lie 32 bytes apart in the patch -- read through ds_read_b64_tr_b16 like the dy tile.  A workgroup (8 waves:
// co half x kernel-row group {0,1} {2,3} {4,5} {6}) keeps the whole 64 x 7 x 32 block in registers over its tiles and writes it once; a second
// kernel sums the workgroups' blocks in order into the OIHW fp32 gradient (4 real channels, 7 real taps).
constexpr int DLD = 64 + 8;      // dy tile row stride (elements)
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

struct SWParams {
    const bf16* x; const bf16* dy; float* partial;
    int B, IH, IW, OH, OW, segs, ntiles;
};

__global__ __launch_bounds__(512, 4) void stem7x7_wgrad_kernel(const SWParams P) {   // (two workgroups per CU: with the 8-channel patch's 64 accumulators that spilled and ran 199 us instead of 137)
    // (the patch keeps the FOUR real channels of a pixel, as in the forward: the (kw, c) axis of a kernel row is 8 taps x 4 channels = 32 values, ONE 32-wide
    //  fragment per kernel row instead of two, half of which were the zero channels)
    __shared__ __attribute__((aligned(16))) bf16 patch[PVEC * 4 + 64];   // (+ slack: the zero-tap columns of the last pixels read just past the last row)
    __shared__ __attribute__((aligned(16))) bf16 dyt[TPX * DLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cof = wave & 1, khg = wave >> 1;                 // co half; kernel rows 2 khg, 2 khg + 1 (khg = 3: row 6 only)
    const int nkh = khg == 3 ? 1 : 2;
    int s_kh[PSLOTS], s_pc[PSLOTS];
    uint32_t s_used = 0;
#pragma unroll
    for (int i = 0; i < PSLOTS; ++i) {
        const int v = tid + 512 * i;
        const bool used = v < PVEC;
        s_kh[i] = used ? v / PW : 0;
        s_pc[i] = used ? v - s_kh[i] * PW : 0;
        s_used |= (uint32_t)used << i;
    }
    u32x4 rp[PSLOTS], rd[2];
    uint32_t rok = 0, dok = 0;
    auto gload = [&](int t) {
        const int seg = t % P.segs, row = t / P.segs, oy = row % P.OH, b = row / P.OH;
        const int iy0 = 2 * oy - 3, ix0 = 2 * seg * TPX - 3;
        uint32_t ok = 0;
#pragma unroll
        for (int i = 0; i < PSLOTS; ++i) {
            const int iy = iy0 + s_kh[i], ix = ix0 + s_pc[i];
            const bool in = ((s_used >> i) & 1u) && (unsigned)iy < (unsigned)P.IH && (unsigned)ix < (unsigned)P.IW;
            rp[i] = *reinterpret_cast<const u32x4*>(P.x + (((int64_t)b * P.IH + (in ? iy : 0)) * P.IW + (in ? ix : 0)) * 8);
            ok |= (uint32_t)in << i;
        }
        rok = ok;
        // dy tile: pixel tid / 8 (+ 64), vector tid % 8; pixels beyond the row's end are zero (they contribute nothing)
        const int npx = min(TPX, P.OW - seg * TPX);
        const bf16* drow = P.dy + ((int64_t)row * P.OW + (int64_t)seg * TPX) * 64;
        uint32_t dk = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int px = (tid >> 3) + 64 * h;
            const bool in = px < npx;
            rd[h] = *reinterpret_cast<const u32x4*>(drow + (int64_t)(in ? px : 0) * 64 + (tid & 7) * 8);
            dk |= (uint32_t)in << h;
        }
        dok = dk;
    };
    auto stage = [&]() {
#pragma unroll
        for (int i = 0; i < PSLOTS; ++i) {
            if ((s_used >> i) & 1u) {
                const bool ok = (rok >> i) & 1u;
                const u32x4 v = rp[i];
                uint2 lo;
                lo.x = ok ? v.x : 0u; lo.y = ok ? v.y : 0u;
                *reinterpret_cast<uint2*>(&patch[(s_kh[i] * PW + s_pc[i]) * 4]) = lo;
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const bool ok = (dok >> h) & 1u;
            u32x4 v = rd[h];
            v.x = ok ? v.x : 0u; v.y = ok ? v.y : 0u; v.z = ok ? v.z : 0u; v.w = ok ? v.w : 0u;
            *reinterpret_cast<u32x4*>(&dyt[((tid >> 3) + 64 * h) * DLD + (tid & 7) * 8]) = v;
        }
    };
    if (tid < 64) patch[PVEC * 4 + tid] = (bf16)0.f;   // the slack behind the patch

    // transpose-read geometry (gemm_impl.h): a 16-lane group reads [4 k-rows][16 rows]
    const int lh = lane >> 5, trL = lane & 15, tr_rowblk = ((lane >> 4) & 1) * 16, tr_k = lh * 8 + (trL >> 2), tr_c = (trL & 3) * 4;
    f32x16 acc[2];   // [kernel row of the group]: co x the 32 (kw, c) values of the row
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

    int t = blockIdx.x;
    if (t < P.ntiles) gload(t);
    for (; t < P.ntiles; t += gridDim.x) {
        __syncthreads();
        stage();
        __syncthreads();
        const int tn = t + gridDim.x;
        if (tn < P.ntiles) gload(tn);
#pragma unroll
        for (int ks = 0; ks < TPX / 16; ++ks) {   // 16 pixels per k-step
            const int kr = ks * 16 + tr_k;         // this lane's pixel (low half; high half + 4)
            const bf16* qa = dyt + kr * DLD + cof * 32 + tr_rowblk + tr_c;
            const bf16x4 alo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, qa));
            const bf16x4 ahi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, qa + 4 * DLD));
            const bf16x8 a = __builtin_shufflevector(alo, ahi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                if (kk < nkh) {   // (wave-uniform)
                    const int kh = 2 * khg + kk;
                    const bf16* qb = patch + (kh * PW + 2 * kr) * 4 + tr_rowblk + tr_c;   // k-row stride = 2 pixels = 8 elements
                    const bf16x4 blo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, qb));
                    const bf16x4 bhi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, qb + 4 * 8));
                    const bf16x8 b = __builtin_shufflevector(blo, bhi, 0, 1, 2, 3, 4, 5, 6, 7);
                    acc[kk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc[kk], 0, 0, 0);
                }
            }
        }
    }
    // ---- this workgroup's block -> partial[blockIdx.x][co][kh][(kw, c) 0..31] ----
    const int co = cof * 32 + (lane & 31);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        if (kk < nkh) {
            float* o = P.partial + (((int64_t)blockIdx.x * 64 + co) * 7 + (2 * khg + kk)) * 32 + 4 * lh;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(o + 8 * g) = make_float4(acc[kk][4 * g], acc[kk][4 * g + 1], acc[kk][4 * g + 2], acc[kk][4 * g + 3]);
        }
    }
}

// dW[co][c][kh][kw] (OIHW fp32, c < 4, kw < 7) = sum over the workgroups' blocks in a FIXED order: 64 elements per workgroup, four slices of the blocks
// (b = slice, slice + 4, ...) summed by four waves and combined as (s0 + s1) + (s2 + s3).  (One thread per element over all 256 blocks was a chain of 256
// dependent 114-KB-strided loads on 112 workgroups: 65 us at the very end of the backward, in front of the optimizer.)
__global__ __launch_bounds__(256) void stem7x7_wgrad_reduce_kernel(const float* __restrict__ partial, int nblk, float* __restrict__ dW, int accumulate) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;   // index into [co][kh][kw 0..7][c 0..3]  (64 * 7 * 32 elements: a multiple of 64)
    float s = 0.f;
#pragma unroll 8
    for (int b = slice; b < nblk; b += 4) s += partial[(int64_t)b * (64 * 7 * 32) + e];
    red[slice][lane] = s;
    __syncthreads();
    if (slice == 0) {
        s = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
        const int c = e & 3, kw = (e >> 2) & 7, kh = (e >> 5) % 7, co = (e >> 5) / 7;
        if (kw < 7) {
            float* o = dW + ((co * 4 + c) * 7 + kh) * 7 + kw;
            *o = accumulate ? *o + s : s;
        }
    }
}
}  // namespace

constexpr int SW_GRID = 512;   // persistent workgroups, two per CU (92 registers since the accumulators halved: 81 -> 68 us with the reduction)

extern "C" size_t ralf_stem7x7_wgrad_workspace_bytes(int B, int IH, int IW) {
    if (B <= 0 || IH < 7 || IW < 7) return 0;
    const int OH = (IH - 1) / 2 + 1, OW = (IW - 1) / 2 + 1;
    const int64_t nt = (int64_t)B * OH * ((OW + TPX - 1) / TPX);
    return (size_t)(nt < SW_GRID ? nt : SW_GRID) * 64 * 7 * 32 * sizeof(float);
}

/* x [B, IH, IW, 8] bf16, dy [B, OH, OW, 64] bf16 -> dW [64][4][7][7] fp32 (OIHW; = or +=) */
extern "C" int ralf_stem7x7_wgrad(const void* x, const void* dy, float* dW, int B, int IH, int IW, int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    RALF_REQUIRE(x && dy && dW && B > 0 && IH >= 7 && IW >= 7, "stem7x7_wgrad: bad arguments");
    RALF_REQUIRE((((uintptr_t)x | (uintptr_t)dy) & 15) == 0, "stem7x7_wgrad: operands must be 16-byte aligned");
    const size_t need = ralf_stem7x7_wgrad_workspace_bytes(B, IH, IW);
    if (!workspace || workspace_bytes < need) { ralf::set_error("stem7x7_wgrad: workspace %zu < required %zu bytes", workspace_bytes, need); return RALF_ERR_WORKSPACE; }
    SWParams P;
    P.x = (const bf16*)x; P.dy = (const bf16*)dy; P.partial = (float*)workspace;
    P.B = B; P.IH = IH; P.IW = IW; P.OH = (IH - 1) / 2 + 1; P.OW = (IW - 1) / 2 + 1;
    P.segs = (P.OW + TPX - 1) / TPX;
    const int64_t nt = (int64_t)B * P.OH * P.segs;
    RALF_REQUIRE(nt < (1ll << 30), "stem7x7_wgrad: problem too large");
    P.ntiles = (int)nt;
    const int grid = (int)(nt < SW_GRID ? nt : SW_GRID);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(stem7x7_wgrad_kernel, dim3(grid), dim3(512), 0, st, P);
    hipLaunchKernelGGL(stem7x7_wgrad_reduce_kernel, dim3(64 * 7 * 32 / 64), dim3(256), 0, st, (const float*)workspace, grid, dW, accumulate);
    return ralf::check_launch("stem7x7_wgrad");
}

/* x [B, IH, IW, 8] bf16 (channels 4..7 zero), w [64][7][7][8] bf16 -> y [B, OH, OW, 64] bf16, OH = (IH - 1) / 2 + 1, OW = (IW - 1) / 2 + 1;
 * part (may be NULL): fp32 [B * OH * ceil(OW / 128)][2][64] per-tile channel sums / sums of squares of y */
extern "C" int ralf_stem7x7_fwd(const void* x, const void* w, void* y, float* part, int B, int IH, int IW, void* stream) {
    RALF_REQUIRE(x && w && y && B > 0 && IH >= 7 && IW >= 7, "stem7x7_fwd: bad arguments");
    RALF_REQUIRE((((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) & 15) == 0, "stem7x7_fwd: operands must be 16-byte aligned");
    SParams P;
    P.x = (const bf16*)x; P.w = (const bf16*)w; P.y = (bf16*)y; P.part = part;
    P.B = B; P.IH = IH; P.IW = IW; P.OH = (IH - 1) / 2 + 1; P.OW = (IW - 1) / 2 + 1;
    P.segs = (P.OW + TPX - 1) / TPX;
    const int64_t nt = (int64_t)B * P.OH * P.segs;
    RALF_REQUIRE(nt < (1ll << 30) && (int64_t)B * IH * IW * 8 < (1ll << 40), "stem7x7_fwd: problem too large");
    P.ntiles = (int)nt;
    static const int fwd_grid = [] { const char* e = getenv("RALF_STEM_GRID"); return e ? atoi(e) : 512; }();   // two per CU: one's staging / statistics phases under the other's products (90 -> 80 us; three or four: 85-87)
    const int grid = (int)(nt < fwd_grid ? nt : fwd_grid);   // persistent workgroups (the weights sit in their registers)
    hipLaunchKernelGGL(stem7x7_fwd_kernel, dim3(grid), dim3(512), 0, (hipStream_t)stream, P);
    return ralf::check_launch("stem7x7_fwd");
}
