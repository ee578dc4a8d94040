// EXPERIMENT (decode): cross-attention of one generated token over the RAW memory rows instead of per-layer K / V caches.
//   scores_h[j] = mem_j . u_h   (u_h = Wk_h^T q_h: the key projection folded into the query; the bias term is constant over j and cancels)
//   r_h = sum_j softmax(scores_h)[j] mem_j   (the value projection Wv_h r_h + bv follows as a matrix-vector product)
// ONE pass over the sample's 276 KB of memory rows (bf16 [M][256]) with an online softmax, on the matrix cores: a workgroup per sample, every wave walks its own
// 16-key tiles (LDS-DMA into a private double buffer, XOR-swizzled 16-byte pieces), scores by v_mfma_f32_16x16x32_bf16 (keys x heads), the weighted sum by
// v_mfma_f32_16x16x16_bf16 with the probabilities straight from the score accumulators (their C layout IS the A layout) and the rows read transposed
// (ds_read_b64_tr_b16); the eight waves' partial (max, sum, r) states meet in LDS.  Against the K pass + softmax + V pass of decode_token.hip (552 KB per layer).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lab/xattn_raw_lab.hip -o tools/lab/_xattn_raw_lab.bin;  tools/lab/_xattn_raw_lab.bin [M=540] [B=256]
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int D = 256, NT = 512, NW = 8, TK = 16;            // model width, threads, waves, keys per tile
#ifndef NWA
#define NWA 8    // waves that walk tiles
#endif
#ifndef NBUF
#define NBUF 2   // tile buffers per walking wave (NBUF - 1 tiles in flight)
#endif
constexpr int TILE_B = TK * D * 2;                             // 8 KiB
struct Lds {
    unsigned char tile[NWA][NBUF][TILE_B];                     // per walking wave: NBUF tiles, 16-byte pieces XOR-swizzled by (row & 15)
    unsigned char u[16 * D * 2];                               // u [16][256] bf16 (rows 8 .. 15 zero), swizzled the same way
    float ml[NW][2][16];                                       // per wave: running max, running sum per head
};

__global__ __launch_bounds__(NT) void xattn_raw_kernel(const bf16* __restrict__ mem, const bf16* __restrict__ u, float* __restrict__ out, int M) {
    __shared__ Lds L;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, b = blockIdx.x;
    const int n = lane & 15, g = lane >> 4;
    // u -> LDS (swizzled): 16 rows x 32 pieces
    for (int v = tid; v < 16 * 32; v += NT) {
        const int row = v >> 5, piece = v & 31;
        bf16x8 val = {};
        if (row < 8) val = *reinterpret_cast<const bf16x8*>(u + ((int64_t)b * 8 + row) * D + piece * 8);
        *reinterpret_cast<bf16x8*>(L.u + row * 512 + ((piece ^ (row & 15)) << 4)) = val;
    }
    const bf16* mb = mem + (int64_t)b * M * D;
    const int ntiles = (M + TK - 1) / TK;
    auto issue = [&](int t, int buf) {   // eight 1-KiB pieces: rows 2 i, 2 i + 1; lane -> (row, slot), the global piece is slot ^ (row & 15)
        unsigned char* dst = L.tile[wv][buf];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = 2 * i + (lane >> 5), slot = lane & 31;
            const int key = min(t * TK + row, M - 1);
            const bf16* src = mb + (int64_t)key * D + ((slot ^ (row & 15)) << 3);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, LDS_PTR(void, dst + i * 1024), 16, 0, 0);
        }
    };
    float m_run = -__builtin_inff(), l_run = 0.f;   // of head n (replicated over g)
    f32x4 r[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) r[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    int t = wv, buf = 0;
    if (wv < NWA) {
#pragma unroll
        for (int q = 0; q < NBUF - 1; ++q)
            if (t + q * NWA < ntiles) issue(t + q * NWA, q);
    }
    __syncthreads();   // (u staged; drains the first tiles as well)
    for (; wv < NWA && t < ntiles; t += NWA, buf = (buf + 1 == NBUF) ? 0 : buf + 1) {
        // tile t + (NBUF - 1) NWA goes into the buffer the previous iteration finished with; then wait until only the younger tiles are in flight
        const int ahead = t + (NBUF - 1) * NWA;
        if (ahead < ntiles) issue(ahead, (buf + NBUF - 1) % NBUF);
        const int inflight = min((ntiles - 1 - t) / NWA, NBUF - 1);   // tiles behind this one that have been requested
        if (inflight >= 7) asm volatile("s_waitcnt vmcnt(56)" ::: "memory");
        else if (inflight == 6) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
        else if (inflight == 5) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
        else if (inflight == 4) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        else if (inflight == 3) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        else if (inflight == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (inflight == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned char* tl = L.tile[wv][buf];
        // ---- scores: keys x heads (two accumulators: a 4-deep instead of an 8-deep dependent chain) ----
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < 8; kc += 2) {
            const int p0 = kc * 4 + g, p1 = p0 + 4;
            const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(tl + n * 512 + ((p0 ^ n) << 4));
            const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(L.u + n * 512 + ((p0 ^ n) << 4));
            const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(tl + n * 512 + ((p1 ^ n) << 4));
            const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(L.u + n * 512 + ((p1 ^ n) << 4));
            s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0, s, 0, 0, 0);
            s2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, s2, 0, 0, 0);
        }
        s += s2;
        // lane (head n, key group g): keys 4 g + i of the tile
        float tmax = -__builtin_inff();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (t * TK + 4 * g + i >= M) s[i] = -__builtin_inff();
            tmax = fmaxf(tmax, s[i]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float m_new = fmaxf(m_run, tmax);
        const float alpha = m_new > -__builtin_inff() ? __expf(m_run - m_new) : 1.f;
        float p[4], ps = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { p[i] = m_new > -__builtin_inff() ? __expf(s[i] - m_new) : 0.f; ps += p[i]; }
        ps += __shfl_xor(ps, 16);
        ps += __shfl_xor(ps, 32);
        l_run = l_run * alpha + ps;
        const bool moved = __any(m_new != m_run);   // (wave-uniform: after the first tiles the running maxima seldom move -- no rescale then)
        m_run = m_new;
        const bf16x4 pa = {(bf16)p[0], (bf16)p[1], (bf16)p[2], (bf16)p[3]};
        if (moved) {
            // the accumulators' rows are heads 4 g + i: their rescale factors live in the lanes of those heads
            float al[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) al[i] = __shfl(alpha, 4 * g + i);
#pragma unroll
            for (int c = 0; c < 16; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i) r[c][i] *= al[i];
        }
        // ---- weighted sum: heads x columns, 16 column tiles ----
        const int trow = 4 * g + (n >> 2);   // the key row this lane points the transposing read at
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int piece = c * 2 + ((n & 3) >> 1), half = n & 1;
            const bf16x4 vb = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, tl + trow * 512 + ((piece ^ trow) << 4) + half * 8));
            r[c] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pa, vb, r[c], 0, 0, 0);
        }
    }
    // ---- the eight waves' states meet: (max, sum) per head, r [8 heads][256] per wave through the (now idle) tile area ----
    __syncthreads();
    float* part = reinterpret_cast<float*>(&L.tile[0][0][0]);   // [NW][8][256] fp32 = 64 KiB
    if (wv < NWA && g == 0) { L.ml[wv][0][n] = m_run; L.ml[wv][1][n] = l_run; }
    if (wv < NWA && g < 2) {
#pragma unroll
        for (int c = 0; c < 16; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) part[(wv * 8 + 4 * g + i) * D + c * 16 + n] = r[c][i];
    }
    __syncthreads();
    for (int e = tid; e < 8 * D; e += NT) {
        const int h = e >> 8;
        float mm = -__builtin_inff();
#pragma unroll
        for (int w = 0; w < NWA; ++w) mm = fmaxf(mm, L.ml[w][0][h]);
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int w = 0; w < NWA; ++w) {
            const float f = L.ml[w][0][h] > -__builtin_inff() ? __expf(L.ml[w][0][h] - mm) : 0.f;
            num += f * part[(w * 8 + h) * D + (e & 255)];
            den += f * L.ml[w][1][h];
        }
        out[(int64_t)b * 8 * D + e] = num / den;
    }
}

// variant 2: the tile's rows come into REGISTERS as the score product's A fragments (lane (key n, piece group g): eight 16-byte loads per tile), TWO tiles in
// flight per wave (128 KB per CU on their way instead of 64: the LDS-DMA form above waits a memory round trip per tile); a tile is copied to the wave's LDS
// buffer only for the transposing reads of the weighted sum
#if NWA == 8
__global__ __launch_bounds__(NT) void xattn_raw_reg_kernel(const bf16* __restrict__ mem, const bf16* __restrict__ u, float* __restrict__ out, int M) {
    __shared__ Lds L;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, b = blockIdx.x;
    const int n = lane & 15, g = lane >> 4;
    for (int v = tid; v < 16 * 32; v += NT) {
        const int row = v >> 5, piece = v & 31;
        bf16x8 val = {};
        if (row < 8) val = *reinterpret_cast<const bf16x8*>(u + ((int64_t)b * 8 + row) * D + piece * 8);
        *reinterpret_cast<bf16x8*>(L.u + row * 512 + ((piece ^ (row & 15)) << 4)) = val;
    }
    const bf16* mb = mem + (int64_t)b * M * D;
    const int ntiles = (M + TK - 1) / TK;
    auto load = [&](bf16x8 (&a)[8], int t) {
        const bf16* src = mb + (int64_t)min(t * TK + n, M - 1) * D + g * 8;
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) a[kc] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(src + kc * 32));
    };
    float m_run = -__builtin_inff(), l_run = 0.f;
    f32x4 r[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) r[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a0[8], a1[8], a2[8];
    unsigned char* tl = L.tile[wv][0];
    auto tile = [&](const bf16x8 (&a)[8], int t) {
        // the rows into the wave's LDS tile (swizzled) for the transposing reads; the wave's own writes: no barrier, the LDS counter orders them
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) *reinterpret_cast<bf16x8*>(tl + n * 512 + (((kc * 4 + g) ^ n) << 4)) = a[kc];
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) {
            const bf16x8 bb = *reinterpret_cast<const bf16x8*>(L.u + n * 512 + (((kc * 4 + g) ^ n) << 4));
            s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kc], bb, s, 0, 0, 0);
        }
        float tmax = -__builtin_inff();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (t * TK + 4 * g + i >= M) s[i] = -__builtin_inff();
            tmax = fmaxf(tmax, s[i]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float m_new = fmaxf(m_run, tmax);
        const float alpha = m_new > -__builtin_inff() ? __expf(m_run - m_new) : 1.f;
        float p[4], ps = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { p[i] = m_new > -__builtin_inff() ? __expf(s[i] - m_new) : 0.f; ps += p[i]; }
        ps += __shfl_xor(ps, 16);
        ps += __shfl_xor(ps, 32);
        l_run = l_run * alpha + ps;
        m_run = m_new;
        const bf16x4 pa = {(bf16)p[0], (bf16)p[1], (bf16)p[2], (bf16)p[3]};
        float al[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) al[i] = __shfl(alpha, 4 * g + i);
        const int trow = 4 * g + (n >> 2);
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int piece = c * 2 + ((n & 3) >> 1), half = n & 1;
            const bf16x4 vb = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, tl + trow * 512 + ((piece ^ trow) << 4) + half * 8));
#pragma unroll
            for (int i = 0; i < 4; ++i) r[c][i] *= al[i];
            r[c] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pa, vb, r[c], 0, 0, 0);
        }
    };
    int t = wv;
    if (t < ntiles) load(a0, t);
    if (t + NW < ntiles) load(a1, t + NW);
    __syncthreads();
    for (; t < ntiles; t += 3 * NW) {   // three register sets in rotation: two tiles on their way while one is multiplied
        if (t + 2 * NW < ntiles) load(a2, t + 2 * NW);
        tile(a0, t);
        if (t + NW >= ntiles) break;
        if (t + 3 * NW < ntiles) load(a0, t + 3 * NW);
        tile(a1, t + NW);
        if (t + 2 * NW >= ntiles) break;
        if (t + 4 * NW < ntiles) load(a1, t + 4 * NW);
        tile(a2, t + 2 * NW);
    }
    __syncthreads();
    float* part = reinterpret_cast<float*>(&L.tile[0][0][0]);
    if (g == 0) { L.ml[wv][0][n] = m_run; L.ml[wv][1][n] = l_run; }
    __syncthreads();   // (every wave is done with its tile buffer before the partial sums overwrite the area)
    if (g < 2) {
#pragma unroll
        for (int c = 0; c < 16; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) part[(wv * 8 + 4 * g + i) * D + c * 16 + n] = r[c][i];
    }
    __syncthreads();
    for (int e = tid; e < 8 * D; e += NT) {
        const int h = e >> 8;
        float mm = -__builtin_inff();
#pragma unroll
        for (int w = 0; w < NW; ++w) mm = fmaxf(mm, L.ml[w][0][h]);
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const float f = L.ml[w][0][h] > -__builtin_inff() ? __expf(L.ml[w][0][h] - mm) : 0.f;
            num += f * part[(w * 8 + h) * D + (e & 255)];
            den += f * L.ml[w][1][h];
        }
        out[(int64_t)b * 8 * D + e] = num / den;
    }
}
#endif

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 540, B = argc > 2 ? atoi(argv[2]) : 256;
    std::vector<bf16> hm((size_t)B * M * D), hu((size_t)B * 8 * D);
    unsigned s = 777;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((int)(s >> 16) % 2001 - 1000) * 1e-3f; };
    for (auto& x : hm) x = (bf16)rnd();
    for (auto& x : hu) x = (bf16)(rnd() * 0.6f);
    bf16 *dm, *du; float* dout;
    CK(hipMalloc(&dm, hm.size() * 2)); CK(hipMalloc(&du, hu.size() * 2)); CK(hipMalloc(&dout, (size_t)B * 8 * D * 4));
    CK(hipMemcpy(dm, hm.data(), hm.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(du, hu.data(), hu.size() * 2, hipMemcpyHostToDevice));
    int rc = 0;
    for (int variant = 0; variant < (NWA == 8 && NBUF == 2 ? 2 : 1); ++variant) {
    auto launch = [&]() {
        if (variant == 0) hipLaunchKernelGGL(xattn_raw_kernel, dim3(B), dim3(NT), 0, 0, dm, du, dout, M);
#if NWA == 8
        else hipLaunchKernelGGL(xattn_raw_reg_kernel, dim3(B), dim3(NT), 0, 0, dm, du, dout, M);
#endif
    };
    CK(hipMemset(dout, 0xff, (size_t)B * 8 * D * 4));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int it = 50;
    hipEventRecord(e0, 0);
    for (int i = 0; i < it; ++i) launch();
    hipEventRecord(e1, 0);
    CK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / it, bytes = (double)B * M * D * 2;
    printf("[%d walking waves x %d buffers] %s  M = %d, B = %d: %.1f us per launch = %.2f TB/s of the memory rows (%.1f MB)\n", NWA, NBUF, variant ? "rows through registers, two tiles in flight" : "rows by LDS-DMA", M, B, us, bytes / us / 1e6, bytes / 1e6);
    std::vector<float> ho((size_t)B * 8 * D);
    CK(hipMemcpy(ho.data(), dout, ho.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int bb : {0, 1, B / 2, B - 1}) {
        for (int h = 0; h < 8; ++h) {
            std::vector<double> sc(M);
            double mx = -1e300;
            for (int j = 0; j < M; ++j) {
                double a = 0;
                for (int c = 0; c < D; ++c) a += (double)(float)hm[((size_t)bb * M + j) * D + c] * (double)(float)hu[((size_t)bb * 8 + h) * D + c];
                sc[j] = a; mx = std::max(mx, a);
            }
            double den = 0;
            for (int j = 0; j < M; ++j) { sc[j] = exp(sc[j] - mx); den += sc[j]; }
            for (int c = 0; c < D; ++c) {
                double a = 0;
                for (int j = 0; j < M; ++j) a += sc[j] * (double)(float)hm[((size_t)bb * M + j) * D + c];
                a /= den;
                worst = std::max(worst, fabs(a - (double)ho[((size_t)bb * 8 + h) * D + c]));
            }
        }
    }
    printf("    worst |r - reference| over 4 samples x 8 heads x 256 columns: %.3g %s\n", worst, worst < 5e-3 ? "(ok)" : "(MISMATCH)");
    if (!(worst < 5e-3)) rc = 2;
    }
    return rc;
}
