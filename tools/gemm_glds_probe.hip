// Experiment (not product code): bf16 NT GEMM, 128x128 tile on 8 waves, operands loaded global -> LDS directly
// (global_load_lds_dwordx4) into an UNPADDED tile with a source-side XOR swizzle, two LDS tiles, counted vmcnt + raw barriers.
// Purpose: what the production kernel's mid-size products could gain from dropping the register staging + ds_write_b128 pass
// (tools/gemm_probe.hip shows them LDS / latency bound).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_glds_probe.hip -o tools/_gemm_glds_probe.bin
//   tools/_gemm_glds_probe.bin M N K        (M, N multiples of 128, K multiple of 64)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <cmath>
#include <vector>

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define LDS_VOID(p) ((__attribute__((address_space(3))) void*)(p))

#define CK(x)                                                                     \
    do {                                                                          \
        hipError_t e = (x);                                                       \
        if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } \
    } while (0)

constexpr int BM = 128, BN = 128, BK = 64, NT = 512;

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

__global__ __launch_bounds__(NT, 4) void gemm_nt_glds(const bf16* __restrict__ A, const bf16* __restrict__ B, bf16* __restrict__ C, int M, int N, int K) {
    __shared__ __attribute__((aligned(1024))) bf16 la[2][BM * BK];
    __shared__ __attribute__((aligned(1024))) bf16 lb[2][BN * BK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 2, wn = wave & 3;
    const int l31 = lane & 31, lh = lane >> 5;
    const int tiles_n = N / BN, nwg = (M / BM) * tiles_n;
    const int vid = xcd_remap(blockIdx.x, nwg);
    const int tm = vid / tiles_n, tn = vid - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    // per-thread source pointers of its two 16-byte chunks per operand and k-tile: LDS slot (row r, chunk c) holds global chunk
    // c ^ ((r >> 1) & 7) of row r; the destination of a wave-instruction is its wave-uniform base + lane * 16
    const bf16* ga[2];
    const bf16* gb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int v = tid + NT * i, r = v >> 3, c = v & 7, q = c ^ ((r >> 1) & 7);
        ga[i] = A + (int64_t)(m0 + r) * K + q * 8;
        gb[i] = B + (int64_t)(n0 + r) * K + q * 8;
    }
    auto load = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_global_load_lds(ga[i], LDS_VOID(&la[buf][(NT * i + 64 * wave) * 8]), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(gb[i], LDS_VOID(&lb[buf][(NT * i + 64 * wave) * 8]), 16, 0, 0);
            ga[i] += BK; gb[i] += BK;
        }
    };
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int swz = (l31 >> 1) & 7;
    auto compute = [&](int buf) {
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const int q = ((ks * 2 + lh) ^ swz) * 8;
            const bf16x8 b = *reinterpret_cast<const bf16x8*>(&lb[buf][(wn * 32 + l31) * BK + q]);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(&la[buf][(wm * 64 + i * 32 + l31) * BK + q]);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc[i], 0, 0, 0);
            }
        }
    };
    const int nt = K / BK;
    load(0);
    for (int t = 0; t + 1 < nt; ++t) {
        load((t + 1) & 1);                                   // tile t+1 on its way while tile t is consumed
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     // this thread's 4 loads of tile t have landed
        __builtin_amdgcn_s_barrier();                        // ... and everybody else's
        compute(t & 1);
        __builtin_amdgcn_s_barrier();                        // tile t may be overwritten (by the load of tile t+2)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    compute((nt - 1) & 1);
    // accumulator register r of a lane: n = (r & 3) + 8 * (r >> 2) + 4 * lh, m = l31
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + wm * 64 + i * 32 + l31;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = n0 + wn * 32 + 8 * g + 4 * lh;
            bf16x4 o;
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = (bf16)acc[i][4 * g + q];
            *reinterpret_cast<bf16x4*>(C + (int64_t)m * N + n) = o;
        }
    }
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 16384, N = argc > 2 ? atoi(argv[2]) : 1024, K = argc > 3 ? atoi(argv[3]) : 256;
    if (M % BM || N % BN || K % BK) { printf("M, N multiples of 128 and K of 64\n"); return 1; }
    bf16 *A, *B, *C;
    CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&B, (size_t)N * K * 2)); CK(hipMalloc(&C, (size_t)M * N * 2));
    std::vector<bf16> ha((size_t)M * K), hb((size_t)N * K);
    unsigned s = 12345;
    for (auto& x : ha) { s = s * 1664525u + 1013904223u; x = (bf16)(((int)(s >> 16) % 2001 - 1000) * 1e-3f); }
    for (auto& x : hb) { s = s * 1664525u + 1013904223u; x = (bf16)(((int)(s >> 16) % 2001 - 1000) * 1e-3f); }
    CK(hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
    const int grid = (M / BM) * (N / BN);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gemm_nt_glds, dim3(grid), dim3(NT), 0, 0, A, B, C, M, N, K);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 50;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(gemm_nt_glds, dim3(grid), dim3(NT), 0, 0, A, B, C, M, N, K);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<bf16> hc((size_t)M * N);
    CK(hipMemcpy(hc.data(), C, hc.size() * 2, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int t = 0; t < 4096; ++t) {
        s = s * 1664525u + 1013904223u; const int m = (int)((s >> 8) % (unsigned)M);
        s = s * 1664525u + 1013904223u; const int n = (int)((s >> 8) % (unsigned)N);
        double ref = 0;
        for (int k = 0; k < K; ++k) ref += (double)(float)ha[(size_t)m * K + k] * (double)(float)hb[(size_t)n * K + k];
        worst = std::max(worst, fabs((double)(float)hc[(size_t)m * N + n] - ref) / (fabs(ref) + 1.0));
    }
    printf("glds NT M=%d N=%d K=%d: %.1f us = %.0f TFLOP/s; spot check worst rel err %.3g %s\n", M, N, K, ms * 1e3 / iters,
           2.0 * M * N * K / (ms * 1e-3 / iters) / 1e12, worst, worst < 1e-2 ? "(ok)" : "(MISMATCH)");
    return worst < 1e-2 ? 0 : 2;
}
