// Variants of the 3 x 3 / stride-1 convolution kernels (gemm_impl.h GATHER 1 = tap gather, GATHER 15 = input patch resident in LDS), back-to-back launches:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DRALF_GEMM_PROBE] tools/lab/patch_lab.hip ralf_amd/csrc/error.cpp -o tools/lab/_patch_lab.bin
//   tools/lab/_patch_lab.bin H C [mode 0 | 1] [iters]      (B = 64 images of H x H pixels, C -> C channels)
// Every variant's output is compared bit for bit with the tap gather's.
#include <algorithm>
#include <cstring>
#include <vector>

#include "../../ralf_amd/csrc/gemm_impl.h"

#define CK(x)                                                                     \
    do {                                                                          \
        hipError_t e = (x);                                                       \
        if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } \
    } while (0)

typedef int (*LaunchFn)(KParams&, int, hipStream_t);
struct Variant { const char* name; LaunchFn fn; int bm, fn_; };

int main(int argc, char** argv) {
    const int H = argc > 1 ? atoi(argv[1]) : 16, C = argc > 2 ? atoi(argv[2]) : 256, mode = argc > 3 ? atoi(argv[3]) : 0, iters = argc > 4 ? atoi(argv[4]) : 50;
    const int Bn = 64, M = Bn * H * H, N = C, K = 9 * C;
    bf16 *A, *B, *C0, *C1;
    CK(hipMalloc(&A, (size_t)M * C * 2)); CK(hipMalloc(&B, (size_t)N * K * 2)); CK(hipMalloc(&C0, (size_t)M * N * 2)); CK(hipMalloc(&C1, (size_t)M * N * 2));
    std::vector<bf16> ha((size_t)M * C), hb((size_t)N * K);
    unsigned s = 12345;
    for (auto& x : ha) { s = s * 1664525u + 1013904223u; x = (bf16)(((int)(s >> 16) % 2001 - 1000) * 1e-3f); }
    for (auto& x : hb) { s = s * 1664525u + 1013904223u; x = (bf16)(((int)(s >> 16) % 2001 - 1000) * 1e-3f); }
    CK(hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
    KParams P;
    memset(&P, 0, sizeof(P));
    RalfGemmDesc& d = P.d;
    d.A = A; d.B = B; d.C = C0;
    d.M = M; d.N = N; d.K = K; d.nb0 = d.nb1 = 1; d.splitk = 1; d.alpha = 1.f; d.dtype = RALF_BF16;
    d.a_kcontig = 1; d.b_kcontig = 1; d.lda = K; d.ldb = K; d.ldc = N;
    d.gather = 1;
    RalfConvGeom& g = d.g;
    g.RH = g.RW = g.SH = g.SW = H; g.SC = C; g.KH = g.KW = 3; g.stride = 1; g.pad = 1; g.mode = mode;
    P.fd_hw.set(H * H); P.fd_rw.set(H); P.fd_sc.set(C); P.fd_kw.set(3); P.fd_st.set(1); P.fd_tap.set(1);
    P.ln_sgn = mode ? -1 : 1; P.ln_sh = 0; P.ln_pm = 0; P.img = H * H * C; P.swsc = H * C;
    P.fd_q.set(1); P.fd_hw2.set(1); P.fd_rw2.set(1);
    P.kchunk = K; P.fast = 0; P.tapuni = 1; P.vec_epi = 2;
    P.patch = 1; P.p_pw = H + 2; P.p_str = 2 * C + 16;
    while ((1 << P.p_swsh) < H) ++P.p_swsh;
    while ((8 << P.p_c8sh) < C) ++P.p_c8sh;
    P.fd_pw.set(H + 2);
    const Variant vs[] = {
        {"tap gather (shipped: 128 x 128 on 8 waves, 64 x 64 on 4)", launch_cfg<bf16, true, true, 1>, 128, 0},
        {"patch, 128 x 128, 8 waves of 64 x 32                  ", launch<bf16, true, true, 15, 2, 2, 0, 8>, 128, 2},
        {"patch, 128 x 128, 4 waves of 64 x 64, pipelined        ", launch<bf16, true, true, 15, 2, 2, 0, 4>, 128, 2},
        {"patch, 256 x 128, 8 waves of 64 x 64 (4 x 2), pipelined", launch<bf16, true, true, 15, 4, 2, 0, 8, 4>, 256, 2},
        {"patch, 256 x  64, 8 waves of 64 x 32 (4 x 2), 2 per CU ", launch<bf16, true, true, 15, 4, 1, 0, 8, 4>, 256, 1},
        {"patch, 256 x  64, 4 waves of 64 x 64 (4 x 1), pipelined", launch<bf16, true, true, 15, 4, 1, 0, 4, 4>, 256, 1},
    };
    printf("3 x 3 stride 1, %d -> %d channels at %d x %d, B = 64, mode %d: M = %d N = %d K = %d\n", C, C, H, H, mode, M, N, K);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<unsigned short> h0((size_t)M * N), h1((size_t)M * N);
    const char* only = getenv("LAB_VARIANTS");
    for (size_t vi = 0; vi < sizeof(vs) / sizeof(vs[0]); ++vi) {
        const Variant& v = vs[vi];
        if (vi && only && !strchr(only, '0' + (int)vi)) continue;
        const int rows = v.bm / H;
        if (vi && (v.bm % H || (H * H) % v.bm || N % (64 * v.fn_) || (int64_t)(rows + 2) * (H + 2) * (2 * C + 16) > gemm_patch_bytes(v.fn_))) { printf("  %s  does not fit\n", v.name); continue; }
        d.C = vi ? C1 : C0;
        if (vi) CK(hipMemset(C1, 0xff, (size_t)M * N * 2));
        for (int i = 0; i < 3; ++i) if (v.fn(P, 1, 0)) { printf("launch failed: %s\n", ralf_last_error()); return 1; }
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; ++i) v.fn(P, 1, 0);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / iters;
        bool same = true;
        if (vi) {
            CK(hipMemcpy(h1.data(), C1, h1.size() * 2, hipMemcpyDeviceToHost));
            same = memcmp(h0.data(), h1.data(), h0.size() * 2) == 0;
        } else {
            CK(hipMemcpy(h0.data(), C0, h0.size() * 2, hipMemcpyDeviceToHost));
        }
        printf("  %s %7.1f us  %6.0f TFLOP/s  %s\n", v.name, us, 2.0 * M * N * K / us / 1e6, vi ? (same ? "bit-identical" : "MISMATCH") : "");
#ifdef RALF_GEMM_PROBE
        {
            CK(hipDeviceSynchronize());
            v.fn(P, 1, 0);
            CK(hipDeviceSynchronize());
            const int nblk = std::min(P.nwg, 65536);
            std::vector<unsigned long long> h((size_t)nblk * 8);
            CK(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(ralf_probe_buf), h.size() * 8));
            const char* names[4] = {"entry -> setup", "setup -> operands staged (patch + first k-tiles)", "k-loop", "epilogue"};
            for (int p = 0; p < 4; ++p) {
                std::vector<double> w(nblk);
                for (int b = 0; b < nblk; ++b) w[b] = (double)(h[b * 8 + p + 1] - h[b * 8 + p]);
                std::sort(w.begin(), w.end());
                printf("      %-50s median %8.0f  p90 %8.0f ticks\n", names[p], w[nblk / 2], w[nblk * 9 / 10]);
            }
        }
#endif
    }
    return 0;
}
