// Phase timing of ONE ralf_gemm kernel instantiation (bf16, interior fast path) with s_memtime stamps per workgroup:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DRALF_GEMM_PROBE tools/gemm_probe.hip ralf_amd/csrc/error.cpp -o tools/_gemm_probe.bin
//   tools/_gemm_probe.bin M N K [tile: 11 | 22] [layout: nt | nn | tn]
// Prints, per phase, the mean / median cycles over all workgroups, the number of workgroups resident per CU over time,
// and the kernel time measured with HIP events (without the stamps the kernel is ~10 % faster).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <vector>

#include "../ralf_amd/csrc/gemm_impl.h"

#define CK(x)                                                                     \
    do {                                                                          \
        hipError_t e = (x);                                                       \
        if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } \
    } while (0)

template <bool AK, bool BKC, int FM, int FN, int NW = 4, int G = 3>
int run(KParams& P, int iters) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch<bf16, AK, BKC, G, FM, FN, 0, NW>(P, 1, 0);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) launch<bf16, AK, BKC, G, FM, FN, 0, NW>(P, 1, 0);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("kernel (with stamps, back-to-back eager launches): %.1f us  = %.0f TFLOP/s\n", ms * 1e3 / iters,
           2.0 * P.d.M * P.d.N * P.d.K / (ms * 1e-3 / iters) / 1e12);
    // one clean launch for the stamps
    CK(hipDeviceSynchronize());
    launch<bf16, AK, BKC, G, FM, FN, 0, NW>(P, 1, 0);
    CK(hipDeviceSynchronize());
    const int nblk = std::min(P.nwg * P.d.splitk, 65536);
    std::vector<unsigned long long> h((size_t)nblk * 8);
    CK(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(ralf_probe_buf), h.size() * 8));
    const char* names[4] = {"setup (entry -> first load issued)", "first tile (load issued -> staged + barrier)", "k-loop remainder", "epilogue (issue)"};
    for (int p = 0; p < 4; ++p) {
        std::vector<double> v(nblk);
        for (int b = 0; b < nblk; ++b) v[b] = (double)(h[b * 8 + p + 1] - h[b * 8 + p]);
        std::sort(v.begin(), v.end());
        double s = 0;
        for (double x : v) s += x;
        printf("  %-48s mean %8.0f  median %8.0f  p10 %8.0f  p90 %8.0f cycles\n", names[p], s / nblk, v[nblk / 2], v[nblk / 10], v[nblk * 9 / 10]);
    }
    {
        std::vector<double> v(nblk);
        for (int b = 0; b < nblk; ++b) v[b] = (double)(h[b * 8 + 4] - h[b * 8 + 0]);
        std::sort(v.begin(), v.end());
        double s = 0;
        for (double x : v) s += x;
        printf("  %-48s mean %8.0f  median %8.0f cycles\n", "workgroup lifetime (entry -> last store issued)", s / nblk, v[nblk / 2]);
    }
    // placement: workgroups per CU, wall span from the 100 MHz counter
    std::map<unsigned long long, std::vector<std::pair<unsigned long long, unsigned long long>>> cu;
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int b = 0; b < nblk; ++b) {
        const unsigned long long id = h[b * 8 + 7];
        const unsigned hw = (unsigned)id, xcc = (unsigned)(id >> 32) & 0xf;
        const unsigned long long key = ((unsigned long long)xcc << 32) | (hw & 0xffffff00u & ~0xc0u);   // drop wave / simd / pipe bits
        cu[key].push_back({h[b * 8 + 0], h[b * 8 + 4]});
        t0 = std::min(t0, h[b * 8 + 6]); t1 = std::max(t1, h[b * 8 + 6]);
    }
    double conc = 0, span = 0;
    size_t mx = 0, mn = ~0ul;
    for (auto& kv : cu) {
        auto& v = kv.second;
        mx = std::max(mx, v.size()); mn = std::min(mn, v.size());
        unsigned long long a = ~0ull, b = 0, busy = 0;
        for (auto& iv : v) { a = std::min(a, iv.first); b = std::max(b, iv.second); busy += iv.second - iv.first; }
        conc += (double)busy / (double)(b - a);
        span += (double)(b - a);
    }
    printf("  %zu distinct CUs, workgroups per CU %zu..%zu, mean resident workgroups per CU %.2f, mean busy span per CU %.0f cycles; last workgroup entered %.2f us after the first\n",
           cu.size(), mn, mx, conc / cu.size(), span / cu.size(), (t1 - t0) * 0.01);
    return 0;
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 16384, N = argc > 2 ? atoi(argv[2]) : 1024, K = argc > 3 ? atoi(argv[3]) : 256;
    const int tile = argc > 4 ? atoi(argv[4]) : 11;
    const char* lay = argc > 5 ? argv[5] : "nt";
    const bool AK = lay[0] == 'n', BKC = lay[1] == 't';
    bf16 *A, *B, *C;
    CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&B, (size_t)N * K * 2)); CK(hipMalloc(&C, (size_t)M * N * 2));
    std::vector<bf16> ha((size_t)M * K), hb((size_t)N * K);
    unsigned s = 12345;
    for (auto& x : ha) { s = s * 1664525u + 1013904223u; x = (bf16)(((int)(s >> 16) % 2001 - 1000) * 1e-3f); }
    for (auto& x : hb) { s = s * 1664525u + 1013904223u; x = (bf16)(((int)(s >> 16) % 2001 - 1000) * 1e-3f); }
    CK(hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
    KParams P;
    memset(&P, 0, sizeof(P));
    RalfGemmDesc& d = P.d;
    d.A = A; d.B = B; d.C = C;
    d.M = M; d.N = N; d.K = K; d.nb0 = d.nb1 = 1; d.splitk = 1; d.alpha = 1.f; d.dtype = RALF_BF16;
    d.a_kcontig = AK; d.b_kcontig = BKC;
    d.lda = AK ? K : M; d.ldb = BKC ? K : N; d.ldc = N;
    P.fd_hw.set(1); P.fd_rw.set(1); P.fd_sc.set(1); P.fd_kw.set(1); P.fd_st.set(1);
    P.kchunk = K; P.fast = 1; P.vec_epi = 2;
    printf("%s M=%d N=%d K=%d tile %d\n", lay, M, N, K, tile);
    int rc;
    if (tile == 58) rc = run<true, true, 2, 2, 8, 5>(P, 50);          // direct-to-LDS ring, 128x128 on 8 waves (nt)
    else if (tile == 54) rc = run<true, true, 1, 1, 4, 5>(P, 50);     // direct-to-LDS ring, 64x64 on 4 waves (nt)
    else if (tile == 28) rc = (AK && BKC) ? run<true, true, 2, 2, 8>(P, 50) : AK ? run<true, false, 2, 2, 8>(P, 50) : run<false, false, 2, 2, 8>(P, 50);
    else if (tile == 22) rc = (AK && BKC) ? run<true, true, 2, 2>(P, 50) : AK ? run<true, false, 2, 2>(P, 50) : run<false, false, 2, 2>(P, 50);
    else rc = (AK && BKC) ? run<true, true, 1, 1>(P, 50) : AK ? run<true, false, 1, 1>(P, 50) : run<false, false, 1, 1>(P, 50);
    if (rc) return rc;
    // spot check against the host (fp64 accumulate of the bf16 operands), every tile position class
    std::vector<bf16> hc((size_t)M * N);
    CK(hipMemcpy(hc.data(), C, hc.size() * 2, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int t = 0; t < 4096; ++t) {
        s = s * 1664525u + 1013904223u; const int m = (t < 64) ? (M - 1 - t % 8) : (int)((s >> 8) % (unsigned)M);
        s = s * 1664525u + 1013904223u; const int n = (t < 64) ? (N - 1 - t / 8) : (int)((s >> 8) % (unsigned)N);
        double ref = 0;
        for (int k = 0; k < K; ++k) ref += (double)(float)ha[AK ? (size_t)m * K + k : (size_t)k * M + m] * (double)(float)hb[BKC ? (size_t)n * K + k : (size_t)k * N + n];
        const double err = fabs((double)(float)hc[(size_t)m * N + n] - ref) / (fabs(ref) + 1.0);
        worst = std::max(worst, err);
    }
    printf("  spot check of 4096 entries: worst relative error %.3g %s\n", worst, worst < 1e-2 ? "(ok)" : "(MISMATCH)");
    return worst < 1e-2 ? 0 : 2;
}
