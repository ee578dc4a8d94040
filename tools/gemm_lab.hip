// Within-process A/B of GEMM main-loop variants on the model's own shapes (bf16, NT, interior fast path):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_lab.hip ralf_amd/csrc/error.cpp -o tools/_gemm_lab.bin
//   tools/_gemm_lab.bin [rounds]
// Variants: register-staged ring (GATHER 3: the shipped kernel) vs direct-to-LDS rings of 2 / 3 stages (GATHER 5 / 6), 128x128 tiles on
// 8 waves and 64x64 tiles on 4 waves.  Every variant's output is compared BIT FOR BIT with the register-staged kernel's (same MFMA
// chain in the same order), then the variants are timed interleaved over `rounds` rounds (median and minimum per shape).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "../ralf_amd/csrc/gemm_impl.h"

#define CK(x)                                                                     \
    do {                                                                          \
        hipError_t e = (x);                                                       \
        if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } \
    } while (0)

struct Shape { int M, N, K; const char* what; };
static const Shape SHAPES[] = {
    {16384, 256, 2304, "layer3 3x3 (as NT)"},    {65536, 128, 1152, "layer2 3x3 (as NT)"}, {262144, 64, 576, "layer1 3x3 (as NT)"},
    {4096, 512, 4608, "layer4 3x3 (as NT)"},     {16384, 1024, 256, "encoder FFN1 / l3 conv3"}, {16384, 256, 1024, "encoder FFN2 / l3 conv1"},
    {34048, 512, 256, "cross-attn K/V"},         {33792, 1024, 256, "head FFN1"},         {262144, 256, 64, "layer1 conv3"},
    {65536, 512, 128, "layer2 conv3"},           {4096, 2048, 512, "layer4 conv3"},       {4096, 512, 2048, "layer4 conv1"},
    {65536, 128, 512, "layer2 conv1"},           {3200, 1024, 256, "decoder FFN1"},       {8192, 8192, 8192, "8192^3"},
    {4096, 4096, 4096, "4096^3"},            {1024, 61548, 1792, "kNN coarse nq=1024"}, {512, 61548, 1792, "kNN coarse nq=512"},
    {256, 61548, 1792, "kNN coarse nq=256"},   {128, 61548, 1792, "kNN coarse nq=128"},
};

typedef int (*LaunchFn)(KParams&, int, hipStream_t);
struct Variant { const char* name; LaunchFn fn; };

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 7;
    const Variant V[] = {
        {"reg 128x128/8w", launch<bf16, true, true, 3, 2, 2, 0, 8>}, {"glds2 128x128/8w", launch<bf16, true, true, 5, 2, 2, 0, 8>},
        {"glds3 128x128/8w", launch<bf16, true, true, 6, 2, 2, 0, 8>},
        {"mt3 128x128/4w", launch<bf16, true, true, 10, 2, 2, 0, 4>}, {"mt4 128x128/4w", launch<bf16, true, true, 12, 2, 2, 0, 4>},
        {"mt2 128x128/4w", launch<bf16, true, true, 13, 2, 2, 0, 4>}, {"mt3 128x128/8w", launch<bf16, true, true, 10, 2, 2, 0, 8>},
        {"mt3 256x128/8w4x2", launch<bf16, true, true, 10, 4, 2, 0, 8, 4>}, {"mt3 256x64/4w4x1", launch<bf16, true, true, 10, 4, 1, 0, 4, 4>},
        {"glds2 256x256/8w", launch<bf16, true, true, 5, 4, 4, 0, 8>}, {"mt2 256x256/8w", launch<bf16, true, true, 13, 4, 4, 0, 8>},
    };
    const int NVALL = sizeof(V) / sizeof(V[0]);
    // LAB_SHAPES / LAB_VARIANTS: comma-separated indices (profiling runs: one shape, a few variants); LAB_ITERS: launches per timed round
    auto pick = [](const char* env, int n) {
        std::vector<int> v;
        const char* e = getenv(env);
        if (!e) { for (int i = 0; i < n; ++i) v.push_back(i); return v; }
        for (const char* p = e; *p;) { v.push_back(atoi(p)); while (*p && *p != ',') ++p; if (*p) ++p; }
        return v;
    };
    const std::vector<int> vsel = pick("LAB_VARIANTS", NVALL), ssel = pick("LAB_SHAPES", (int)(sizeof(SHAPES) / sizeof(SHAPES[0])));
    Variant VS[24];
    int NV = 0;
    for (int i : vsel) VS[NV++] = V[i];
    const int iters_env = getenv("LAB_ITERS") ? atoi(getenv("LAB_ITERS")) : 0;
    size_t maxA = 0, maxB = 0, maxC = 0;
    for (const Shape& s : SHAPES) {
        maxA = std::max(maxA, (size_t)s.M * s.K); maxB = std::max(maxB, (size_t)s.N * s.K); maxC = std::max(maxC, (size_t)s.M * s.N);
    }
    bf16 *A, *B, *C, *Cref;
    CK(hipMalloc(&A, maxA * 2)); CK(hipMalloc(&B, maxB * 2)); CK(hipMalloc(&C, maxC * 2)); CK(hipMalloc(&Cref, maxC * 2));
    {
        std::vector<bf16> h(std::max(maxA, maxB));
        unsigned s = 12345;
        for (auto& x : h) { s = s * 1664525u + 1013904223u; x = (bf16)(((int)(s >> 16) % 2001 - 1000) * 1e-3f); }   // full-range random (the guide: never bench on zeros)
        CK(hipMemcpy(A, h.data(), maxA * 2, hipMemcpyHostToDevice));
        for (auto& x : h) { s = s * 1664525u + 1013904223u; x = (bf16)(((int)(s >> 16) % 2001 - 1000) * 1e-3f); }
        CK(hipMemcpy(B, h.data(), maxB * 2, hipMemcpyHostToDevice));
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("%-22s %18s |", "shape", "M,N,K");
    for (int v = 0; v < NV; ++v) printf(" %18s", VS[v].name);
    printf("   (median us / min us over %d interleaved rounds; * = bitwise mismatch)\n", rounds);
    int bad = 0;
    for (int si : ssel) {
        const Shape& sh = SHAPES[si];
        KParams P;
        memset(&P, 0, sizeof(P));
        RalfGemmDesc& d = P.d;
        d.A = A; d.B = B; d.C = Cref;
        d.M = sh.M; d.N = sh.N; d.K = sh.K; d.nb0 = d.nb1 = 1; d.splitk = 1; d.alpha = 1.f; d.dtype = RALF_BF16;
        d.a_kcontig = 1; d.b_kcontig = 1; d.lda = sh.K; d.ldb = sh.K; d.ldc = sh.N;
        P.fd_hw.set(1); P.fd_rw.set(1); P.fd_sc.set(1); P.fd_kw.set(1); P.fd_st.set(1); P.fd_tap.set(1);
        P.kchunk = sh.K; P.fast = 1; P.vec_epi = 2;
        CK(hipMemset(Cref, 0, (size_t)sh.M * sh.N * 2));
        V[0].fn(P, 1, 0);   // (the reference is always the register-staged kernel)
        CK(hipDeviceSynchronize());
        std::vector<unsigned short> ref((size_t)sh.M * sh.N), got((size_t)sh.M * sh.N);
        CK(hipMemcpy(ref.data(), Cref, ref.size() * 2, hipMemcpyDeviceToHost));
        d.C = C;
        bool mismatch[24] = {false};
        for (int v = 0; v < NV; ++v) {
            CK(hipMemset(C, 0xff, (size_t)sh.M * sh.N * 2));
            VS[v].fn(P, 1, 0);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(got.data(), C, got.size() * 2, hipMemcpyDeviceToHost));
            mismatch[v] = memcmp(ref.data(), got.data(), got.size() * 2) != 0;
            bad += mismatch[v];
        }
        const int iters = iters_env ? iters_env : sh.K >= 4096 ? 3 : 20;
        std::vector<std::vector<float>> t(NV);
        for (int r = 0; r < rounds + 1; ++r) {
            for (int v = 0; v < NV; ++v) {
                CK(hipEventRecord(e0, 0));
                for (int i = 0; i < iters; ++i) VS[v].fn(P, 1, 0);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms = 0;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (r > 0) t[v].push_back(ms * 1e3f / iters);
            }
        }
        char mnk[64];
        snprintf(mnk, sizeof(mnk), "%d,%d,%d", sh.M, sh.N, sh.K);
        printf("%-22s %18s |", sh.what, mnk);
        for (int v = 0; v < NV; ++v) {
            std::sort(t[v].begin(), t[v].end());
            printf(" %8.1f /%7.1f%s", t[v][t[v].size() / 2], t[v][0], mismatch[v] ? "*" : " ");
        }
        const double fl = 2.0 * sh.M * sh.N * sh.K;
        printf("   TF:");
        for (int v = 0; v < NV; ++v) printf(" %4.0f", fl / t[v][t[v].size() / 2] / 1e6);
        printf("\n");
    }
    printf(bad ? "MISMATCHES: %d\n" : "all variants bit-identical to the register-staged kernel\n", bad);
    return bad ? 2 : 0;
}
