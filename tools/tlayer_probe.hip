// Phase timing of the one-launch transformer layer (ralf_amd/csrc/tlayer.hip) with s_memtime stamps per workgroup:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DRALF_TLAYER_PROBE -I ralf_amd/csrc tools/tlayer_probe.hip ralf_amd/csrc/error.cpp -o tools/_tlayer_probe.bin
//   tools/_tlayer_probe.bin [B=64] [S=50] [M (unused)] [decoder=1] [p=0.1]
// Prints the mean stamp-to-stamp cycles over the workgroups and the kernel time from HIP events (back-to-back launches).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../ralf_amd/csrc/tlayer.hip"

#define CK(x)                                                                     \
    do {                                                                          \
        hipError_t e = (x);                                                       \
        if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } \
    } while (0)

static unsigned rs = 12345;
static float frand() { rs = rs * 1664525u + 1013904223u; return ((int)(rs >> 16) % 2001 - 1000) * 1e-3f; }
template <typename T> static T* dev(size_t n, float scale, float off = 0.f) {
    std::vector<T> h(n);
    for (auto& v : h) v = (T)(frand() * scale + off);
    T* p;
    if (hipMalloc(&p, n * sizeof(T)) != hipSuccess) exit(1);
    hipMemcpy(p, h.data(), n * sizeof(T), hipMemcpyHostToDevice);
    return p;
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 64, S = argc > 2 ? atoi(argv[2]) : 50, M = argc > 3 ? atoi(argv[3]) : 532;
    const int cross = argc > 4 ? atoi(argv[4]) : 1;
    const float p = argc > 5 ? atof(argv[5]) : 0.1f;
    RalfTLayerDesc d;
    memset(&d, 0, sizeof(d));
    const size_t R = (size_t)B * S;
    d.x = dev<bf16>(R * 256, 1.f);
    d.ln1_g = dev<float>(256, 0.1f, 1.f); d.ln1_b = dev<float>(256, 0.1f);
    d.ln2_g = dev<float>(256, 0.1f, 1.f); d.ln2_b = dev<float>(256, 0.1f);
    d.ln3_g = dev<float>(256, 0.1f, 1.f); d.ln3_b = dev<float>(256, 0.1f);
    d.w_in = dev<bf16>(768 * 256, 0.06f); d.b_in = dev<float>(768, 0.1f);
    d.w_o = dev<bf16>(256 * 256, 0.06f); d.b_o = dev<float>(256, 0.1f);
    d.w_q = dev<bf16>(256 * 256, 0.06f); d.b_q = dev<float>(256, 0.1f);
    d.w_o2 = dev<bf16>(256 * 256, 0.06f); d.b_o2 = dev<float>(256, 0.1f);
    d.w1 = dev<bf16>(1024 * 256, 0.06f); d.b1 = dev<float>(1024, 0.1f);
    d.w2 = dev<bf16>(256 * 1024, 0.03f); d.b2 = dev<float>(256, 0.1f);
    {   // weights -> fragment order
        RalfPackJob jobs[6] = {{d.w_in, dev<bf16>(768 * 256, 0), 256, 768, 256}, {d.w_o, dev<bf16>(256 * 256, 0), 256, 256, 256}, {d.w_q, dev<bf16>(256 * 256, 0), 256, 256, 256},
                               {d.w_o2, dev<bf16>(256 * 256, 0), 256, 256, 256}, {d.w1, dev<bf16>(1024 * 256, 0), 256, 1024, 256}, {d.w2, dev<bf16>(256 * 1024, 0), 1024, 256, 1024}};
        if (ralf_tlayer_pack(jobs, 6, 0)) { printf("pack failed: %s\n", ralf_last_error()); return 1; }
        d.w_in = jobs[0].dst; d.w_o = jobs[1].dst; d.w_q = jobs[2].dst; d.w_o2 = jobs[3].dst; d.w1 = jobs[4].dst; d.w2 = jobs[5].dst;
    }
    d.o2 = dev<bf16>(R * 256, 1.f);
    d.h1 = dev<bf16>(R * 256, 0); d.h2 = dev<bf16>(R * 256, 0); d.h3 = dev<bf16>(R * 256, 0);
    d.mean1 = dev<float>(R, 0); d.rstd1 = dev<float>(R, 0); d.mean2 = dev<float>(R, 0); d.rstd2 = dev<float>(R, 0); d.mean3 = dev<float>(R, 0); d.rstd3 = dev<float>(R, 0);
    d.qkv = dev<bf16>(R * 768, 0); d.o1 = dev<bf16>(R * 256, 0); d.x1 = dev<bf16>(R * 256, 0); d.q = dev<bf16>(R * 256, 0); d.x2 = dev<bf16>(R * 256, 0);
    d.lse1 = dev<float>(R * 8, 0);
    d.hid = dev<bf16>(R * 1024, 0); d.out = dev<bf16>(R * 256, 0);
    int64_t* seed;
    CK(hipMalloc(&seed, 8));
    const int64_t hs = 777;
    CK(hipMemcpy(seed, &hs, 8, hipMemcpyHostToDevice));
    d.seed = seed;
    d.call_attn1 = 1; d.call_out1 = 2; d.call_out2 = 4; d.call_ffn1 = 5; d.call_ffn2 = 6;
    d.B = B; d.S = S; d.causal = 1; d.scale = 0.17677669529663687f; d.p_attn = p; d.p_res = p; d.eps = 1e-5f;

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[12] = {"", "LayerNorm 1 (+ first weight rows)", "qkv projection", "self-attention (+ qkv store)", "out-projection (+ o1 store)", "x1 + LayerNorm", "q projection + store",
                             "o2 load", "out-projection 2", "x2 + LayerNorm 3", "feed-forward (4 hidden chunks)", "last epilogue + store"};
    for (int part = cross ? 1 : 0; part <= (cross ? 2 : 0); ++part) {
        d.part = part;
        for (int i = 0; i < 3; ++i) if (ralf_tlayer_fwd(&d, 0)) { printf("launch failed: %s\n", ralf_last_error()); return 1; }
        CK(hipDeviceSynchronize());
        const int iters = 20;
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; ++i) ralf_tlayer_fwd(&d, 0);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("part %d  B=%d S=%d p=%.2f: %.1f us per launch (with stamps, back-to-back)\n", part, B, S, p, ms * 1e3 / iters);
        std::vector<unsigned long long> h((size_t)16 * B);
        CK(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(tlayer_probe_buf), h.size() * 8));
        const int seq[3][12] = {{0, 1, 2, 3, 4, 5, 9, 10, 11, -1}, {0, 1, 2, 3, 4, 5, 6, -1}, {0, 7, 8, 9, 10, 11, -1}};
        double tot = 0;
        for (int k = 1; seq[part][k] >= 0; ++k) {
            double sum = 0;
            for (int b = 0; b < B; ++b) sum += (double)(h[b * 16 + seq[part][k]] - h[b * 16 + seq[part][k - 1]]);
            sum /= B;
            tot += sum;
            printf("  %-36s %9.0f cycles\n", names[seq[part][k]], sum);
        }
        printf("  %-36s %9.0f cycles (s_memtime ticks)\n", "workgroup lifetime", tot);
        if (part != 1) {   // second hidden chunk: matrix work of W1, its epilogue, the wait at the barrier
            double a = 0, b2 = 0, c2 = 0;
            for (int b = 0; b < B; ++b) { a += (double)(h[b * 16 + 13] - h[b * 16 + 12]); b2 += (double)(h[b * 16 + 14] - h[b * 16 + 13]); c2 += (double)(h[b * 16 + 15] - h[b * 16 + 14]); }
            printf("  hidden chunk 1: W1 tile %.0f, epilogue %.0f, barrier %.0f cycles\n", a / B, b2 / B, c2 / B);
        }
    }
    return 0;
}
