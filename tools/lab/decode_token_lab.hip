// ralf_decode_token alone on a synthetic decoder (B = 256, 6 layers, M memory rows, position pos): time per launch + the cycle stamps of
// workgroup 0 at the phase boundaries (decode_token.hip built with -DRALF_DT_PROBE).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DRALF_DT_PROBE tools/lab/decode_token_lab.hip ralf_amd/csrc/error.cpp -o tools/lab/_decode_token_lab.bin
//   tools/lab/_decode_token_lab.bin [M=540] [pos=25]
#include <vector>
#include <cstdlib>
#include <cstring>
#include "../../ralf_amd/csrc/decode_token.hip"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static void* dalloc(size_t bytes, int mode) {   // mode 0: zeros, 1: small random bf16, 2: fp32 ones-ish
    void* p; hipMalloc(&p, bytes);
    std::vector<unsigned char> h(bytes);
    unsigned s = 12345 + (unsigned)bytes;
    if (mode == 1) { auto* q = (__bf16*)h.data(); for (size_t i = 0; i < bytes / 2; ++i) { s = s * 1664525u + 1013904223u; q[i] = (__bf16)(((int)(s >> 16) % 2001 - 1000) * 5e-5f); } }
    else if (mode == 2) { auto* q = (float*)h.data(); for (size_t i = 0; i < bytes / 4; ++i) { s = s * 1664525u + 1013904223u; q[i] = 1.f + ((int)(s >> 16) % 201 - 100) * 1e-3f; } }
    hipMemcpy(p, h.data(), bytes, hipMemcpyHostToDevice);
    return p;
}
int main(int argc, char** argv) {
    const int B = 256, NL = 6, L = 52, V = 518, M = argc > 1 ? atoi(argv[1]) : 540, pos = argc > 2 ? atoi(argv[2]) : 25;
    RalfDecodeTokenDesc d;
    memset(&d, 0, sizeof(d));
    d.B = B; d.L = L; d.M = M; d.V = V; d.nlayers = NL; d.pos = pos; d.emb_scale = 16.f; d.eps = 1e-5f;
    d.tok = (const int64_t*)dalloc(B * 8, 0); d.emb = (const float*)dalloc((size_t)V * 256 * 4, 2); d.pe = (const float*)dalloc(64 * 256 * 4, 2);
    d.lnh_g = (const float*)dalloc(1024, 2); d.lnh_b = (const float*)dalloc(1024, 0); d.w_head = dalloc((size_t)V * 256 * 2, 1); d.logits = (float*)dalloc((size_t)B * V * 4, 0);
    for (int i = 0; i < NL; ++i) {
        RalfDecodeTokenLayer& w = d.layer[i];
        w.w_qkv = dalloc(768 * 256 * 2, 1); w.b_qkv = (const float*)dalloc(768 * 4, 0); w.ln1_g = (const float*)dalloc(1024, 2); w.ln1_b = (const float*)dalloc(1024, 0);
        w.w_o1 = dalloc(256 * 256 * 2, 1); w.b_o1 = (const float*)dalloc(1024, 0); w.ln2_g = (const float*)dalloc(1024, 2); w.ln2_b = (const float*)dalloc(1024, 0);
        w.w_q2 = dalloc(256 * 256 * 2, 1); w.b_q2 = (const float*)dalloc(1024, 0); w.w_o2 = dalloc(256 * 256 * 2, 1); w.b_o2 = (const float*)dalloc(1024, 0);
        w.ln3_g = (const float*)dalloc(1024, 2); w.ln3_b = (const float*)dalloc(1024, 0); w.w_f1 = dalloc(1024 * 256 * 2, 1); w.b_f1 = (const float*)dalloc(4096, 0);
        w.w_f2 = dalloc(256 * 1024 * 2, 1); w.b_f2 = (const float*)dalloc(1024, 0);
        w.self_kv = dalloc((size_t)B * L * 512 * 2, 1); w.cross_kv = (getenv("LAB_SHARE_KV") && i > 0) ? d.layer[0].cross_kv : dalloc((size_t)B * 8 * M * 64 * 2, 1);   // LAB_SHARE_KV=1: one K / V cache for all layers (infinity-cache resident)
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 3; ++r) if (ralf_decode_token(&d, nullptr)) { printf("launch failed\n"); return 1; }
    CK(hipDeviceSynchronize());
    hipEventRecord(e0, 0);
    const int it = 20;
    for (int r = 0; r < it; ++r) ralf_decode_token(&d, nullptr);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("M = %d, pos = %d: %.1f us per launch (%.1f per layer)\n", M, pos, ms * 1e3 / it, ms * 1e3 / it / NL);
    unsigned long long st[256];
    CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(ralf_dt_probe), sizeof(st)));
    const char* nm[12] = {"LN1", "gemv qkv", "qkv epi + self-attn", "gemv o1", "x + LN2 + gemv q2", "cross K pass", "softmax", "cross V pass", "gemv o2", "x + LN3 .. gemv f1", "relu", "gemv f2"};
    // stamps: 0 = after the embedding; per layer sb + {0: after LN1, 1: after qkv, 2: after self-attn, 3: after o1 + x, 4: before the K pass, 5: after it, 6: after softmax,
    //          7: after the V pass + o, 8: after o2 + x + LN3, 9: after f1, 10: after relu, 11: after f2}
    for (int li = 0; li < NL; li += 5) {
        const int sb = 1 + li * 12;
        unsigned long long prev = li == 0 ? st[0] : st[sb - 1];
        printf("layer %d (cycles of the 100 MHz.. s_memtime clock):", li);
        for (int j = 0; j < 12; ++j) { printf(" %s %llu |", nm[j], st[sb + j] - prev); prev = st[sb + j]; }
        printf("\n");
    }
    printf("whole kernel, workgroup 0: %llu ticks\n", st[1 + 12 * 8] - st[0]);
    return 0;
}
