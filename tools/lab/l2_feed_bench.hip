// How fast can ONE CU pull L2-resident operand bytes?  Every workgroup streams the same few-MB region (L2 / infinity-cache resident after the
// first pass) with (a) global_load_dwordx4 into registers, (b) global_load_lds_dwordx4 into LDS (the GEMM ring's primitive), for 4 / 8 / 16 waves
// per CU and 1 / 2 / 4 KiB row strides.  Prints bytes per clock per CU (2.4 GHz) and the chip aggregate.
//   hipcc --offload-arch=gfx950 -O3 tools/lab/l2_feed_bench.hip -o tools/lab/_l2_feed_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

// MODE 0: registers, 1: LDS-DMA.  Each wave reads `iters` x UNR KiB; region = `region_kb` KiB shared by all workgroups of an XCD-ish group
template <int MODE, int NT>
__global__ __launch_bounds__(NT) void feed(const unsigned char* __restrict__ src, size_t region, int iters, unsigned* sink, int tilelike) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[64 * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = NT / 64;
    constexpr int UNR = 8;
    u32x4 acc = {0, 0, 0, 0};
    // tilelike: lane l reads 16 B of row (l >> 3) at column (l & 7) * 16 of a [rows][row_stride] matrix (8 rows x 128 B per instruction: the GEMM ring's shape)
    // else: 1 KiB contiguous per instruction
    const size_t rstride = tilelike ? (size_t)tilelike : 128;
    size_t off = ((size_t)blockIdx.x * 7919 * 1024 + (size_t)wave * 8 * rstride) % region;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            size_t o = off + (size_t)(lane >> 3) * rstride + (lane & 7) * 16 + (size_t)u * 128 * (tilelike ? 1 : 8);
            if (o >= region) o -= region;
            if constexpr (MODE == 0) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(src + o);
                acc += v;
            } else {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + o), LDS_PTR(void, lds + (wave * UNR + u) % 64 * 1024), 16, 0, 0);
            }
        }
        off += (size_t)nw * 8 * rstride;
        if (off >= region) off -= region;
        if constexpr (MODE == 1) { if ((it & 3) == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
    }
    if constexpr (MODE == 1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); acc.x = lds[threadIdx.x * 4]; }
    if (acc.x + acc.y + acc.z + acc.w == 0x12345678u) sink[0] = 1;
}

template <int MODE, int NT>
float run(const unsigned char* src, size_t region, int iters, unsigned* sink, int blocks, int tilelike) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((feed<MODE, NT>), dim3(blocks), dim3(NT), 0, 0, src, region, iters, sink, tilelike);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((feed<MODE, NT>), dim3(blocks), dim3(NT), 0, 0, src, region, iters, sink, tilelike);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    unsigned char* src; unsigned* sink;
    const size_t maxr = 256u << 20;
    CK(hipMalloc(&src, maxr + (1 << 20))); CK(hipMalloc(&sink, 64));
    CK(hipMemset(src, 1, maxr + (1 << 20)));
    printf("%-8s %-5s %-9s %-10s %8s %10s %10s\n", "mode", "waves", "region", "pattern", "us", "B/clk/CU", "TB/s chip");
    for (int tilelike : {0, 512, 4608}) {
        for (size_t region : {(size_t)1 << 20, (size_t)16 << 20, (size_t)96 << 20}) {
            for (int cfg = 0; cfg < 6; ++cfg) {
                const int mode = cfg / 3, wsel = cfg % 3;
                const int nt = wsel == 0 ? 256 : wsel == 1 ? 512 : 1024;
                const int iters = 256 / (nt / 256);   // every CU reads the same total: 4 waves x 256 iters x 8 KiB = 8 MiB
                float ms;
                if (mode == 0) ms = nt == 256 ? run<0, 256>(src, region, iters, sink, 256, tilelike) : nt == 512 ? run<0, 512>(src, region, iters, sink, 256, tilelike) : run<0, 1024>(src, region, iters, sink, 256, tilelike);
                else ms = nt == 256 ? run<1, 256>(src, region, iters, sink, 256, tilelike) : nt == 512 ? run<1, 512>(src, region, iters, sink, 256, tilelike) : run<1, 1024>(src, region, iters, sink, 256, tilelike);
                const double bytes_cu = (double)(nt / 64) * iters * 8 * 1024;
                char pat[32]; snprintf(pat, sizeof(pat), tilelike ? "8x128B/%d" : "1KiB", tilelike);
                printf("%-8s %-5d %6zuMiB %-10s %8.1f %10.1f %10.2f\n", mode ? "lds-dma" : "regs", nt / 64, region >> 20, pat, ms * 1e3, bytes_cu / (ms * 1e-3 * 2.4e9), bytes_cu * 256 / (ms * 1e-3) / 1e12);
            }
        }
    }
    return 0;
}
