// Shared host-side helpers for libralf_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/ralf_hip.h"
#include "wave_ops.h"

namespace ralf {
void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return RALF_ERR_LAUNCH;
    }
    return RALF_OK;
}
}  // namespace ralf

#define RALF_REQUIRE(cond, ...)              \
    do {                                     \
        if (!(cond)) {                       \
            ralf::set_error(__VA_ARGS__);    \
            return RALF_ERR_INVALID;         \
        }                                    \
    } while (0)

static inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// exact (erf) GELU and its derivative, shared by the GEMM epilogues and tlayer.hip (the derivative's last step is an EXPLICIT fma: left to the
// compiler, `a + b * c` is contracted in one kernel and not in another, and the two then differ in the last bit)
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad(float x) {
    return __fmaf_rn(x * 0.3989422804014327f, __expf(-0.5f * x * x), 0.5f * (1.f + erff(x * 0.70710678118654752f)));
}

// Element dropout mask (GEMM epilogues, ralf_dropout, ralf_scale_pe_dropout, the LayerNorm backward's masked gradient, tlayer.hip):
//   keep(e) = 16-bit field (e & 3) of drop_hash4(seed, call, e >> 2) >= p * 2^16
// ONE hash decides FOUR consecutive elements of the contiguous tensor (the kernels hold 4 or 8 consecutive elements per lane).  The stream
// (seed, call) gives two 32-bit keys (drop_keys: uniform over the launch, scalar unit); each key and the group index go through the attention
// kernels' full-rate 24-bit-multiply mixer (attn_rng2x16 below) for two fields each.
// (A 64-bit splitmix per ELEMENT -- three 64-bit multiplies = ~12 quarter-rate 32-bit multiplies -- took a quarter of the transformer-layer
//  kernel's time: tools/tlayer_probe.hip.)  p is quantised to 1 / 65536 (0.1 -> 0.099991); kept elements are scaled by 1 / (1 - p), nominal p.
// Checked in tests/test_ops_gpu.py: keep rate, and correlation between the four fields, neighbouring groups, rows and streams.
__device__ __forceinline__ uint32_t attn_rng2x16(uint32_t rowkey, uint32_t pair);
struct DropKeys { uint32_t k0, k1; };   // the two 32-bit stream keys of (seed, call): uniform over a launch (scalar unit), 32-bit finalisers only -- a
                                        // 64-bit splitmix here was ~40 dependent scalar instructions, recomputed wherever the compiler did not hoist it
__device__ __forceinline__ uint32_t drop_fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
__device__ __forceinline__ DropKeys drop_keys(uint64_t seed, uint64_t call) {
    const uint32_t lo = (uint32_t)seed, hi = (uint32_t)(seed >> 32), c = (uint32_t)call ^ (uint32_t)(call >> 32);
    const uint32_t k0 = drop_fmix32(lo ^ drop_fmix32(c * 0x9E3779B1u + 0x7F4A7C15u) ^ (hi * 0x85EBCA77u));
    const uint32_t k1 = drop_fmix32((k0 + 0x9E3779B9u) ^ (hi + c * 0xC2B2AE3Du));
    return DropKeys{k0, k1};
}
__device__ __forceinline__ uint64_t drop_hash4k(const DropKeys k, uint64_t group) {
    const uint32_t glo = (uint32_t)group & 0xffffffu;
    const uint32_t hi = __umul24((uint32_t)(group >> 24), 0x85EBCBu);   // (0 below 2^26 elements)
    const uint32_t a = attn_rng2x16(k.k0 ^ hi, glo);
    const uint32_t b = attn_rng2x16(k.k1 ^ hi, glo);
    return (uint64_t)a | ((uint64_t)b << 32);
}
__device__ __forceinline__ uint64_t drop_hash4k32(const DropKeys k, uint32_t group) {   // the same value for group < 2^32 with 32-bit index arithmetic
    const uint32_t hi = __umul24(group >> 24, 0x85EBCBu);
    const uint32_t a = attn_rng2x16(k.k0 ^ hi, group & 0xffffffu);
    const uint32_t b = attn_rng2x16(k.k1 ^ hi, group & 0xffffffu);
    return (uint64_t)a | ((uint64_t)b << 32);
}
__device__ __forceinline__ uint64_t drop_hash4(uint64_t seed, uint64_t call, uint64_t group) { return drop_hash4k(drop_keys(seed, call), group); }
__device__ __forceinline__ uint32_t drop_thr16(float p) { return (uint32_t)(p * 65536.f); }
__device__ __forceinline__ bool drop_keep(uint64_t h, int field, uint32_t thr16) { return ((uint32_t)(h >> (16 * field)) & 0xffffu) >= thr16; }
__device__ __forceinline__ bool drop_keep1(uint64_t seed, uint64_t call, uint64_t e, uint32_t thr16) {
    return drop_keep(drop_hash4(seed, call, e >> 2), (int)(e & 3), thr16);
}
// W (4 or 8) consecutive elements starting at e0: v[q] = keep ? v[q] * inv : 0
template <int W>
__device__ __forceinline__ void drop_apply_k(float (&v)[W], const DropKeys k, uint64_t e0, uint32_t thr16, float inv) {
    if ((e0 & 3) == 0 && W % 4 == 0) {
#pragma unroll
        for (int g = 0; g < W / 4; ++g) {
            const uint64_t h = drop_hash4k(k, (e0 >> 2) + g);
#pragma unroll
            for (int q = 0; q < 4; ++q) v[4 * g + q] = drop_keep(h, q, thr16) ? v[4 * g + q] * inv : 0.f;
        }
    } else {
#pragma unroll
        for (int q = 0; q < W; ++q) v[q] = drop_keep(drop_hash4k(k, (e0 + q) >> 2), (int)((e0 + q) & 3), thr16) ? v[q] * inv : 0.f;
    }
}
template <int W>
__device__ __forceinline__ void drop_apply(float (&v)[W], uint64_t seed, uint64_t call, uint64_t e0, uint32_t thr16, float inv) {
    drop_apply_k<W>(v, drop_keys(seed, call), e0, thr16, inv);
}

// Attention-probability dropout mask (all attention kernels, forward and backward, VALU and MFMA variants):
// keep(b, h, q, key) = attn_rng24(attn_rowkey(seed, call, (b*H + h)*Sq + q), key) >= p * 2^24.
// The 64-bit mix runs once per query ROW (lane-invariant in the per-query kernels, staged through LDS in the per-key
// kernels); each element then costs one 32-bit integer mix.  The single-level 64-bit hash per element (three 64-bit
// multiplies = ~10 quarter-rate v_mul instructions) was most of the attention kernels' time at S = 256.
__device__ __forceinline__ uint32_t attn_rowkey(uint64_t seed, uint64_t call, uint64_t row) {
    uint64_t z = seed + call * 0x9E3779B97F4A7C15ull + row * 0xD1B54A32D192ED03ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (uint32_t)(z >> 32) ^ (uint32_t)z;
}
// keep(row, key) = field(attn_rng2x16(rowkey(row), key >> 1), key & 1) >= p * 2^16: ONE hash decides a PAIR of adjacent keys
// (two 16-bit fields), so the kernels whose lanes hold adjacent keys of a row (forward, dQ) hash once per two scores.
// Full-rate integer instructions only (24-bit multiplies, shifts, xors): the attention kernels are VALU-bound on their per-score
// work, and a 32-bit v_mul_lo_u32 costs four issue slots.  Checked on 4 M (row, key) pairs (also with sequential row keys):
// keep rate 0.8998-0.9000 at p = 0.1; correlation within a pair, between pairs and between rows < 1.1e-3.
// (A two-round form -- without the last multiply -- was tried: three instructions fewer per hash, keep rate and correlations ALONG a row as good,
//  but masks of different streams / rows correlate at 0.5-2.5 % at p = 0.5, tools/dropout_hash_stats.py; the attention kernels gain ~5 % from it.  Not taken.)
__device__ __forceinline__ uint32_t attn_rng2x16(uint32_t rowkey, uint32_t pair) {
    uint32_t x = rowkey ^ __umul24(pair, 0x9E3779u);
    x ^= x >> 16; x = __umul24(x, 0xEB352Du);
    x ^= x >> 15; x = __umul24(x, 0xA68B6Bu);
    x ^= x >> 15;
    return x;
}
__device__ __forceinline__ uint32_t attn_thr16(float p_drop) { return (uint32_t)(p_drop * 65536.f); }
__device__ __forceinline__ bool attn_keep(uint32_t rowkey, uint32_t key, uint32_t thr16) {
    const uint32_t h = attn_rng2x16(rowkey, key >> 1);
    return ((key & 1u) ? (h >> 16) : (h & 0xffffu)) >= thr16;
}
