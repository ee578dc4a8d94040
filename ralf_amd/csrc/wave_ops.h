// Cross-lane reductions of a 64-lane wavefront WITHOUT the LDS crossbar.  __shfl_xor compiles to ds_bpermute_b32 on gfx950: an address
// computation, an LDS-pipe instruction and an lgkmcnt wait per butterfly level (the LayerNorm row passes and the attention softmax
// are chains of them).  Here the levels inside a 16-lane row are DPP operands of the add itself (quad_perm / row_half_mirror / row_mirror:
// VALU rate), and the two levels across rows are v_permlane16_swap / v_permlane32_swap (gfx950).
// The results are the SAME butterflies: level o combines lane l with lane l ^ o (mirrors differ from xor only in which EQUAL-valued partner a
// lane reads), so every lane ends with the identical value and sum64 is bit-identical to `for (o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o)`.
#pragma once

namespace wave {
template <int CTRL>
__device__ __forceinline__ float dpp(float v) {   // v of the lane selected by the DPP control word (all rows / banks enabled)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
constexpr int QUAD_XOR1 = 0xB1, QUAD_XOR2 = 0x4E, ROW_HALF_MIRROR = 0x141, ROW_MIRROR = 0x140;

// (a, b) := copies of v; after the swap a holds the even rows' (lower half's) values in both partner positions, b the odd rows' (upper half's):
// lane l reads v[l] and v[l ^ 16] (v[l ^ 32]) from the pair.  Inline assembly: the builtins' second result is mis-allocated by this compiler
// (ROCm 7.2 emits `v_add v1, v1, v1` after the swap).  The s_nop covers the VALU-write -> permlane-read hazard the assembler does not see.
// NOT enough behind a v_dot2_f32_bf16 chain that produced the operands (stale reads seen: decode_token.hip uses five wait states there).
__device__ __forceinline__ void swap16(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void swap32(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }

__device__ __forceinline__ float sum_x16_x32(float v) {   // v + lanes ^16, then ^32 (the attention kernels' 4 lane groups of one query / key)
    float a = v, b = v;
    swap16(a, b);
    a = b = a + b;
    swap32(a, b);
    return a + b;
}
__device__ __forceinline__ float max_x16_x32(float v) {
    float a = v, b = v;
    swap16(a, b);
    a = b = fmaxf(a, b);
    swap32(a, b);
    return fmaxf(a, b);
}
__device__ __forceinline__ float sum64(float v) {   // butterfly in ascending level order 1, 2, 4, 8, 16, 32
    v += dpp<QUAD_XOR1>(v);
    v += dpp<QUAD_XOR2>(v);
    v += dpp<ROW_HALF_MIRROR>(v);
    v += dpp<ROW_MIRROR>(v);
    return sum_x16_x32(v);
}
// lane l ^ 4 / l ^ 8 of a row for values that are NOT yet equal inside the 8-lane groups (descending butterflies): rotations of the 16-lane row;
// xor 4 = +4 for the lane quads 0 and 2, -4 (= +12) for the quads 1 and 3 (bank masks)
constexpr int ROW_ROR4 = 0x124, ROW_ROR8 = 0x128, ROW_ROR12 = 0x12C;
__device__ __forceinline__ float xor8(float v) { return dpp<ROW_ROR8>(v); }
__device__ __forceinline__ float xor4(float v) {
    const int a = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ROW_ROR12, 0xF, 0x5, false);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(a, __builtin_bit_cast(int, v), ROW_ROR4, 0xF, 0xA, false));
}
// butterflies in DESCENDING level order 32, 16, 8, 4, 2, 1: bit-identical to `for (o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o)`
__device__ __forceinline__ float sum64_desc(float v) {
    float a = v, b = v;
    swap32(a, b);
    a = b = a + b;
    swap16(a, b);
    v = a + b;
    v += xor8(v);
    v += xor4(v);
    v += dpp<QUAD_XOR2>(v);
    v += dpp<QUAD_XOR1>(v);
    return v;
}
__device__ __forceinline__ float max64(float v) {   // (order is irrelevant for max)
    v = fmaxf(v, dpp<QUAD_XOR1>(v));
    v = fmaxf(v, dpp<QUAD_XOR2>(v));
    v = fmaxf(v, dpp<ROW_HALF_MIRROR>(v));
    v = fmaxf(v, dpp<ROW_MIRROR>(v));
    return max_x16_x32(v);
}
}  // namespace wave
