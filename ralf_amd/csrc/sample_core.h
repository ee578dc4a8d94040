// The row body of ralf_mask_sample_step (pointwise.hip mask_sample_kernel) as a device function: ONE wave takes the fp32 logits of one row (global memory or
// LDS) to the token -- shared with decode_token.hip, which samples in the kernel that produced the logits.
#pragma once
#include "wave_ops.h"

namespace sample_core {
__device__ __forceinline__ float wave_sum(float v) { return wave::sum64_desc(v); }   // the descending butterfly, bit for bit, without the LDS crossbar (wave_ops.h)
__device__ __forceinline__ float wave_max(float v) { return wave::max64(v); }

// stateless counter-based RNG: 24 uniform bits from (seed, stream id, element index)
__device__ __forceinline__ uint32_t rng24(uint64_t seed, uint64_t call, uint64_t idx) {
    uint64_t z = seed + call * 0x9E3779B97F4A7C15ull + idx * 0xD1B54A32D192ED03ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (uint32_t)(z >> 40);
}

// per-row decode-space mask + token choice (one wave per row of fp32 logits [B,V]):
//   allowed[c] == 0 -> -inf (tokenizer.token_mask row);  forced[b] >= 0 -> that token is the only candidate
//   mode 0: argmax (first maximum);  mode 1: keep logits >= k-th largest, softmax(x/T), one multinomial draw
//   with the counter-based generator (inverse CDF in lane-major order);
//   mode 2 (top_p, helpers/sampling.py:35-58): with the candidates sorted by descending logit, those whose INCLUSIVE cumulative probability exceeds
//           top_p are removed, the first always stays -- without a sort: the kept set is {x >= t} for the smallest t whose tail mass
//           S(t) = sum of p over {x >= t} is <= top_p (the reference's cumulative sum at an element is S(its logit)), found by bisection over the
//           ORDERED BIT PATTERNS of fp32 (32 steps of a masked wave sum), united with the arg-max; equal logits are kept or dropped together;
//   mode 3 (random): softmax(x/T) over all candidates;  mode 4 (gumbel): x/T - log(-log(u + 1e-30) + 1e-30) with a counter-based u per candidate,
//           then the same softmax + draw (as the reference does).
// x: the row's V logits; f: the row's forced token or -1; rrow: the row's number in the WHOLE batch (the draws of a batch decoded in slices are the unsplit
// call's); emit(token) is called by lane 0.
template <class Emit>
__device__ __forceinline__ void mask_sample_row(const float* x, const uint8_t* __restrict__ allowed, const int64_t f, const int mode, const int top_k, const float temperature,
                                                const float top_p, const int64_t* __restrict__ seed, const uint64_t call, const uint64_t rrow, const int V, const int lane, Emit emit) {
    const float NEG = -__builtin_inff();
    if (f >= 0) { if (lane == 0) emit(f); return; }
    constexpr int MAXPER = 16;  // V <= 1024
    float v[MAXPER];
    float best = NEG;
    int bi = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < MAXPER; ++i) {
        const int c = lane + 64 * i, cc = min(c, V - 1);   // unconditional loads of a clamped column (conditional ones are serialised)
        const float xv = x[cc];
        const uint8_t al = allowed ? allowed[cc] : (uint8_t)1;
        v[i] = (c < V && al) ? xv : NEG;
        if (v[i] > best) { best = v[i]; bi = c; }
    }
    // wave arg-max with lowest-index tie break
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o);
        const int oi = __shfl_xor(bi, o);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (mode == 0 || (mode == 1 && top_k <= 1)) { if (lane == 0) emit(bi); return; }
    const float invT = 1.f / temperature;
    if (mode == 4) {   // Gumbel noise on the temperature-scaled logits; the maximum moves
        best = NEG;
#pragma unroll
        for (int i = 0; i < MAXPER; ++i) {
            const int c = lane + 64 * i;
            if (v[i] > NEG) {
                const float u = (rng24((uint64_t)seed[0], call, (1ull << 40) + rrow * 1024 + c) + 0.5f) * (1.f / 16777216.f);
                v[i] = v[i] * invT - __logf(-__logf(u + 1e-30f) + 1e-30f);
            }
            best = fmaxf(best, v[i]);
        }
        best = wave_max(best);
    }
    // k-th largest value: peel the maximum k-1 times (ties are removed one at a time)
    float kth = mode == 1 ? best : NEG;
    if (mode == 1) {
        float w[MAXPER];
#pragma unroll
        for (int i = 0; i < MAXPER; ++i) w[i] = v[i];
        int wi = bi;
        for (int r = 1; r < top_k; ++r) {
#pragma unroll
            for (int i = 0; i < MAXPER; ++i) if (lane + 64 * i == wi) w[i] = NEG;
            float b2 = NEG; int i2 = 0x7fffffff;
#pragma unroll
            for (int i = 0; i < MAXPER; ++i) if (w[i] > b2) { b2 = w[i]; i2 = lane + 64 * i; }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ob = __shfl_xor(b2, o);
                const int oi = __shfl_xor(i2, o);
                if (ob > b2 || (ob == b2 && oi < i2)) { b2 = ob; i2 = oi; }
            }
            if (b2 == NEG) break;
            kth = b2; wi = i2;
        }
    }
    const float pscale = mode == 4 ? 1.f : invT;   // (the Gumbel branch scaled its logits already)
    float p[MAXPER], ls = 0.f;
#pragma unroll
    for (int i = 0; i < MAXPER; ++i) { p[i] = v[i] >= kth && v[i] > NEG ? __expf((v[i] - best) * pscale) : 0.f; ls += p[i]; }
    if (mode == 2) {
        const float mass = wave_sum(ls) * top_p;   // un-normalised p: compare against top_p x total
        // order-preserving map float -> uint (negative floats reversed); bisection for the smallest key t with S(t) <= mass
        auto key_of = [](float f) { const uint32_t b = __float_as_uint(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); };
        uint32_t kv[MAXPER];
#pragma unroll
        for (int i = 0; i < MAXPER; ++i) kv[i] = key_of(v[i]);
        uint32_t lo = 0u, hi = key_of(best);   // S(hi) may exceed the mass (then only the arg-max stays: it is united below); S(lo) = total
        // invariant: S(hi_candidate) checked on the fly; find the smallest t in [lo, hi] with S(t) <= mass, or hi + 1 if none
        uint32_t ans = 0xffffffffu;
        for (int it = 0; it < 33 && lo <= hi; ++it) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            float sm = 0.f;
#pragma unroll
            for (int i = 0; i < MAXPER; ++i) sm += kv[i] >= mid ? p[i] : 0.f;
            sm = wave_sum(sm);
            if (sm <= mass) { ans = mid; if (mid == 0u) break; hi = mid - 1u; }
            else { if (mid == 0xffffffffu) break; lo = mid + 1u; }
        }
        ls = 0.f;
#pragma unroll
        for (int i = 0; i < MAXPER; ++i) { if (!(kv[i] >= ans || lane + 64 * i == bi)) p[i] = 0.f; ls += p[i]; }
    }
    // exclusive scan of the lane sums (lane-major CDF)
    float inc = ls;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const float t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    const float total = __shfl(inc, 63);
    const float u = (rng24((uint64_t)seed[0], call, rrow) + 0.5f) * (1.f / 16777216.f) * total;
    float acc = inc - ls;
    int pick = -1;
#pragma unroll
    for (int i = 0; i < MAXPER; ++i) { if (pick < 0 && p[i] > 0.f && acc + p[i] >= u) pick = lane + 64 * i; acc += p[i]; }
    // the first lane whose range contains u owns the draw
    const bool mine = (u > inc - ls) && (u <= inc) && pick >= 0;
    const unsigned long long ball = __ballot(mine);
    const int owner = ball ? __ffsll((long long)ball) - 1 : -1;
    const int res = owner >= 0 ? __shfl(pick, owner) : bi;
    if (lane == 0) emit(res);
}
}  // namespace sample_core
