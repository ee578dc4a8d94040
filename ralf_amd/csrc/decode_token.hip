// One KV-cached decode step of the WHOLE decoder stack, one workgroup per SAMPLE (round 6).
//
// Replaces, for one generated token of every batch element, the per-token decoder call of the reference's sampling loop
//   image2layout/train/models/retrieval_augmented_autoreg.py:274-279 (self.decoder(tgt, memory, ...) on the growing prefix)
//   common/common.py:84-135 (BaseDecoder.forward: embedding, 6 x nn.TransformerDecoderLayer(norm_first), head)
// i.e. the ~45 launches per token of nn.decoder_step's fused path (embedding, per layer: self-attention block, out-projection, cross-attention
// block, out-projection, LayerNorm + feed-forward (2), head) by ONE launch: samples never interact, so a workgroup carries its sample's
// residual row (256 floats in LDS) through all layers with workgroup barriers only.
//   * the layer weights (1.83 MB per layer, bf16, row-major [n_out][n_in]) stream from L2: every workgroup reads all of them, all 256 nearly in step;
//   * matrix-vector products on the vector unit (v_dot2_f32_bf16): a 32-lane group owns one output row per pass (16 bytes per lane, one 512-byte
//     row per load instruction), its partial sums meet in a DPP / permlane butterfly; four (eight) rows of a group are in flight at once;
//   * the self-attention reads its <= 64 cached keys per head from the [B, L, 2d] cache (L2), one wave per head;
//   * the cross-attention streams the sample's K and V of the layer from HBM (head-pair-major cache [B, 8, M, 64], slice 4 kv + hp:
//     nn.decoder_init_cache): wave = (head pair, key half), eight 16-byte loads per lane in flight.
// Rounding points are those of the unfused path (LayerNorm outputs, q / k / v, attention outputs, the residual stream and the feed-forward hidden
// layer in bf16; accumulation, softmax and logits in fp32), summation orders differ: bf16 throughput mode only (the fp32 parity mode keeps the
// per-kernel path).  d = 256, 8 heads, feed-forward 1024, <= 8 layers, <= 64 self-attention keys, <= DT_MAXM memory rows.
#include "common.h"
#include "sample_core.h"

namespace {
typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int D = 256, NH = 8, DH = 32, FF = 1024, NT = 512, DT_MAXM = 1024, DT_MAXL = 64, KVS_LD = 2 * D + 8;

struct Lds {
    float x[D];                 // the residual stream (values representable in bf16)
    float y[FF];                // output of the current matrix-vector product
    float sc[NH][DT_MAXM];      // attention scores -> probabilities, per head
    float part[2][D];           // cross-attention: the two key halves' weighted sums
    float red[32];              // LayerNorm / softmax hand-offs
    __attribute__((aligned(16))) bf16 hb[D];    // input of the current 256-wide matrix-vector product (LayerNorm / attention output)
    __attribute__((aligned(16))) bf16 hb2[FF];  // the feed-forward hidden layer
    __attribute__((aligned(16))) bf16 kvs[DT_MAXL][KVS_LD];   // self-attention: the sample's cached k | v rows (row stride 1040 bytes: 16 rows of a 16-byte read hit 64 banks)
};

__device__ __forceinline__ float bfr(float v) { return (float)(bf16)v; }
__device__ __forceinline__ float dot8(const bf16x8 w, const bf16x8 x, float acc) {
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(w, w, 0, 1), __builtin_shufflevector(x, x, 0, 1), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(w, w, 2, 3), __builtin_shufflevector(x, x, 2, 3), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(w, w, 4, 5), __builtin_shufflevector(x, x, 4, 5), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(w, w, 6, 7), __builtin_shufflevector(x, x, 6, 7), acc, false);
    return acc;
}
// sum over the 32 lanes l ^ {1, 2, 4, 8, 16} (a half wave)
__device__ __forceinline__ float sum32(float v) {
    v += wave::dpp<wave::QUAD_XOR1>(v);
    v += wave::dpp<wave::QUAD_XOR2>(v);
    v += wave::dpp<wave::ROW_HALF_MIRROR>(v);
    v += wave::dpp<wave::ROW_MIRROR>(v);
    float a = v, b = v;
    wave::swap16(a, b);
    return a + b;
}

// Matrix-vector products  out[r] = epi(sum_k W[r][k] in[k] + bias[r]),  r < R,  W bf16 row-major [R][IN] in global memory (L2), the input vector
// in LDS (bf16).  A 32-lane group owns one output row per pass (16 bytes per lane = one 512-byte row piece per load instruction), sixteen groups
// = sixteen rows per pass; EIGHT passes (IN = 1024: two, of four pieces each) are requested at once and the next eight while these are
// multiplied: 256 bytes per lane in flight -- with four rows in flight the 256-wide products ran at 21-25 B/clk per CU, the 1024-wide one
// (eight loads in flight) at 47 (tools/lab/decode_token_lab.hip).  The epilogue runs in the lane that ends up with the row's sum.  Ends with a barrier.
enum { EPI_QKV, EPI_RES, EPI_Q, EPI_RELU, EPI_LOGITS };
template <int EPI>
__device__ __forceinline__ void gemv_epi(int r, float a, float bias, Lds& L, bf16* kvrow, float* logits) {
    if constexpr (EPI == EPI_QKV) {            // q | k | v in bf16; k / v of the new token also go to their cache row
        const float v = bfr(a + bias);
        L.y[r] = v;
        if (r >= D) kvrow[r - D] = (bf16)v;
    } else if constexpr (EPI == EPI_RES) {     // x += W in + b (the residual stream, stored in bf16 by the unfused path)
        L.x[r] = bfr(a + bias + L.x[r]);
    } else if constexpr (EPI == EPI_Q) {       // the cross-attention's query, pre-scaled
        L.y[r] = bfr(a + bias) * 0.17677669529663687f;
    } else if constexpr (EPI == EPI_RELU) {    // the feed-forward hidden layer
        L.hb2[r] = (bf16)fmaxf(a + bias, 0.f);
    } else {
        logits[r] = a;
        L.y[r] = a;                            // (the in-kernel token choice reads the row from LDS: V <= 1024)
    }
}
// the 32-lane sums of EIGHT values per lane in 14 cross-lane steps instead of 8 x 5: each level halves the values a lane keeps (its partner
// keeps the other half), so lane gl ends with the complete sum of row u = 4 (gl >> 4) + 2 ((gl >> 3) & 1) + ((gl >> 2) & 1), replicated over
// its quad.  (One butterfly per row made the 256-wide products issue-bound: 24 B/clk per CU whatever the number of loads in flight.)
// v_permlane16_swap right behind the v_dot2_f32_bf16 chain that produced its operands: wave::swap16's `s_nop 1` covers a plain VALU write, the dot
// unit's result was seen stale (a few wrong rows per product); five wait states are not measurable here
__device__ __forceinline__ void swap16_after_dot(float& a, float& b) { asm volatile("s_nop 4\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ float reduce8(const float (&a)[8], int gl) {
    float c[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {   // level 16: v_permlane16_swap leaves BOTH halves of row i in the even 16-lane rows, of row i + 4 in the odd ones
        float x = a[i], y = a[i + 4];
        swap16_after_dot(x, y);
        c[i] = x + y;
    }
    const bool b3 = (gl & 8) != 0, b2 = (gl & 4) != 0;
    float e[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {   // level 8: a lane keeps rows j (bit 3 clear) or j + 2 (set) and hands the other to its partner
        const float keep = b3 ? c[j + 2] : c[j], send = b3 ? c[j] : c[j + 2];
        e[j] = keep + wave::xor8(send);
    }
    const float keep = b2 ? e[1] : e[0], send = b2 ? e[0] : e[1];   // level 4
    float f = keep + wave::xor4(send);
    f += wave::dpp<wave::QUAD_XOR2>(f);
    f += wave::dpp<wave::QUAD_XOR1>(f);
    return f;
}
__device__ __forceinline__ float reduce2(float a0, float a1) {   // two values: row gl >> 4, replicated over the 16-lane row
    swap16_after_dot(a0, a1);
    float f = a0 + a1;
    f += wave::xor8(f);
    f += wave::xor4(f);
    f += wave::dpp<wave::QUAD_XOR2>(f);
    f += wave::dpp<wave::QUAD_XOR1>(f);
    return f;
}

template <int IN, int EPI>
__device__ __forceinline__ void gemv(const bf16* __restrict__ W, const float* __restrict__ bias, int R, Lds& L, int tid, bf16* kvrow = nullptr, float* logits = nullptr) {
    const int g = tid >> 5, gl = tid & 31;
    constexpr int NC = IN / 256;               // 16-byte pieces per lane and row
    constexpr int U = 8 / NC;                  // rows of a group per batch of loads
    const bf16* in = IN == 1024 ? L.hb2 : L.hb;
    bf16x8 xr[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) xr[c] = *reinterpret_cast<const bf16x8*>(in + 8 * gl + 256 * c);
    const int umine = U == 8 ? 4 * (gl >> 4) + 2 * ((gl >> 3) & 1) + ((gl >> 2) & 1) : (gl >> 4);   // the row of a batch this lane finishes (reduce8 / reduce2)
    const bool writer = U == 8 ? (gl & 3) == 0 : (gl & 15) == 0;
    bf16x8 wa[U][NC], wb[U][NC];
    float ba, bb;
    auto load = [&](bf16x8 (&w)[U][NC], float& bv, int r0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = min(r0 + 16 * u + g, R - 1);
#pragma unroll
            for (int c = 0; c < NC; ++c) w[u][c] = *reinterpret_cast<const bf16x8*>(W + (size_t)r * IN + 8 * gl + 256 * c);
        }
        bv = (EPI != EPI_LOGITS) ? bias[min(r0 + 16 * umine + g, R - 1)] : 0.f;
    };
    auto comp = [&](const bf16x8 (&w)[U][NC], float bv, int r0) {
        float a[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            a[u] = 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c) a[u] = dot8(w[u][c], xr[c], a[u]);
        }
        float f;
        if constexpr (U == 8) f = reduce8(a, gl); else f = reduce2(a[0], a[1]);
        const int r = r0 + 16 * umine + g;
        if (writer && r < R) gemv_epi<EPI>(r, f, bv, L, kvrow, logits);
    };
    constexpr int STEP = 16 * U;
    load(wa, ba, 0);
    for (int r0 = 0; r0 < R; r0 += 2 * STEP) {
        if (r0 + STEP < R) load(wb, bb, r0 + STEP);
        comp(wa, ba, r0);
        if (r0 + 2 * STEP < R) load(wa, ba, r0 + 2 * STEP);
        if (r0 + STEP < R) comp(wb, bb, r0 + STEP);
    }
    __syncthreads();
}

// hb[0..255] = bf16(LayerNorm(x) * g + b) (the formulas of ln_fwd_kernel: mean, then the centred sum of squares); g / b = this thread's element of
// the parameters, requested long before (a load behind the statistics cost ~1 k cycles per LayerNorm).  Ends with a barrier.
__device__ __forceinline__ void layer_norm(float gam, float bet, float eps, Lds& L, int tid) {
    const int lane = tid & 63, wv = tid >> 6;
    const float xv = tid < D ? L.x[tid] : 0.f;
    const float s = wave::sum64(xv);
    if (lane == 0) L.red[wv] = s;
    __syncthreads();
    const float mu = (L.red[0] + L.red[1] + L.red[2] + L.red[3]) * (1.f / D);
    const float dv = tid < D ? xv - mu : 0.f;
    const float q = wave::sum64(dv * dv);
    if (lane == 0) L.red[8 + wv] = q;
    __syncthreads();
    const float rs = rsqrtf((L.red[8] + L.red[9] + L.red[10] + L.red[11]) * (1.f / D) + eps);
    if (tid < D) L.hb[tid] = (bf16)(dv * rs * gam + bet);
    __syncthreads();
}

// tools/lab/decode_token_lab.hip builds this file with -DRALF_DT_PROBE: cycle stamps of workgroup 0 at the phase boundaries
#ifdef RALF_DT_PROBE
__device__ unsigned long long ralf_dt_probe[256];
#define DT_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) ralf_dt_probe[(i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define DT_STAMP(i)
#endif

#ifndef DT_KV_LOAD
// the memory keys / values are read once per token: streaming loads.  Plain loads let the 848 MB per token sweep the weights out of the L2 (350.9 us per
// token against 297.8); with all six layers pointed at ONE cache (141 MB: infinity-cache resident) the K / V passes take the same cycles as from HBM
// (tools/lab/decode_token_lab.hip LAB_SHARE_KV, profiles/r06_decode_kv_source.txt) -- the phase is bound by the rate at which a CU takes the rows
// in, wherever they come from; only fewer bytes would shorten it.
#define DT_KV_LOAD(p) __builtin_nontemporal_load(p)
#endif
__global__ __launch_bounds__(NT) void decode_token_kernel(const RalfDecodeTokenDesc d) {
    __shared__ Lds L;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, b = blockIdx.x;
    // (every workgroup walks the same phases at the same pace, so all of them ask HBM for the memory keys / values at once and the L2 for the weights at
    //  once.  Starting every second workgroup of an XCD late -- its HBM phases inside the others' weight phases -- was measured: 297.6 us per token
    //  without, 286.5 / 292.5 / 290.7 / 305 with 20 / 35 / 50 / 65 k cycles of delay (profiles/r06_decode_stagger.txt): the delay costs what the
    //  interleaving gains, not kept.)
    const int pos = d.pos_vec ? d.pos_vec[b] : d.pos;             // this element's position = its number of cached rows
    const uint8_t* kpm = d.kpm ? d.kpm + (int64_t)b * d.kpm_bs : nullptr;
    const float scale = 0.17677669529663687f;                     // 32^-0.5
    // ---- embedding: x = bf16(emb[tok] * sqrt(d) + pe[pos]) (ralf_embed_fwd) ----
    if (tid < D) L.x[tid] = bfr(d.emb[(int64_t)d.tok[b] * D + tid] * d.emb_scale + d.pe[(int64_t)pos * D + tid]);
    __syncthreads();
    DT_STAMP(0);
    const int pt = tid & (D - 1);                                 // (threads 256 .. 511 request the same elements: no divergent loads)
    const float gh = d.lnh_g[pt], bh = d.lnh_b[pt];
    float g1 = d.layer[0].ln1_g[pt], b1 = d.layer[0].ln1_b[pt];   // LayerNorm 1 of the first layer; the later ones are requested a phase ahead
    for (int li = 0; li < d.nlayers; ++li) {
        const RalfDecodeTokenLayer& W = d.layer[li];
        const int sb = 1 + li * 12; (void)sb;
        const float g2 = W.ln2_g[pt], b2 = W.ln2_b[pt], g3 = W.ln3_g[pt], b3 = W.ln3_b[pt];
        // the sample's cached k | v rows (<= 63 KB from L2) start their way now and land in LDS behind the q | k | v product: one round trip
        // instead of one per key (the self-attention block took 6.4 k cycles with per-key loads)
        constexpr int KVU = DT_MAXL * 2 * D / 8 / NT;             // 16-byte pieces per thread: 8
        bf16x8 kvr[KVU];
        {
            const bf16* KV = (const bf16*)W.self_kv + (int64_t)b * d.L * 2 * D;
#pragma unroll
            for (int u = 0; u < KVU; ++u) {
                const int piece = tid + NT * u, row = piece >> 6;  // 64 pieces per 1-KiB row
                kvr[u] = *reinterpret_cast<const bf16x8*>(KV + (int64_t)min(row, max(pos - 1, 0)) * 2 * D + (piece & 63) * 8);
            }
        }
        // ================= self-attention block: x += Wo1 attn(LN1(x)) + bo1 =================
        layer_norm(g1, b1, d.eps, L, tid);
        DT_STAMP(sb + 0);
        gemv<256, EPI_QKV>((const bf16*)W.w_qkv, W.b_qkv, 3 * D, L, tid, (bf16*)W.self_kv + ((int64_t)b * d.L + pos) * 2 * D);   // (q | k | v in bf16; k / v of the new token -> cache row `pos`)
        DT_STAMP(sb + 1);
#pragma unroll
        for (int u = 0; u < KVU; ++u) {
            const int piece = tid + NT * u, row = piece >> 6;
            if (row < pos) *reinterpret_cast<bf16x8*>(&L.kvs[row][(piece & 63) * 8]) = kvr[u];
        }
        __syncthreads();
        {   // one wave per head; lane = key (cached rows 0 .. pos - 1, the new one = pos)
            const int h = wv;
            float s = -__builtin_inff();
            if (lane <= pos) {
                float a = 0.f;
                if (lane < pos) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const bf16x8 kk = *reinterpret_cast<const bf16x8*>(&L.kvs[lane][h * DH + 8 * c]);
#pragma unroll
                        for (int e = 0; e < 8; ++e) a += L.y[h * DH + 8 * c + e] * scale * (float)kk[e];
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < DH; ++e) a += L.y[h * DH + e] * scale * L.y[D + h * DH + e];
                }
                s = (kpm && kpm[lane]) ? -__builtin_inff() : a;
            }
            const float m = wave::max64(s);
            const float mref = m > -__builtin_inff() ? m : 0.f;
            const float p = lane <= pos ? __expf(s - mref) : 0.f;
            const float l = wave::sum64(p);
            L.sc[h][lane] = p;
            // (the wave's own writes: visible to its lanes after the LDS counter drains -- no other wave touches sc[h])
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane < DH) {
                float o = 0.f;
#pragma unroll 8
                for (int j = 0; j < pos; ++j) o += L.sc[h][j] * (float)L.kvs[j][D + h * DH + lane];
                o += L.sc[h][pos] * L.y[2 * D + h * DH + lane];
                L.hb[h * DH + lane] = (bf16)(l > 0.f ? o / l : 0.f);
            }
        }
        __syncthreads();
        DT_STAMP(sb + 2);
        gemv<256, EPI_RES>((const bf16*)W.w_o1, W.b_o1, D, L, tid);
        // ================= cross-attention block: x += Wo2 attn(LN2(x) Wq^T + bq; memory K, V) + bo2 =================
        DT_STAMP(sb + 3);
        // wave = (head pair, key half); lane = (key slot of eight, 16-byte piece of the pair's 128-byte row).  The FIRST batch of keys (128 per wave: eight
        // 16-byte loads per lane) starts its way before LayerNorm 2 and the query projection -- it needs neither --, every later batch while
        // the one before it is multiplied, and the first batch of values before the softmax.
        const int hp = wv & 3, half = wv >> 2, chunk = lane & 7, slot = lane >> 3, head = chunk >> 2;
        const int M = d.M, kstart = half * 8 + slot;
        const bf16* Kb = (const bf16*)W.cross_kv + ((int64_t)b * 8 + hp) * M * 64 + chunk * 8;
        const bf16* Vb = Kb + (int64_t)4 * M * 64;
        auto ldkv = [&](bf16x8 (&r)[8], const bf16* base, int k0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) r[u] = DT_KV_LOAD(reinterpret_cast<const bf16x8*>(base + (int64_t)min(k0 + 16 * u, M - 1) * 64));
        };
        bf16x8 ka[8], kb[8];
        ldkv(ka, Kb, kstart);
        layer_norm(g2, b2, d.eps, L, tid);
        gemv<256, EPI_Q>((const bf16*)W.w_q2, W.b_q2, D, L, tid);
        {
            DT_STAMP(sb + 4);
            float qv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) qv[e] = L.y[hp * 64 + chunk * 8 + e];
            auto score = [&](const bf16x8 (&r)[8], int k0) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    float sv = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) sv += qv[e] * (float)r[u][e];
                    sv += wave::dpp<wave::QUAD_XOR1>(sv);
                    sv += wave::dpp<wave::QUAD_XOR2>(sv);
                    const int key = k0 + 16 * u;
                    if ((chunk & 3) == 0 && key < M) L.sc[hp * 2 + head][key] = sv;
                }
            };
            // pass 1: scores
            for (int k0 = kstart; k0 < M; k0 += 256) {
                if (k0 + 128 < M) ldkv(kb, Kb, k0 + 128);
                score(ka, k0);
                if (k0 + 256 < M) ldkv(ka, Kb, k0 + 256);
                if (k0 + 128 < M) score(kb, k0 + 128);
            }
            ldkv(ka, Vb, kstart);   // (the first values: on their way through the softmax)
            __syncthreads();
            DT_STAMP(sb + 5);
            {   // softmax per head: wave = head
                float m = -__builtin_inff();
                for (int k = lane; k < M; k += 64) m = fmaxf(m, L.sc[wv][k]);
                m = wave::max64(m);
                float l = 0.f;
                for (int k = lane; k < M; k += 64) {
                    const float p = __expf(L.sc[wv][k] - m);
                    L.sc[wv][k] = p;
                    l += p;
                }
                l = wave::sum64(l);
                if (lane == 0) L.red[16 + wv] = l;
            }
            __syncthreads();
            DT_STAMP(sb + 6);
            // pass 2: weighted sum of the values
            float acc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = 0.f;
            auto pv = [&](const bf16x8 (&r)[8], int k0) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int key = k0 + 16 * u;
                    const float p = key < M ? L.sc[hp * 2 + head][key] : 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] += p * (float)r[u][e];
                }
            };
            for (int k0 = kstart; k0 < M; k0 += 256) {
                if (k0 + 128 < M) ldkv(kb, Vb, k0 + 128);
                pv(ka, k0);
                if (k0 + 256 < M) ldkv(ka, Vb, k0 + 256);
                if (k0 + 128 < M) pv(kb, k0 + 128);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {   // the eight key slots of the wave (lane bits 3, 4, 5)
                acc[e] += wave::xor8(acc[e]);
                acc[e] = wave::sum_x16_x32(acc[e]);
            }
            if (slot == 0) {
#pragma unroll
                for (int e = 0; e < 8; ++e) L.part[half][hp * 64 + chunk * 8 + e] = acc[e];
            }
            __syncthreads();
            if (tid < D) L.hb[tid] = (bf16)((L.part[0][tid] + L.part[1][tid]) / L.red[16 + (tid >> 5)]);
            __syncthreads();
        }
        DT_STAMP(sb + 7);
        gemv<256, EPI_RES>((const bf16*)W.w_o2, W.b_o2, D, L, tid);
        // ================= feed-forward block: x += W2 relu(W1 LN3(x) + b1) + b2 =================
        if (li + 1 < d.nlayers) { g1 = d.layer[li + 1].ln1_g[pt]; b1 = d.layer[li + 1].ln1_b[pt]; }   // (the next layer's first LayerNorm)
        layer_norm(g3, b3, d.eps, L, tid);
        DT_STAMP(sb + 8);
        gemv<256, EPI_RELU>((const bf16*)W.w_f1, W.b_f1, FF, L, tid);
        DT_STAMP(sb + 9);
        DT_STAMP(sb + 10);
        gemv<1024, EPI_RES>((const bf16*)W.w_f2, W.b_f2, D, L, tid);
        DT_STAMP(sb + 11);
    }
    DT_STAMP(1 + 12 * RALF_DECODE_TOKEN_MAX_LAYERS);
    // ---- head: logits = Wh LN(x) (fp32, no bias) ----
    layer_norm(gh, bh, d.eps, L, tid);
    gemv<256, EPI_LOGITS>((const bf16*)d.w_head, nullptr, d.V, L, tid, nullptr, d.logits + (int64_t)b * d.V);
    // ---- optional: the token choice of ralf_mask_sample_step on this row, by one wave, from the logits in LDS (the same function on the same values) ----
    if (d.s_out && wv == 0) {
        auto emit = [&](int64_t tok) {
            d.s_out[b] = tok;
            if (d.s_seq_out) d.s_seq_out[(int64_t)b * d.s_seq_ld] = tok;
            if (d.s_flag_out) d.s_flag_out[(int64_t)b * d.s_flag_ld] = tok == d.s_pad_id ? 1 : 0;
        };
        sample_core::mask_sample_row(L.y, d.s_allowed, d.s_forced ? d.s_forced[b] : -1, d.s_mode, d.s_top_k, d.s_temperature, d.s_top_p, d.s_seed, d.s_call,
                                     (uint64_t)b + (uint64_t)d.s_row0, d.V, lane, emit);
    }
}
}  // namespace

extern "C" int ralf_decode_token(const RalfDecodeTokenDesc* dp, void* stream) {
    RALF_REQUIRE(dp, "decode_token: null descriptor");
    const RalfDecodeTokenDesc& d = *dp;
    RALF_REQUIRE(d.B > 0 && d.nlayers > 0 && d.nlayers <= RALF_DECODE_TOKEN_MAX_LAYERS && d.V > 0, "decode_token: B=%d layers=%d V=%d", d.B, d.nlayers, d.V);
    RALF_REQUIRE(d.L > 0 && d.L <= DT_MAXL && d.M > 0 && d.M <= DT_MAXM, "decode_token: L=%d (<= %d cached positions), M=%d (<= %d memory rows)", d.L, DT_MAXL, d.M, DT_MAXM);
    RALF_REQUIRE(d.pos_vec || (d.pos >= 0 && d.pos < d.L), "decode_token: pos=%d outside the cache of %d rows", d.pos, d.L);
    RALF_REQUIRE(d.tok && d.emb && d.pe && d.lnh_g && d.lnh_b && d.w_head && d.logits, "decode_token: null pointer");
    RALF_REQUIRE(((uintptr_t)d.w_head & 15) == 0, "decode_token: the head matrix must be 16-byte aligned");
    for (int i = 0; i < d.nlayers; ++i) {
        const RalfDecodeTokenLayer& w = d.layer[i];
        RALF_REQUIRE(w.w_qkv && w.b_qkv && w.ln1_g && w.ln1_b && w.w_o1 && w.b_o1 && w.ln2_g && w.ln2_b && w.w_q2 && w.b_q2 && w.w_o2 && w.b_o2 && w.ln3_g && w.ln3_b &&
                     w.w_f1 && w.b_f1 && w.w_f2 && w.b_f2 && w.self_kv && w.cross_kv, "decode_token: layer %d: null pointer", i);
        RALF_REQUIRE((((uintptr_t)w.w_qkv | (uintptr_t)w.w_o1 | (uintptr_t)w.w_q2 | (uintptr_t)w.w_o2 | (uintptr_t)w.w_f1 | (uintptr_t)w.w_f2 | (uintptr_t)w.self_kv | (uintptr_t)w.cross_kv) & 15) == 0,
                     "decode_token: layer %d: weights and caches must be 16-byte aligned", i);
    }
    if (d.s_out) {
        RALF_REQUIRE(d.V <= DT_MAXM, "decode_token: the in-kernel token choice keeps the row's logits in LDS (V = %d <= %d)", d.V, DT_MAXM);
        RALF_REQUIRE(d.s_mode >= 0 && d.s_mode <= 4 && (d.s_mode == 0 || (d.s_seed && d.s_temperature > 0.f)) && (d.s_mode != 1 || d.s_top_k >= 1) &&
                     (d.s_mode != 2 || (d.s_top_p > 0.f && d.s_top_p <= 1.f)) && (!d.s_seq_out || d.s_seq_ld > 0) && (!d.s_flag_out || d.s_flag_ld > 0),
                     "decode_token: token choice arguments (ralf_mask_sample_step's rules)");
    }
    hipLaunchKernelGGL(decode_token_kernel, dim3(d.B), dim3(NT), 0, (hipStream_t)stream, d);
    return ralf::check_launch("decode_token");
}

extern "C" int ralf_decode_token_limits(int* max_self_rows, int* max_memory_rows) {
    if (max_self_rows) *max_self_rows = DT_MAXL;
    if (max_memory_rows) *max_memory_rows = DT_MAXM;
    return RALF_OK;
}
