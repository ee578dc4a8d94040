// Exact inner-product top-k scan for gfx950 (MI355X).
//
// Replaces faiss.IndexFlat(d, METRIC_INNER_PRODUCT).search as reached from
// image2layout/train/models/retrieval/retriever.py:200-202 (one query per call in the reference;
// batched here).  Two phases:
//   1. knn_scores_kernel : S[q][n] = <Q[q], X[n]> on the fp32 matrix cores
//      (v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32).  A chain of K-steps on one accumulator is
//      bit-for-bit the ascending-d fmaf chain of oracle/knn_oracle.c, so scores are bit-exact.
//      The index is streamed from HBM exactly once per query tile; X/Q k-tiles are staged
//      global -> registers -> LDS (rotated rows: conflict-free ds_read_b32 fragment reads).
//   2. selection: knn_select_kernel per 8192-score segment (lower bound of the k-th best from the per-thread maxima, the few
//      survivors ranked exactly by counting; exact bisection over the 32 order-preserving key bits + 13 position bits for
//      mass ties).  k <= 64 and <= 64 segments (every shipped index): the segment lists of a query are merged by the last
//      workgroup of that query to finish (agent-scope release / ticket / acquire), i.e. ONE launch for the whole selection;
//      otherwise merge rounds with the same kernel.  Result order (score desc, index asc), bit-identical on all paths.
//    Measured and dropped (MI355X, 61548 x 1792, k = 16): (c, round 3) the scan on a direct-to-LDS ring (global_load_lds_dwordx4 into 2-4
//    stages, XOR-swizzled 16-byte slots, ds_read_b128 / ds_read_b32 fragment reads, one barrier per k-tile, counted vmcnt; scores
//    bit-identical): nq <= 16 81-83 us vs 79-82, nq = 17-32 100-104 us vs 92 -- the register-staged scan's rotated rows give it
//    conflict-free 4-byte fragment reads, which a lane-linear LDS-DMA image cannot (16-byte granules: two rows per bank pair, or a
//    select per operand), and at <= 16 queries the stream is already at 0.67-0.70 of peak;
//    (a) the selection fused into the scan's epilogue -- per-workgroup
//    top-k of the [queries x 128..256 rows] tile in LDS, no score matrix: a wave-level exact top-16 of 128 is thousands of
//    dependent scalar/vector round trips per query, the scan got 14-23 us slower at nq = 16-32 and 39 % slower at nq = 1024;
//    (b) one 1024-thread workgroup per query over the whole score row: a single CU pulls ~25 GB/s, 17-19 us per launch vs
//    16 us for the two launches it replaced;
//    (d, round 4) two k-tiles requested back to back (256 contiguous bytes of every row within a few hundred ns, the second tile waiting in a
//    second register set) and streaming (non-temporal) index loads: scan 76.9 vs 76.8 us at nq = 1, 107 vs 101 us at nq = 32; 73.4 vs 69.4 us.
#include <cstring>

#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MF> struct Frag;
template <> struct Frag<32> {
    using acc_t = f32x16;
    static constexpr int NREG = 16, KS = 2;
    static __device__ __forceinline__ acc_t mfma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
    // C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    static __device__ __forceinline__ int crow(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
};
template <> struct Frag<16> {
    using acc_t = f32x4;
    static constexpr int NREG = 4, KS = 4;
    static __device__ __forceinline__ acc_t mfma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    // C/D layout: col = lane&15, row = (lane>>4)*4 + r
    static __device__ __forceinline__ int crow(int r, int lane) { return (lane >> 4) * 4 + r; }
};

// Workgroups that share a row chunk (different query tiles) get consecutive virtual ids on ONE XCD
// (block b runs on XCD b%8) so the chunk is fetched from HBM once and re-read from that XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// order-preserving float -> uint (larger float = larger key); key 0 is reserved for "no entry"
__device__ __forceinline__ uint32_t f2key(float v) {
    const uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(uint32_t k) {
    return (k == 0u) ? -__builtin_inff() : __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
__device__ __forceinline__ bool better(uint32_t ka, int pa, uint32_t kb, int pb) { return ka > kb || (ka == kb && pa < pb); }

// A operand = query tile (M dim), B operand = index-row tile (N dim): for a fixed accumulator
// register the 32 (16) lanes of a half-wave hold CONSECUTIVE index rows of one query, so the
// score stores are contiguous 128 B (64 B) segments of S[q][*].
// GATHER (exact re-scoring of per-query candidate lists, two-stage search): workgroup <-> (query q, chunk of its `pool`
// candidates); the index rows come from cand[q][*], the query tile holds the single query q, S is [nq][pool].  The
// accumulation code is the one of the exhaustive scan, so a re-scored pair is bit-identical to its exhaustive score.
template <int MF, int TQ, int TR, bool GATHER = false>
__global__ __launch_bounds__(256) void knn_scores_kernel(const float* __restrict__ X, int64_t N, int D,
                                                          const float* __restrict__ Q, int nq, float* __restrict__ S,
                                                          int n_qtiles, int nwg, const int64_t* __restrict__ cand = nullptr, int pool = 0,
                                                          unsigned int* __restrict__ zero_me = nullptr, int nzero = 0) {
    using F = Frag<MF>;
    // the hand-off counters of the selection that FOLLOWS this scan are cleared here (stream order makes the zeros visible to it):
    // no memset node in front of the scan
    if (zero_me && blockIdx.x == 0) for (int i = threadIdx.x; i < nzero; i += 256) zero_me[i] = 0u;
    constexpr int RW = 4 * TR * MF, QW = TQ * MF, BK = 32, KS = F::KS, ROT = 32 / MF;
    constexpr int XV = RW * BK / 4 / 256;              // float4 per thread per k-tile (index rows)
    constexpr int QV = (QW * BK / 4 + 255) / 256;      // float4 per thread per k-tile (queries)
    __shared__ float lds[(RW + QW) * BK];
    float* lx = lds;
    float* lq = lds + RW * BK;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int vid = xcd_remap(blockIdx.x, nwg);
    const int qt = vid % n_qtiles, rc = vid / n_qtiles;
    const int64_t row0 = (int64_t)rc * RW;
    const int q0 = GATHER ? qt : qt * QW;                  // GATHER: qt is the query itself (n_qtiles = nq)
    const int64_t nrows = GATHER ? pool : N;               // rows addressable by this workgroup's list
    const int nq_hi = GATHER ? q0 + 1 : nq;                // GATHER: only row 0 of the query tile is real

    float4 xr[XV], qr[QV];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            const int f = tid + 256 * i, r = f >> 3, kq = f & 7;
            const int64_t n = row0 + r;
            const int k = k0 + kq * 4;
            const int64_t src = (GATHER && n < nrows) ? cand[(int64_t)q0 * pool + n] : n;
            xr[i] = (n < nrows && k < D) ? *reinterpret_cast<const float4*>(X + src * D + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < QV; ++i) {
            const int f = tid + 256 * i, r = f >> 3, kq = f & 7;
            const int qi = q0 + r, k = k0 + kq * 4;
            qr[i] = (f < QW * 8 && qi < nq_hi && k < D) ? *reinterpret_cast<const float4*>(Q + (int64_t)qi * D + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    // element (row, k) lives at row*32 + ((k + ROT*row) & 31): both the 4 scalar writes of a float4
    // and the per-lane fragment reads (32 or 16 rows x same k) hit distinct banks.
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            const int f = tid + 256 * i, r = f >> 3, kq = f & 7;
            float* base = lx + r * BK;
            const int rot = kq * 4 + ROT * r;
            base[(rot + 0) & 31] = xr[i].x; base[(rot + 1) & 31] = xr[i].y;
            base[(rot + 2) & 31] = xr[i].z; base[(rot + 3) & 31] = xr[i].w;
        }
#pragma unroll
        for (int i = 0; i < QV; ++i) {
            const int f = tid + 256 * i, r = f >> 3, kq = f & 7;
            if (f < QW * 8) {
                float* base = lq + r * BK;
                const int rot = kq * 4 + ROT * r;
                base[(rot + 0) & 31] = qr[i].x; base[(rot + 1) & 31] = qr[i].y;
                base[(rot + 2) & 31] = qr[i].z; base[(rot + 3) & 31] = qr[i].w;
            }
        }
    };

    typename F::acc_t acc[TQ][TR];
#pragma unroll
    for (int a = 0; a < TQ; ++a)
#pragma unroll
        for (int b = 0; b < TR; ++b)
#pragma unroll
            for (int r = 0; r < F::NREG; ++r) acc[a][b][r] = 0.f;

    const int lr = lane & (MF - 1), lk = lane / MF;  // row within fragment, k within K-step
    int qoff[TQ], qrot[TQ], xoff[TR], xrot[TR];
#pragma unroll
    for (int a = 0; a < TQ; ++a) { const int r = a * MF + lr; qoff[a] = r * BK; qrot[a] = ROT * r + lk; }
#pragma unroll
    for (int b = 0; b < TR; ++b) { const int r = (wave * TR + b) * MF + lr; xoff[b] = r * BK; xrot[b] = ROT * r + lk; }

    const int nkt = (D + BK - 1) / BK;
    gload(0);
    lstore();
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) gload((kt + 1) * BK);  // next tile's HBM loads fly under this tile's MFMAs
#pragma unroll
        for (int s = 0; s < BK / KS; ++s) {
            float a[TQ], b[TR];
#pragma unroll
            for (int i = 0; i < TQ; ++i) a[i] = lq[qoff[i] + ((qrot[i] + s * KS) & 31)];
#pragma unroll
            for (int j = 0; j < TR; ++j) b[j] = lx[xoff[j] + ((xrot[j] + s * KS) & 31)];
#pragma unroll
            for (int i = 0; i < TQ; ++i)
#pragma unroll
                for (int j = 0; j < TR; ++j) acc[i][j] = F::mfma(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
        if (kt + 1 < nkt) {
            lstore();
            __syncthreads();
        }
    }

#pragma unroll
    for (int i = 0; i < TQ; ++i)
#pragma unroll
        for (int j = 0; j < TR; ++j) {
            const int64_t n = row0 + (wave * TR + j) * MF + lr;
#pragma unroll
            for (int r = 0; r < F::NREG; ++r) {
                const int qi = q0 + i * MF + F::crow(r, lane);
                if (qi < nq_hi && n < nrows) S[(int64_t)qi * nrows + n] = acc[i][j][r];
            }
        }
}

// ------------------------------------------------------------------------------------------
// selection
// ------------------------------------------------------------------------------------------
constexpr int SEG = 8192;       // scores handled by one workgroup
constexpr int EPT = SEG / 256;  // keys per thread (registers)
constexpr int KMAX = 1024;


__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// sum over the 256-thread block; `slot` alternates so one barrier per call suffices
__device__ __forceinline__ int block_sum(int v, int* red, int& slot) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[slot * 4 + (threadIdx.x >> 6)] = v;
    __syncthreads();
    const int s = red[slot * 4] + red[slot * 4 + 1] + red[slot * 4 + 2] + red[slot * 4 + 3];
    slot ^= 1;
    return s;
}

// Winners of one list held in registers: key[i] of thread t is the entry at position t + 256*i (0 = no entry).
// On return wkey/wpos[0..nw) hold every entry that can be in the top-k (exactly min(k, #valid) of them on the bisection
// path, a few more on the fast path); the caller ranks them.  Ends with a barrier.
//
// Fast path: the k-th largest of the 256 per-thread maxima is a lower bound L of the k-th largest
// key, so only keys >= L (a few more than k on non-degenerate data) are candidates; they are ranked
// exactly by counting.  Degenerate inputs (more than KMAX candidates: massive ties) or k > 256 take
// the exact bisection over all SEG keys instead.  Both paths give identical results.
struct SelLds {
    int red[8];
    uint32_t wkey[KMAX];
    int wpos[KMAX];
    uint32_t tmax[256];
    int wcnt;
    uint32_t bound;
    int last;
};
__device__ __forceinline__ int select_winners(const uint32_t (&key)[EPT], int k, SelLds& L_) {
    const int tid = threadIdx.x, lane = tid & 63;
    uint32_t mx = 0;
#pragma unroll
    for (int i = 0; i < EPT; ++i) mx = key[i] > mx ? key[i] : mx;
    L_.tmax[tid] = mx;
    if (tid == 0) L_.wcnt = 0;
    __syncthreads();
    if (tid < 64) {  // wave 0: L = k-th largest thread maximum (0 if fewer than k non-empty threads)
        uint32_t L = 0;
        if (k <= 256) {
            const uint32_t m0 = L_.tmax[lane], m1 = L_.tmax[lane + 64], m2 = L_.tmax[lane + 128], m3 = L_.tmax[lane + 192];
            for (int bit = 31; bit >= 0; --bit) {
                const uint32_t t = L | (1u << bit);
                const int c = __popcll(__ballot(m0 >= t)) + __popcll(__ballot(m1 >= t)) + __popcll(__ballot(m2 >= t)) + __popcll(__ballot(m3 >= t));
                if (c >= k) L = t;
            }
        }
        if (lane == 0) L_.bound = L;
    }
    __syncthreads();
    const uint32_t L = L_.bound;
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        if (key[i] >= L && key[i] != 0u) {
            const int s = atomicAdd(&L_.wcnt, 1);
            if (s < KMAX) { L_.wkey[s] = key[i]; L_.wpos[s] = tid + 256 * i; }
        }
    }
    __syncthreads();
    int nw = L_.wcnt;
    if (nw > KMAX) {
        // ---- exact bisection over all keys: largest T with |{key >= T}| >= k ----
        __syncthreads();
        if (tid == 0) L_.wcnt = 0;
        int slot = 0;
        uint32_t T = 0;
        for (int bit = 31; bit >= 0; --bit) {
            const uint32_t t = T | (1u << bit);
            int c = 0;
#pragma unroll
            for (int i = 0; i < EPT; ++i) c += (key[i] >= t);
            if (block_sum(c, L_.red, slot) >= k) T = t;
        }
        int cgt = 0, ceq = 0;
#pragma unroll
        for (int i = 0; i < EPT; ++i) { cgt += (key[i] > T); ceq += (key[i] == T); }
        cgt = block_sum(cgt, L_.red, slot);
        ceq = block_sum(ceq, L_.red, slot);
        const int need = k - cgt;  // entries equal to T to keep, lowest positions first
        int P = SEG;               // keep key == T entries with position <= P
        if (need < ceq) {          // smallest P with |{key == T, pos <= P}| >= need (13 position bits)
            int lo = -1;
            for (int bit = 12; bit >= 0; --bit) {
                const int t = lo + (1 << bit);
                int c = 0;
#pragma unroll
                for (int i = 0; i < EPT; ++i) c += (key[i] == T && (tid + 256 * i) <= t);
                if (block_sum(c, L_.red, slot) < need) lo = t;
            }
            P = lo + 1;
        }
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
            const int p = tid + 256 * i;
            if (key[i] != 0u && (key[i] > T || (key[i] == T && p <= P))) {
                const int s = atomicAdd(&L_.wcnt, 1);
                L_.wkey[s] = key[i];
                L_.wpos[s] = p;
            }
        }
        __syncthreads();
        nw = L_.wcnt;  // == min(k, #valid) <= KMAX
    }
    return nw;
}

// One workgroup selects the top-k of `cnt` (<= SEG) entries of query blockIdx.y, list blockIdx.x.
//   FROM_SCORES: entries are S[q][base + p], index = base + p
//   else       : entries are (cs, ci)[q][base + p] candidate lists; equal scores appear in ascending
//                index order, so "lower position wins" == "lower index wins" in both modes.
// Output list (sorted by score desc, index asc; padded with (-inf,-1)): os/oi[q][blockIdx.x][k].
//
// FUSED_MERGE (FROM_SCORES, k <= 64, <= 64 lists): the segment lists of a query are merged by the LAST workgroup of that
// query to finish -- one launch for the whole selection.  Hand-off: every wave drains its list stores, the workgroup's
// lane 0 issues an agent-scope release, then takes a ticket on the query's counter (zeroed by a memset node ahead of the
// launch); the workgroup that draws the last ticket issues one agent-scope acquire and reads the lists with plain loads.
template <bool FROM_SCORES, bool FUSED_MERGE = false>
__global__ __launch_bounds__(256) void knn_select_kernel(const float* __restrict__ S, const float* __restrict__ cs,
                                                          const int64_t* __restrict__ ci, int64_t in_stride, int64_t in_count,
                                                          int64_t per_wg, int k, float* __restrict__ os, int64_t* __restrict__ oi,
                                                          unsigned int* __restrict__ tickets = nullptr, float* __restrict__ fs = nullptr,
                                                          int64_t* __restrict__ fi = nullptr) {
    __shared__ SelLds L_;
    const int tid = threadIdx.x;
    const int q = blockIdx.y, lst = blockIdx.x, nlst = gridDim.x;
    const int64_t base = (int64_t)lst * per_wg;
    int64_t rem = in_count - base;
    const int cnt = (int)(rem < per_wg ? rem : per_wg);
    const float* src = (FROM_SCORES ? S : cs) + (int64_t)q * in_stride + base;
    float* so = os + ((int64_t)q * nlst + lst) * k;
    int64_t* io = oi + ((int64_t)q * nlst + lst) * k;

    // (loads are UNCONDITIONAL on a clamped position: a load under `p < cnt ? .. : ..` makes hipcc branch around every one of the
    //  32 loads and wait for it -- 32 dependent memory round trips, 10 of the kernel's 15 us)
    uint32_t key[EPT];
    float raw[EPT];
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        const int p = tid + 256 * i;  // position (coalesced loads)
        raw[i] = src[p < cnt ? p : cnt - 1];
    }
#pragma unroll
    for (int i = 0; i < EPT; ++i) key[i] = (tid + 256 * i < cnt) ? f2key(raw[i]) : 0u;
    int nw = select_winners(key, k, L_);
    // rank the nw candidates exactly; the best min(nw,k) are the output
    for (int i = tid; i < nw; i += 256) {
        const uint32_t ki = L_.wkey[i];
        const int pi = L_.wpos[i];
        int rank = 0;
        for (int j = 0; j < nw; ++j) rank += better(L_.wkey[j], L_.wpos[j], ki, pi);
        if (rank < k) {
            so[rank] = key2f(ki);
            io[rank] = FROM_SCORES ? (base + pi) : ci[(int64_t)q * in_stride + base + pi];
        }
    }
    for (int i = (nw < k ? nw : k) + tid; i < k; i += 256) {
        so[i] = -__builtin_inff();
        io[i] = -1;
    }
    if constexpr (FUSED_MERGE) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave: its list stores have left
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned int t = __hip_atomic_fetch_add(tickets + q, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = (t == (unsigned int)(nlst - 1));
            if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            L_.last = last;
        }
        __syncthreads();
        if (!L_.last) return;
        // ---- merge the nlst lists of this query: nlst * k <= 4096 candidates, positions list-major ----
        const int total = nlst * k;
        const float* mcs = os + (int64_t)q * total;
        const int64_t* mci = oi + (int64_t)q * total;
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
            const int p = tid + 256 * i;
            raw[i] = mcs[p < total ? p : total - 1];
        }
#pragma unroll
        for (int i = 0; i < EPT; ++i) key[i] = (tid + 256 * i < total) ? f2key(raw[i]) : 0u;
        nw = select_winners(key, k, L_);
        for (int i = tid; i < nw; i += 256) {
            const uint32_t ki = L_.wkey[i];
            const int pi = L_.wpos[i];
            int rank = 0;
            for (int j = 0; j < nw; ++j) rank += better(L_.wkey[j], L_.wpos[j], ki, pi);
            if (rank < k) {
                fs[(int64_t)q * k + rank] = key2f(ki);
                fi[(int64_t)q * k + rank] = mci[pi];
            }
        }
        for (int i = (nw < k ? nw : k) + tid; i < k; i += 256) {
            fs[(int64_t)q * k + i] = -__builtin_inff();
            fi[(int64_t)q * k + i] = -1;
        }
    }
}

// ---- two-stage search helpers ------------------------------------------------------------------------------------
// final selection among per-query candidates: exact[q][j] is the score of index row cand[q][j]; the k best by
// (score desc, row index asc) -- the order of the exhaustive scan -- plus the per-query certificate of the two-stage search:
//   bad[q] = !(k-th exact score >= bound[q] + eps[q]),  eps = |q - qb| xmax + |qb| dxmax + 2 D 2^-24 |q| xmax + 1e-6
// (qn = per-query norms {|q|, |qb|, |q - qb|}, xn = index constants {max|x|, max|xb|, max|x - xb|}; see retrieval/knn.py).
// One wave per query; pool <= 1024.  (`over`: see knn_select_lists_kernel.)
__global__ __launch_bounds__(64) void knn_select_cand_kernel(const float* __restrict__ exact, const int64_t* __restrict__ cand, int pool, int k,
                                                             float* __restrict__ os, int64_t* __restrict__ oi, const float* __restrict__ bound,
                                                             int64_t bound_ld, const float* __restrict__ qn, const float* __restrict__ xn, int D,
                                                             int32_t* __restrict__ bad, const int32_t* __restrict__ over = nullptr) {
    __shared__ uint32_t key[KMAX];
    __shared__ int64_t row[KMAX];
    const int q = blockIdx.x, lane = threadIdx.x;
    for (int j = lane; j < pool; j += 64) {
        key[j] = f2key(exact[(int64_t)q * pool + j]);
        row[j] = cand[(int64_t)q * pool + j];
    }
    __syncthreads();
    float kth = -__builtin_inff();
    for (int j = lane; j < pool; j += 64) {
        const uint32_t kj = key[j];
        const int64_t rj = row[j];
        int rank = 0;
#pragma unroll 16
        for (int i = 0; i < pool; ++i) rank += (key[i] > kj) || (key[i] == kj && row[i] < rj);   // (unrolled: the LDS reads of 16 iterations go out together)
        if (rank < k) {
            os[(int64_t)q * k + rank] = key2f(kj);
            oi[(int64_t)q * k + rank] = rj;
            if (rank == k - 1) kth = key2f(kj);
        }
    }
    if (bad) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) kth = fmaxf(kth, __shfl_xor(kth, o));
        if (lane == 0) {
            const float eps = qn[q * 3 + 2] * xn[0] + qn[q * 3 + 1] * xn[2] + 2.f * (float)D * 5.9604645e-8f * qn[q * 3] * xn[0] + 1e-6f;
            // (a -inf bound = the candidate list was padded: fewer than `pool` rows reached a filtered coarse pass's threshold; over = a tile of that pass lost hits)
            const float b = bound[(int64_t)q * bound_ld];
            bad[q] = !(kth >= b + eps * 1.001f) || b == -__builtin_inff() || (over && over[q]);
        }
    }
}

// per row r of X [R, D] (fp32) and its bf16 rounding Xb: norms[r] = {|x_r|, |xb_r|, |x_r - xb_r|} (may be NULL) and
// maxes[0..2] = max over rows (may be NULL; non-negative floats compare like their bit patterns; caller zeroes it).
__global__ __launch_bounds__(256) void knn_rownorms_kernel(const float* __restrict__ X, const __bf16* __restrict__ Xb, int64_t R, int D,
                                                          float* __restrict__ norms, float* __restrict__ maxes) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    float a = 0.f, b = 0.f, c = 0.f;
    for (int d = lane; d < D; d += 64) {
        const float x = X[r * D + d], xb = (float)Xb[r * D + d], e = x - xb;
        a += x * x; b += xb * xb; c += e * e;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); c += __shfl_xor(c, o); }
    if (lane == 0) {
        a = sqrtf(a) * 1.000001f; b = sqrtf(b) * 1.000001f; c = sqrtf(c) * 1.000001f;   // rounded up: these feed an upper bound
        if (norms) { norms[r * 3] = a; norms[r * 3 + 1] = b; norms[r * 3 + 2] = c; }
        if (maxes) {
            atomicMax(reinterpret_cast<unsigned int*>(maxes), __float_as_uint(a));
            atomicMax(reinterpret_cast<unsigned int*>(maxes) + 1, __float_as_uint(b));
            atomicMax(reinterpret_cast<unsigned int*>(maxes) + 2, __float_as_uint(c));
        }
    }
}


// the queries of the two-stage search in one pass: Qb = bf16(Q) (round to nearest even, as ralf_copy2d) and norms[q] = {|q|, |qb|, |q - qb|} rounded up.
// One wave per query, 16 bytes per lane and step.  D % 4 == 0.
__global__ __launch_bounds__(256) void knn_query_prep_kernel(const float* __restrict__ Q, __bf16* __restrict__ Qb, int nq, int D, float* __restrict__ norms,
                                                             int32_t* __restrict__ zero = nullptr) {
    const int lane = threadIdx.x & 63;
    const int r = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (r >= nq) return;
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    float a = 0.f, b = 0.f, c = 0.f;
    for (int d = 4 * lane; d < D; d += 256) {
        const float4 x = *reinterpret_cast<const float4*>(Q + (int64_t)r * D + d);
        const float xs[4] = {x.x, x.y, x.z, x.w};
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o[e] = (__bf16)xs[e];
            const float xb = (float)o[e], err = xs[e] - xb;
            a += xs[e] * xs[e]; b += xb * xb; c += err * err;
        }
        *reinterpret_cast<bf16x4*>(Qb + (int64_t)r * D + d) = o;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); c += __shfl_xor(c, o); }
    if (lane == 0) {
        norms[r * 3] = sqrtf(a) * 1.000001f; norms[r * 3 + 1] = sqrtf(b) * 1.000001f; norms[r * 3 + 2] = sqrtf(c) * 1.000001f;   // rounded up: these feed an upper bound
        if (zero) zero[r] = 0;   // (the filtered search's per-query "a tile lost hits" flag, set by knn_select_lists_kernel further down the stream)
    }
}

// Selection among the candidate slots a FILTERED coarse pass wrote (RalfGemmDesc.flt_*; round 6: the unpack -> dense selection -> row gather chain of the
// Python path in one kernel).  Slot p of query q = list[q * W + p] = {row, score bits}, W = T << capsh, in use when (p & (cap - 1)) < count[q * T + (p >> capsh)].
// One workgroup per (segment of SEG slots = blockIdx.x, query = blockIdx.y): the segment's k best by (score desc, then a fixed order of the slots) -> os / oi[q][segment][k], padded
// with (-inf, row 0): a padded entry must name a row the re-score can read, and knn_select_cand_kernel refuses to certify against a -inf bound.  (The order of
// equal coarse scores is the tile's arrival order, not the row order of the dense selection: the candidate SET may differ among ties at the pool's edge, the
// certificate -- every row outside the candidates scores <= the (pool+1)-th coarse score -- holds for either.)  over[q] |= a tile of the query lost hits.
__global__ __launch_bounds__(256) void knn_select_lists_kernel(const int2* __restrict__ list, const int* __restrict__ count, int T, int capsh, int k,
                                                               float* __restrict__ os, int64_t* __restrict__ oi, int32_t* __restrict__ over) {
    __shared__ SelLds L_;
    const int tid = threadIdx.x, q = blockIdx.y, seg = blockIdx.x, nseg = gridDim.x;
    const int cap = 1 << capsh, W = T << capsh, base = seg * SEG;
    const int2* src = list + (int64_t)q * W;
    const int* cq = count + (int64_t)q * T;
    float* so = os + ((int64_t)q * nseg + seg) * k;
    int64_t* io = oi + ((int64_t)q * nseg + seg) * k;
    // Entry i of thread t is slot 256 i + ((t + i) & 255): the slots of a tile fill from position 0 and hold ~2 hits, so with entry i = slot 256 i + t the
    // threads with (t & 15) >= 3 would hold no key at all -- fewer non-empty threads than k, select_winners' bound drops to 0 and all ~1000 hits of the query
    // are ranked against each other (84 us at 1024 queries; 14 with the rotation).  slot_of() maps select_winners' position t + 256 i back.
    auto slot_of = [](int pos) { return (pos & ~255) | ((pos + (pos >> 8)) & 255); };
    uint32_t key[EPT];
    int sc[EPT], c[EPT];
#pragma unroll
    for (int i = 0; i < EPT; ++i) {   // (unconditional loads on clamped positions, as knn_select_kernel)
        const int p = base + 256 * i + ((tid + i) & 255), pc = p < W ? p : W - 1;
        sc[i] = src[pc].y;
        c[i] = cq[pc >> capsh];
    }
    int lost = 0;
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        const int p = base + 256 * i + ((tid + i) & 255);
        key[i] = (p < W && (p & (cap - 1)) < c[i]) ? f2key(__int_as_float(sc[i])) : 0u;
        lost |= (p < W && c[i] > cap) ? 1 : 0;
    }
    const int nw = select_winners(key, k, L_);
    for (int i = tid; i < nw; i += 256) {
        const uint32_t ki = L_.wkey[i];
        const int pi = L_.wpos[i];
        int rank = 0;
        for (int j = 0; j < nw; ++j) rank += better(L_.wkey[j], L_.wpos[j], ki, pi);
        if (rank < k) {
            so[rank] = key2f(ki);
            io[rank] = (int64_t)src[base + slot_of(pi)].x;
        }
    }
    for (int i = (nw < k ? nw : k) + tid; i < k; i += 256) {
        so[i] = -__builtin_inff();
        io[i] = 0;
    }
    if (lost) atomicOr(over + q, 1);
}

template <int MF, int TQ, int TR>
int launch_scores(const float* X, int64_t N, int D, const float* Q, int nq, float* S, hipStream_t st, unsigned int* zero_me = nullptr, int nzero = 0) {
    constexpr int RW = 4 * TR * MF, QW = TQ * MF;
    const int nrc = ceil_div(N, RW), nqt = ceil_div(nq, QW);
    const int nwg = nrc * nqt;
    hipLaunchKernelGGL((knn_scores_kernel<MF, TQ, TR>), dim3(nwg), dim3(256), 0, st, X, N, D, Q, nq, S, nqt, nwg, nullptr, 0, zero_me, nzero);
    return ralf::check_launch("knn_scores");
}

// Exact re-scoring, one WAVE per (query, 16 candidates) (round 6).  The gathered form of the scan kernel above walks D in 32-wide k-tiles with two
// barriers and one (random-row, HBM-latency) load round trip per tile: 50 us at any batch size, 126 us at nq = 1024.  Here a single-wave
// workgroup streams its 16 candidate rows through a ring of three 128-column chunks in LDS (global_load_lds_dwordx4, 8 per chunk, counted
// vmcnt, no barrier: one wave) and runs the D / 4 v_mfma_f32_16x16x4_f32 of the block behind them: A[m][k] = candidate m's element d + k,
// B[k][n] = the query's element d + k in every column n, so column 0 of the accumulator is <q, x_m> summed by the ascending-d fma chain of
// the scan (v_mfma_f32_16x16x4_f32 on one accumulator == the scan's chain, bit for bit: tests/test_knn_gpu.py).  ~30 KiB of LDS per wave:
// five such waves per CU, each bound by its own 40-cycle MFMA chain.  D % 128 == 0 (a whole number of chunks = of the scan's k-tiles).
#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))
constexpr int RS_CH = 128, RS_NB = 3, RS_PAIR = 2 * RS_CH + 4;   // chunk columns, ring depth, floats per row PAIR (rows 2p, 2p+1 adjacent: one 1-KiB DMA; +4: pairs start in different banks)
__global__ __launch_bounds__(64) void knn_rescore_wave_kernel(const float* __restrict__ X, int D, const float* __restrict__ Q, const int64_t* __restrict__ cand,
                                                              int pool, float* __restrict__ S) {
    extern __shared__ __attribute__((aligned(16))) float rl[];   // [RS_NB][8 pairs][RS_PAIR] | [D] the query
    const int lane = threadIdx.x, q = blockIdx.y, g = blockIdx.x;
    float* lq = rl + RS_NB * 8 * RS_PAIR;
    const int nc = D / RS_CH;
    // DMA j of a chunk: rows 2j, 2j+1; lane l moves 16 bytes: row 2j + (l >> 5), columns 4 (l & 31) ..
    const float* rp[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int r = 2 * j + (lane >> 5);
        rp[j] = X + cand[(int64_t)q * pool + min(g * 16 + r, pool - 1)] * D + 4 * (lane & 31);
    }
    auto issue = [&](int c) {
        float* dst = rl + (c % RS_NB) * 8 * RS_PAIR;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(rp[j] + c * RS_CH), LDS_PTR(void, dst + j * RS_PAIR), 16, 0, 0);
    };
    const int nqd = D / 256;                                       // 1-KiB pieces of the query (D % 256 may leave a 512-byte tail: moved by plain loads)
    for (int v = 0; v < nqd; ++v)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Q + (int64_t)q * D + 256 * v + 4 * lane), LDS_PTR(void, lq + 256 * v), 16, 0, 0);
    if (D % 256) { if (lane < 32) *reinterpret_cast<float4*>(lq + 256 * nqd + 4 * lane) = *reinterpret_cast<const float4*>(Q + (int64_t)q * D + 256 * nqd + 4 * lane); }
    issue(0);
    if (nc > 1) issue(1);
    if (nc > 2) issue(2);
    const int i = lane & 15, kq = lane >> 4;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int xoff = (i >> 1) * RS_PAIR + (i & 1) * RS_CH + kq;    // row i of a chunk image, column kq
    for (int c = 0; c < nc; ++c) {
        // chunk c has landed when at most the DMAs of the chunks behind it are outstanding (c + 1, c + 2: 8 each; fewer at the end)
        const int behind = min(nc - 1 - c, 2);
        if (behind == 2) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
        else if (behind == 1) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const float* xp = rl + (c % RS_NB) * 8 * RS_PAIR + xoff;
        const float* qp = lq + c * RS_CH + kq;
#pragma unroll
        for (int h = 0; h < RS_CH / 32; ++h) {
            float xa[8], qa[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { xa[u] = xp[32 * h + 4 * u]; qa[u] = qp[32 * h + 4 * u]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[u], qa[u], acc, 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // this chunk's reads are done before its ring slot is filled again
        if (c + RS_NB < nc) issue(c + RS_NB);
    }
    if (i == 0) {                                                // column 0: lane (0, kq) holds candidates 4 kq .. 4 kq + 3
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int cc = g * 16 + 4 * kq + r;
            if (cc < pool) S[(int64_t)q * pool + cc] = acc[r];
        }
    }
}

int launch_rescore(const float* X, int64_t N, int D, const float* Q, int nq, const int64_t* cand, int pool, float* S, hipStream_t st) {
    static const int wave_form = [] { const char* e = getenv("RALF_KNN_RESCORE_WAVE"); return e ? atoi(e) : 1; }();   // 0 = the k-tiled form (A/B runs, tests)
    const size_t lds = ((size_t)RS_NB * 8 * RS_PAIR + (size_t)D) * sizeof(float);
    if (wave_form && D % RS_CH == 0 && lds <= 64 * 1024 && nq <= 65535) {
        hipLaunchKernelGGL(knn_rescore_wave_kernel, dim3(ceil_div(pool, 16), nq), dim3(64), lds, st, X, D, Q, cand, pool, S);
        return ralf::check_launch("knn_rescore (one wave per 16 candidates)");
    }
    constexpr int MF = 16, RW = 4 * MF;                    // 64 candidates per workgroup, one query per 16-row query tile
    const int nrc = ceil_div(pool, RW);
    const int nwg = nrc * nq;
    hipLaunchKernelGGL((knn_scores_kernel<MF, 1, 1, true>), dim3(nwg), dim3(256), 0, st, X, N, D, Q, nq, S, nq, nwg, cand, pool);
    return ralf::check_launch("knn_rescore");
}

struct SelectPlan {
    int64_t nseg;      // lists after the score pass
    size_t cand_bytes; // one candidate buffer (scores + indices), sized for the first level
};
SelectPlan plan_select(int64_t N, int nq, int k) {
    SelectPlan p;
    p.nseg = (N + SEG - 1) / SEG;
    p.cand_bytes = (size_t)nq * p.nseg * k * (sizeof(float) + sizeof(int64_t));
    return p;
}
inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

static int scores_impl(const float* X, int64_t N, int D, const float* Q, int nq, float* S, hipStream_t st, unsigned int* zero_me, int nzero) {
    // HBM-bound regime (few queries): small row chunks -> >= 2 workgroups per CU in flight.
    if (nq <= 16) return launch_scores<16, 1, 2>(X, N, D, Q, nq, S, st, zero_me, nzero);
    if (nq <= 32) {
        static const int v32 = [] { const char* e = getenv("RALF_KNN_V32"); return e ? atoi(e) : 0; }();   // tuning aid
        if (v32 == 1) return launch_scores<32, 1, 2>(X, N, D, Q, nq, S, st, zero_me, nzero);
        if (v32 == 2) return launch_scores<16, 2, 2>(X, N, D, Q, nq, S, st, zero_me, nzero);
        if (v32 == 3) return launch_scores<16, 2, 1>(X, N, D, Q, nq, S, st, zero_me, nzero);
        return launch_scores<32, 1, 1>(X, N, D, Q, nq, S, st, zero_me, nzero);
    }
    if (nq <= 64) return launch_scores<32, 2, 2>(X, N, D, Q, nq, S, st, zero_me, nzero);
    // FLOP-bound regime: 256 rows x 128 queries per workgroup, 8 accumulator tiles per wave.
    return launch_scores<32, 4, 2>(X, N, D, Q, nq, S, st, zero_me, nzero);
}

// ---- candidate lists of the filtered coarse pass (RalfGemmDesc.flt_*) -> the dense form the selection kernels take ----
namespace {
// slots [nq][T][cap] {row, score bits} with count [nq][T] hits per column tile -> dense rows / scores [nq][T * cap] (unused slots: row 0, -inf)
__global__ __launch_bounds__(256) void knn_list_unpack_kernel(const int2* __restrict__ list, const int* __restrict__ count, int nq, int T, int cap,
                                                               int64_t* __restrict__ rows, float* __restrict__ scores, int* __restrict__ over) {
    const int q = blockIdx.y;
    const int64_t W = (int64_t)T * cap;
    int lost = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < W; i += (int64_t)gridDim.x * 256) {
        const int t = (int)(i / cap), p = (int)(i % cap);
        const int c = count[(int64_t)q * T + t];
        const bool ok = p < c;
        lost |= (p == 0 && c > cap) ? 1 : 0;
        const int2 e = list[(int64_t)q * W + (ok ? i : (int64_t)t * cap)];
        rows[(int64_t)q * W + i] = ok ? (int64_t)e.x : 0;                         // (an unused slot must still name a valid row)
        scores[(int64_t)q * W + i] = ok ? __int_as_float(e.y) : -INFINITY;
    }
    if (over && lost) atomicOr(over + q, 1);
}
__global__ __launch_bounds__(256) void knn_gather_rows_kernel(const int64_t* __restrict__ rows, int cap, const int64_t* __restrict__ pos, int nq, int m,
                                                               int64_t* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)nq * m) return;
    const int q = (int)(e / m);
    const int64_t p = pos[e];
    out[e] = rows[(int64_t)q * cap + (p >= 0 && p < cap ? p : 0)];
}
}  // namespace

extern "C" int ralf_knn_list_unpack(const int* list, const int* count, int nq, int T, int cap, int64_t* rows, float* scores, int* over, void* stream) {
    RALF_REQUIRE(list && count && rows && scores && nq > 0 && T > 0 && cap > 0, "knn_list_unpack: bad arguments");
    const int64_t W = (int64_t)T * cap;
    hipLaunchKernelGGL(knn_list_unpack_kernel, dim3((unsigned)std::min<int64_t>((W + 255) / 256, 64), nq), dim3(256), 0, (hipStream_t)stream, (const int2*)list, count, nq, T, cap, rows, scores, over);
    return ralf::check_launch("knn_list_unpack");
}
extern "C" int ralf_knn_gather_rows(const int64_t* rows, int cap, const int64_t* pos, int nq, int m, int64_t* out, void* stream) {
    RALF_REQUIRE(rows && pos && out && nq > 0 && m > 0 && cap > 0, "knn_gather_rows: bad arguments");
    hipLaunchKernelGGL(knn_gather_rows_kernel, dim3((unsigned)(((int64_t)nq * m + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rows, cap, pos, nq, m, out);
    return ralf::check_launch("knn_gather_rows");
}

extern "C" int ralf_knn_scores(const float* X, int64_t N, int D, const float* Q, int nq, float* S, void* stream) {
    RALF_REQUIRE(X && Q && S, "knn_scores: null pointer");
    RALF_REQUIRE(N > 0 && nq > 0 && D > 0, "knn_scores: empty problem (n_db=%lld nq=%d dim=%d)", (long long)N, nq, D);
    RALF_REQUIRE(D % 4 == 0, "knn_scores: dim %d must be a multiple of 4 (16-byte row alignment)", D);
    RALF_REQUIRE(((uintptr_t)X & 15) == 0 && ((uintptr_t)Q & 15) == 0, "knn_scores: index/queries must be 16-byte aligned");
    return scores_impl(X, N, D, Q, nq, S, (hipStream_t)stream, nullptr, 0);
}

/* exact fp32 scores of per-query candidate rows: out[q][j] = <Q[q], X[cand[q][j]]>, bit-identical to ralf_knn_scores */
extern "C" int ralf_knn_rescore(const float* X, int64_t N, int D, const float* Q, int nq, const int64_t* cand, int pool, float* out, void* stream) {
    RALF_REQUIRE(X && Q && cand && out && N > 0 && nq > 0 && pool > 0 && D > 0 && D % 4 == 0, "knn_rescore: bad arguments");
    RALF_REQUIRE(((uintptr_t)X & 15) == 0 && ((uintptr_t)Q & 15) == 0, "knn_rescore: index/queries must be 16-byte aligned");
    return launch_rescore(X, N, D, Q, nq, cand, pool, out, (hipStream_t)stream);
}

extern "C" int ralf_knn_select_cand(const float* exact, const int64_t* cand, int nq, int pool, int k, int64_t* out_idx, float* out_score,
                                    const float* bound, int64_t bound_ld, const float* qnorms, const float* xnorms, int D, int32_t* bad, void* stream) {
    RALF_REQUIRE(exact && cand && out_idx && out_score && nq > 0, "knn_select_cand: bad arguments");
    RALF_REQUIRE(pool >= 1 && pool <= KMAX && k >= 1 && k <= pool, "knn_select_cand: pool=%d k=%d outside [1,%d]", pool, k, KMAX);
    RALF_REQUIRE(!bad || (bound && qnorms && xnorms), "knn_select_cand: the certificate needs bound, qnorms and xnorms");
    hipLaunchKernelGGL(knn_select_cand_kernel, dim3(nq), dim3(64), 0, (hipStream_t)stream, exact, cand, pool, k, out_score, out_idx, bound, bound_ld,
                       qnorms, xnorms, D, bad, (const int32_t*)nullptr);
    return ralf::check_launch("knn_select_cand");
}

extern "C" int ralf_knn_rownorms(const float* X, const void* Xb, int64_t R, int D, float* norms, float* maxes, void* stream) {
    RALF_REQUIRE(X && Xb && R > 0 && D > 0 && (norms || maxes), "knn_rownorms: bad arguments");
    hipLaunchKernelGGL(knn_rownorms_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream, X, (const __bf16*)Xb, R, D, norms, maxes);
    return ralf::check_launch("knn_rownorms");
}

extern "C" size_t ralf_knn_topk_ip_workspace_bytes(int64_t N, int D, int nq, int k) {
    (void)D;
    if (N <= 0 || nq <= 0 || k <= 0) return 0;
    SelectPlan p = plan_select(N, nq, k);
    return align256((size_t)nq * N * sizeof(float)) + 2 * align256(p.cand_bytes) + align256((size_t)nq * sizeof(unsigned int)) + 256;
}

// the nl sorted k-entry lists per query in cs[0] / ci[0] -> one (SEG / k lists per workgroup and level, ping-pong between the two buffers)
static int merge_lists(float** cs, int64_t** ci, int64_t nl, int nq, int k, int64_t* out_idx, float* out_score, hipStream_t st) {
    int cur = 0;
    const int64_t group = SEG / k;  // lists merged per workgroup (>= 8 since k <= 1024)
    while (nl > 1) {
        const int64_t nout = (nl + group - 1) / group;
        float* so = nout == 1 ? out_score : cs[cur ^ 1];
        int64_t* io = nout == 1 ? out_idx : ci[cur ^ 1];
        hipLaunchKernelGGL((knn_select_kernel<false>), dim3((unsigned)nout, nq), dim3(256), 0, st, nullptr, cs[cur], ci[cur], nl * k, nl * k, group * k, k, so, io,
                           (unsigned int*)nullptr, (float*)nullptr, (int64_t*)nullptr);
        nl = nout;
        cur ^= 1;
    }
    return ralf::check_launch("knn_select");
}

// workspace for select alone = 2 candidate buffers (+ 256 bytes of ticket counters per 64 queries at its end)
static int select_impl(const float* S, int64_t N, int nq, int k, int64_t* out_idx, float* out_score, void* ws, size_t ws_bytes,
                       hipStream_t st, bool tickets_zeroed, bool one_launch_ok = true) {
    SelectPlan p = plan_select(N, nq, k);
    if (p.nseg == 1) {
        hipLaunchKernelGGL((knn_select_kernel<true>), dim3(1, nq), dim3(256), 0, st, S, nullptr, nullptr, N, N, (int64_t)SEG, k, out_score, out_idx,
                           (unsigned int*)nullptr, (float*)nullptr, (int64_t*)nullptr);
        return ralf::check_launch("knn_select");
    }
    const size_t tick_bytes = align256((size_t)nq * sizeof(unsigned int));
    if (!ws || ws_bytes < 2 * align256(p.cand_bytes) + tick_bytes) {
        ralf::set_error("knn_select: workspace %zu < required %zu bytes", ws_bytes, 2 * align256(p.cand_bytes) + tick_bytes);
        return RALF_ERR_WORKSPACE;
    }
    char* w = (char*)ws;
    float* cs[2];
    int64_t* ci[2];
    for (int i = 0; i < 2; ++i) {
        ci[i] = (int64_t*)(w + i * align256(p.cand_bytes));
        cs[i] = (float*)(ci[i] + (size_t)nq * p.nseg * k);
    }
    // (one_launch_ok = false: the two-stage search's top-64 of every row -- the merge of 64-entry lists by each query's last workgroup took 25.5 us
    //  at nq = 64 against 11 + 9 for two launches, and 2x the two-launch time at nq = 1024: profiles/r06_knn_*)
    if (one_launch_ok && k <= 64 && p.nseg <= 64) {   // one launch: the last workgroup of every query merges that query's segment lists
        unsigned int* tickets = (unsigned int*)(w + 2 * align256(p.cand_bytes));
        if (!tickets_zeroed && hipMemsetAsync(tickets, 0, (size_t)nq * sizeof(unsigned int), st) != hipSuccess) return ralf::check_launch("knn_select memset");
        hipLaunchKernelGGL((knn_select_kernel<true, true>), dim3((unsigned)p.nseg, nq), dim3(256), 0, st, S, nullptr, nullptr, N, N, (int64_t)SEG, k, cs[0], ci[0],
                           tickets, out_score, out_idx);
        return ralf::check_launch("knn_select");
    }
    hipLaunchKernelGGL((knn_select_kernel<true>), dim3((unsigned)p.nseg, nq), dim3(256), 0, st, S, nullptr, nullptr, N, N, (int64_t)SEG, k, cs[0], ci[0],
                       (unsigned int*)nullptr, (float*)nullptr, (int64_t*)nullptr);
    return merge_lists(cs, ci, p.nseg, nq, k, out_idx, out_score, st);
}

// ---- the whole two-stage search behind ONE entry point (round 6) ----
// Host-side, the search used to be eight calls through the Python wrapper (cast, norms, product, selection, re-score, candidate selection, each with
// its own output allocations): ~80 us of host time on top of ~200 us of kernels at nq = 64.  Here the launches go out back to back from C into one
// caller-provided workspace; the caller reads `bad` (one flag per query) and sends the uncertified queries through ralf_knn_topk_ip.
namespace {
// rows of the index the threshold pass ranks (expected list length: (pool + 1) * N / this).  RALF_KNN_FLT_SAMPLE: tuning aid -- at BASELINE config 4, 1024 queries:
// 2048 / 3072 / 4096 / 8192 rows = 431 / 431 / 435 / 446 us whole call (fewer rows: cheaper threshold pass, longer lists)
inline int64_t flt_sample_rows() { static const int64_t v = [] { const char* e = getenv("RALF_KNN_FLT_SAMPLE"); return e ? (int64_t)atoi(e) : (int64_t)4096; }(); return v > 0 ? v : 4096; }
// candidate slots per (query, column tile of the index) = tile / 8 (16 per 128 columns): ~2 hits expected per 128 columns, P(> slots) ~ 1e-10 on unordered data
inline int flt_capsh(int tile) { return tile >= 256 ? 5 : tile >= 128 ? 4 : 3; }
struct TwoStagePlan { size_t qb, qn, coarse, cval, cidx, exact, sel, over, total;
                      size_t f_sample, f_tval, f_tidx, f_cnt, f_lst; int64_t f_ns, f_T; int f_tile, f_capsh; };   // f_*: inside the `coarse` region
int filter_tile(int64_t N, int D, int nq) {
    RalfGemmDesc d;
    memset(&d, 0, sizeof(d));
    d.M = nq; d.N = (int)N; d.K = D; d.lda = D; d.ldb = D; d.ldc = N; d.nb0 = d.nb1 = 1; d.splitk = 1; d.dtype = RALF_BF16; d.a_kcontig = 1; d.b_kcontig = 1;
    return ralf_gemm_filter_tile(&d);
}
TwoStagePlan plan_two_stage(int64_t N, int D, int nq, int pool) {
    TwoStagePlan p;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += align256(bytes); return at; };
    p.qb = take((size_t)nq * D * 2);
    p.qn = take((size_t)nq * 3 * sizeof(float));
    // the score matrix of the dense coarse pass, or (filtered pass) the sample's scores, its selection, the tile counters and the candidate slots
    p.f_ns = N < flt_sample_rows() ? N : flt_sample_rows();
    p.f_tile = filter_tile(N, D, nq);
    p.f_T = p.f_tile > 0 ? (N + p.f_tile - 1) / p.f_tile : 0;
    p.f_capsh = flt_capsh(p.f_tile);
    size_t f = 0;
    auto ftake = [&](size_t bytes) { const size_t at = f; f += align256(bytes); return at; };
    p.f_sample = ftake((size_t)nq * p.f_ns * sizeof(float));
    p.f_tval = ftake((size_t)nq * (pool + 1) * sizeof(float));
    p.f_tidx = ftake((size_t)nq * (pool + 1) * sizeof(int64_t));
    p.f_cnt = ftake((size_t)nq * p.f_T * sizeof(int));
    p.f_lst = ftake(((size_t)nq * p.f_T << p.f_capsh) * 8);
    const size_t dense = (size_t)nq * N * sizeof(float);
    p.coarse = take(dense > f ? dense : f);
    p.cval = take((size_t)nq * (pool + 1) * sizeof(float));
    p.cidx = take((size_t)nq * (pool + 1) * sizeof(int64_t));
    p.exact = take((size_t)nq * (pool + 1) * sizeof(float));
    p.over = take((size_t)nq * sizeof(int32_t));
    const int64_t W = p.f_T << p.f_capsh;
    SelectPlan sp = plan_select(N > W ? N : W, nq, pool + 1);
    p.sel = take(2 * align256(sp.cand_bytes) + align256((size_t)nq * sizeof(unsigned int)) + 256);
    p.total = o;
    return p;
}

int two_stage_impl(const float* X, const void* Xb, int64_t N, int D, const float* Q, int nq, int k, int pool, const float* xnorms,
                   int64_t* out_idx, float* out_score, int32_t* bad, void* workspace, size_t workspace_bytes, void* stream, bool filtered) {
    RALF_REQUIRE(X && Xb && Q && xnorms && out_idx && out_score && bad && workspace, "knn two-stage: null pointer");
    RALF_REQUIRE(N > 0 && nq > 0 && D > 0 && D % 64 == 0, "knn two-stage: dim %d must be a multiple of 64 (bf16 coarse product on the aligned path)", D);
    RALF_REQUIRE(k >= 1 && k <= pool && pool + 1 <= KMAX && pool < N, "knn two-stage: k=%d pool=%d outside 1 <= k <= pool < min(%d, n_db)", k, pool, KMAX);
    RALF_REQUIRE(((uintptr_t)X & 15) == 0 && ((uintptr_t)Q & 15) == 0 && ((uintptr_t)Xb & 15) == 0 && ((uintptr_t)workspace & 255) == 0, "knn two-stage: alignment");
    const TwoStagePlan p = plan_two_stage(N, D, nq, pool);
    if (workspace_bytes < p.total) {
        ralf::set_error("knn two-stage: workspace %zu < required %zu bytes", workspace_bytes, p.total);
        return RALF_ERR_WORKSPACE;
    }
    char* w = (char*)workspace;
    void* qb = w + p.qb;
    float *qn = (float*)(w + p.qn), *coarse = (float*)(w + p.coarse), *cval = (float*)(w + p.cval), *exact = (float*)(w + p.exact);
    int64_t* cidx = (int64_t*)(w + p.cidx);
    int32_t* over = (int32_t*)(w + p.over);
    hipStream_t st = (hipStream_t)stream;
    // queries -> bf16, their norms {|q|, |qb|, |q - qb|}
    hipLaunchKernelGGL(knn_query_prep_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, st, Q, (__bf16*)qb, nq, D, qn, filtered ? over : (int32_t*)nullptr);
    RalfGemmDesc d;
    memset(&d, 0, sizeof(d));
    d.A = qb; d.B = Xb; d.C = coarse;
    d.M = nq; d.N = (int)N; d.K = D; d.lda = D; d.ldb = D; d.ldc = N; d.nb0 = d.nb1 = 1; d.splitk = 1; d.alpha = 1.f; d.aux_scale = 1.f;
    d.dtype = RALF_BF16; d.a_kcontig = 1; d.b_kcontig = 1; d.out_f32 = 1;
    if (!filtered) {
        // coarse scores [nq, N] = Qb Xb^T on the bf16 matrix cores, fp32 out; the pool + 1 best coarse rows per query, sorted by coarse score
        if (int rc = ralf_gemm(&d, nullptr, 0, stream)) return rc;
        if (int rc = select_impl(coarse, N, nq, pool + 1, cidx, cval, w + p.sel, p.total - p.sel, st, false, false)) return rc;
    } else {
        // The coarse scores never reach memory (252 MB written and re-read by the selection at BASELINE config 4).  A first product against the first f_ns rows
        // of the index gives every query a LOWER bound of its (pool+1)-th best coarse score (the (pool+1)-th best of a subset cannot exceed that of the whole);
        // the product over the whole index keeps only the scores at or above it, in per-(query, column tile) slot lists (RalfGemmDesc.flt_*: ~(pool+1) N / f_ns
        // candidates per query); the selection reads the slots.  A tile with more hits than slots flags its query (over -> bad): ordered indexes flood tiles, and
        // the caller is expected to fall back to the dense pass for an index that does (retrieval/knn.py: FlatIPIndex).
        RALF_REQUIRE(p.f_tile > 0 && (p.f_T << p.f_capsh) < ((int64_t)1 << 31) / 2, "knn two-stage (filtered): index of %lld rows too large for 32-bit slot positions", (long long)N);
        char* f = (char*)coarse;
        float *sample = (float*)(f + p.f_sample), *tval = (float*)(f + p.f_tval);
        int64_t* tidx = (int64_t*)(f + p.f_tidx);
        int* cnt = (int*)(f + p.f_cnt);
        void* lst = f + p.f_lst;
        RalfGemmDesc ds = d;
        ds.C = sample; ds.N = (int)p.f_ns; ds.ldc = p.f_ns;
        if (int rc = ralf_gemm(&ds, nullptr, 0, stream)) return rc;
        if (int rc = select_impl(sample, p.f_ns, nq, pool + 1, tidx, tval, w + p.sel, p.total - p.sel, st, false, false)) return rc;
        d.C = lst; d.out_f32 = 0;   // (C is not written by a filtered product; the entry refuses a null one)
        d.flt_thresh = tval + pool; d.flt_thresh_ld = pool + 1; d.flt_count = cnt; d.flt_list = lst; d.flt_cap = 1 << p.f_capsh;
        if (int rc = ralf_gemm(&d, nullptr, 0, stream)) return rc;
        const int64_t W = p.f_T << p.f_capsh, nseg = (W + SEG - 1) / SEG;
        if (nseg == 1) {
            hipLaunchKernelGGL(knn_select_lists_kernel, dim3(1, nq), dim3(256), 0, st, (const int2*)lst, cnt, (int)p.f_T, p.f_capsh, pool + 1, cval, cidx, over);
        } else {
            char* sw = w + p.sel;
            const size_t cb = align256((size_t)nq * nseg * (pool + 1) * (sizeof(float) + sizeof(int64_t)));
            float* cs[2];
            int64_t* ci[2];
            for (int i = 0; i < 2; ++i) {
                ci[i] = (int64_t*)(sw + i * cb);
                cs[i] = (float*)(ci[i] + (size_t)nq * nseg * (pool + 1));
            }
            hipLaunchKernelGGL(knn_select_lists_kernel, dim3((unsigned)nseg, nq), dim3(256), 0, st, (const int2*)lst, cnt, (int)p.f_T, p.f_capsh, pool + 1, cs[0], ci[0], over);
            if (int rc = merge_lists(cs, ci, nseg, nq, pool + 1, cidx, cval, st)) return rc;
        }
        if (int rc = ralf::check_launch("knn_select_lists")) return rc;
    }
    // exact scores of the candidates (the scan's accumulation chain), the k best of them, the certificate against the (pool + 1)-th coarse score
    if (int rc = launch_rescore(X, N, D, Q, nq, cidx, pool + 1, exact, st)) return rc;
    hipLaunchKernelGGL(knn_select_cand_kernel, dim3(nq), dim3(64), 0, st, exact, cidx, pool + 1, k, out_score, out_idx, cval + pool, (int64_t)(pool + 1), qn, xnorms, D, bad,
                       filtered ? (const int32_t*)over : (const int32_t*)nullptr);
    return ralf::check_launch("knn two-stage");
}
}  // namespace

extern "C" size_t ralf_knn_two_stage_workspace_bytes(int64_t N, int D, int nq, int pool) {
    if (N <= 0 || D <= 0 || nq <= 0 || pool <= 0) return 0;
    return plan_two_stage(N, D, nq, pool).total;
}

extern "C" int ralf_knn_topk_ip_two_stage(const float* X, const void* Xb, int64_t N, int D, const float* Q, int nq, int k, int pool, const float* xnorms,
                                          int64_t* out_idx, float* out_score, int32_t* bad, void* workspace, size_t workspace_bytes, void* stream) {
    return two_stage_impl(X, Xb, N, D, Q, nq, k, pool, xnorms, out_idx, out_score, bad, workspace, workspace_bytes, stream, false);
}

extern "C" int ralf_knn_topk_ip_two_stage_filtered(const float* X, const void* Xb, int64_t N, int D, const float* Q, int nq, int k, int pool, const float* xnorms,
                                                   int64_t* out_idx, float* out_score, int32_t* bad, void* workspace, size_t workspace_bytes, void* stream) {
    return two_stage_impl(X, Xb, N, D, Q, nq, k, pool, xnorms, out_idx, out_score, bad, workspace, workspace_bytes, stream, true);
}

extern "C" int ralf_knn_select(const float* S, int64_t N, int nq, int k, int64_t* out_idx, float* out_score, void* ws,
                               size_t ws_bytes, void* stream) {
    RALF_REQUIRE(S && out_idx && out_score, "knn_select: null pointer");
    RALF_REQUIRE(N > 0 && nq > 0, "knn_select: empty problem");
    RALF_REQUIRE(k >= 1 && k <= KMAX, "knn_select: k=%d outside [1,%d]", k, KMAX);
    return select_impl(S, N, nq, k, out_idx, out_score, ws, ws_bytes, (hipStream_t)stream, false);
}

extern "C" int ralf_knn_topk_ip(const float* X, int64_t N, int D, const float* Q, int nq, int k, int64_t* out_idx,
                                float* out_score, void* ws, size_t ws_bytes, void* stream) {
    RALF_REQUIRE(k >= 1 && k <= KMAX, "knn_topk_ip: k=%d outside [1,%d]", k, KMAX);
    RALF_REQUIRE(N > 0 && nq > 0, "knn_topk_ip: empty problem (n_db=%lld nq=%d)", (long long)N, nq);
    const size_t need = ralf_knn_topk_ip_workspace_bytes(N, D, nq, k);
    if (!ws || ws_bytes < need) {
        ralf::set_error("knn_topk_ip: workspace %zu < required %zu bytes", ws_bytes, need);
        return RALF_ERR_WORKSPACE;
    }
    RALF_REQUIRE(((uintptr_t)ws & 255) == 0, "knn_topk_ip: workspace must be 256-byte aligned");
    float* S = (float*)ws;
    const size_t soff = align256((size_t)nq * N * sizeof(float));
    hipStream_t st = (hipStream_t)stream;
    // the ticket counters of the selection's hand-off are zeroed BY the scan kernel (its first workgroup): neither a memset node
    // between scan and selection (critical path) nor one in front of the scan
    SelectPlan p = plan_select(N, nq, k);
    const bool fused = p.nseg > 1 && k <= 64 && p.nseg <= 64;
    RALF_REQUIRE(X && Q, "knn_topk_ip: null pointer");
    RALF_REQUIRE(D > 0 && D % 4 == 0, "knn_topk_ip: dim %d must be a multiple of 4 (16-byte row alignment)", D);
    RALF_REQUIRE(((uintptr_t)X & 15) == 0 && ((uintptr_t)Q & 15) == 0, "knn_topk_ip: index/queries must be 16-byte aligned");
    int rc = scores_impl(X, N, D, Q, nq, S, st, fused ? (unsigned int*)((char*)ws + soff + 2 * align256(p.cand_bytes)) : nullptr, fused ? nq : 0);
    if (rc) return rc;
    return select_impl(S, N, nq, k, out_idx, out_score, (char*)ws + soff, ws_bytes - soff, st, fused);
}
