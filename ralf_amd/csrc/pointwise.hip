// HBM-bound byte/gather kernels of the RALF train step for gfx950: token-embedding gather (+ x sqrt(d)
// + positional table), counter-based dropout, label-smoothed cross-entropy, scalar-flag add, dtype
// casts / weight re-layouts, NCHW->NHWC image packing, 3x3/s2 max-pool and nearest up-sampling.
//
// Reference call sites (paths relative to the reference root):
//   BaseDecoder.forward / UserConstraintTransformerEncoder.forward: emb -> PositionalEncoding1d
//       image2layout/train/models/common/common.py:97-98,245-246; common/positional_encoding.py:92-107
//   nn.Dropout sites of nn.Transformer*Layer, PositionalEncoding1d
//   nn.CrossEntropyLoss(label_smoothing=0.1, ignore_index=pad)  retrieval_augmented_autoreg.py:140-142,213-214
//   task_emb flag add                                            retrieval_augmented_autoreg.py:1022-1028
//   ResnetBackbone max-pool / F.interpolate(nearest)            common/image.py:66-67,103-105
#include "common.h"
#include "sample_core.h"

namespace {
typedef __bf16 bf16;

template <typename T> __device__ __forceinline__ float ld(const T* p, int64_t i) { return (float)p[i]; }
template <typename T> __device__ __forceinline__ void st(T* p, int64_t i, float v) { p[i] = (T)v; }

using sample_core::rng24;
using sample_core::wave_max;
using sample_core::wave_sum;

// division by a launch-time constant as multiply-high + shift (n < 2^31; the FastDiv of gemm_impl.h): the index arithmetic of the per-pixel kernels below
// ran on 64-bit hardware-less divisions -- 6 per 16-byte element in the stem's fused BatchNorm / ReLU / max-pool backward, which streamed at 1.1-2.2 TB/s
struct FDiv {
    uint32_t mul, shr, den;
    __device__ __forceinline__ uint32_t div(uint32_t n) const { const uint32_t t = __umulhi(n, mul) >> shr; return den == 1 ? n : t; }
    __device__ __forceinline__ void divmod(uint32_t n, int& q, int& r) const { const uint32_t t = div(n); q = (int)t; r = (int)(n - t * den); }
};
inline FDiv make_fdiv(uint32_t d) {
    FDiv f;
    f.den = d ? d : 1;
    if (f.den == 1) { f.mul = 0; f.shr = 0; return f; }
    uint32_t lg = 0;
    while ((1u << lg) < f.den) ++lg;
    const uint32_t p = 31 + lg;
    f.mul = (uint32_t)((((uint64_t)1 << p) + f.den - 1) / f.den);
    f.shr = p - 32;
    return f;
}

inline int grid_for(int64_t n, int per_block = 256, int cap = 4096) {
    int64_t b = (n + per_block - 1) / per_block;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

// out[r][:] = W[idx[r]][:] * scale + pe[r % S][:]
template <typename T>
__global__ void embed_fwd_kernel(const int64_t* __restrict__ idx, const float* __restrict__ W, const float* __restrict__ pe, T* __restrict__ out,
                                 int64_t rows, int S, int d, float scale) {
    const int64_t total = rows * d;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / d;
        const int c = (int)(e - r * d);
        float v = W[idx[r] * d + c] * scale;
        if (pe) v += pe[(int64_t)(r % S) * d + c];
        st(out, e, v);
    }
}
template <typename T>
__global__ void embed_bwd_kernel(const int64_t* __restrict__ idx, const T* __restrict__ dy, float* __restrict__ dW, int64_t rows, int d, float scale) {
    const int64_t total = rows * d;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / d;
        atomicAdd(dW + idx[r] * d + (e - r * d), ld(dy, e) * scale);
    }
}

// y = x * keep/(1-p); the same call regenerates the same mask (used for the gradient)
template <typename T>
__global__ void dropout_kernel(const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ y, int64_t n, float p, const int64_t* __restrict__ seed, uint64_t call) {
    const uint64_t s = p > 0.f ? (uint64_t)seed[0] : 0;
    const uint32_t thr = drop_thr16(p);
    const float inv = 1.f / (1.f - p);
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        float v = (p <= 0.f || drop_keep1(s, call, (uint64_t)e, thr)) ? ld(x, e) * inv : 0.f;
        if (res) v += ld(res, e);
        st(y, e, v);
    }
}

// label-smoothed cross entropy, one wave per row of fp32 logits [rows, V]
__global__ void xent_count_kernel(const int64_t* __restrict__ target, int64_t rows, int ignore_index, float* __restrict__ cnt_loss) {
    __shared__ float red[4];
    float c = 0.f;
    for (int64_t r = threadIdx.x; r < rows; r += 256) c += (target[r] != ignore_index);
    c = wave_sum(c);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) { cnt_loss[0] = red[0] + red[1] + red[2] + red[3]; cnt_loss[1] = 0.f; }
}
template <typename T>
__global__ __launch_bounds__(256) void xent_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target, T* __restrict__ dlogits,
                                                    float* __restrict__ cnt_loss, int64_t rows, int V, int ignore_index, float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float inv_cnt = 1.f / cnt_loss[0];
    float lsum = 0.f;
    for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < rows; r += (int64_t)gridDim.x * 4) {
        const float* x = logits + r * V;
        const int64_t t = target[r];
        if (t == ignore_index) {
            if (dlogits) for (int c = lane; c < V; c += 64) st(dlogits, r * V + c, 0.f);
            continue;
        }
        float mx = -__builtin_inff(), sx = 0.f;
        for (int c = lane; c < V; c += 64) { mx = fmaxf(mx, x[c]); sx += x[c]; }
        mx = wave_max(mx);
        sx = wave_sum(sx);
        float se = 0.f;
        for (int c = lane; c < V; c += 64) se += __expf(x[c] - mx);
        se = wave_sum(se);
        const float lse = mx + __logf(se);
        lsum += ((1.f - eps) * (lse - x[t]) + eps * (lse - sx / V)) * inv_cnt;
        if (dlogits)
            for (int c = lane; c < V; c += 64)
                st(dlogits, r * V + c, (__expf(x[c] - lse) - (c == t ? 1.f - eps : 0.f) - eps / V) * inv_cnt);
    }
    // one atomic per workgroup (one per ROW was 3200 same-address atomics: 30 us for 7 MB of logits)
    __shared__ float red[4];
    if (lane == 0) red[wave] = lsum;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(cnt_loss + 1, red[0] + red[1] + red[2] + red[3]);
}

// y = x + s[0]  (learned scalar flag broadcast over all channels);  sum_all: out[0] += sum x
template <typename T>
__global__ void add_scalar_kernel(const T* __restrict__ x, const float* __restrict__ s, T* __restrict__ y, int64_t rows, int cols, int64_t ldx, int64_t ldy) {
    const float sv = s[0];
    const int64_t total = rows * cols;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / cols;
        const int c = (int)(e - r * cols);
        st(y, r * ldy + c, ld(x, r * ldx + c) + sv);
    }
}
// counter += inc on the device (the AdamW step count, the dropout seed): graph-replay safe, no framework kernel
__global__ void counter_add_kernel(void* p, int is64, long long inc) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (is64) *reinterpret_cast<long long*>(p) += inc;
        else *reinterpret_cast<int*>(p) += (int)inc;
    }
}
// frozen layout encoder input (fid/model.py:90-103): geometry columns -> [R*N, 8] rows (cx, cy, w, h, 0, 0, 0, 0) in the compute dtype
// (8 columns: a 16-byte bf16 operand row for the fc_bbox product), and the key-padding mask of the [token; elements] sequence
template <typename T>
__global__ void layout_pack_kernel(const float* __restrict__ cx, const float* __restrict__ cy, const float* __restrict__ w, const float* __restrict__ h,
                                   const unsigned char* __restrict__ mask, T* __restrict__ bbox, unsigned char* __restrict__ kpm, int64_t R, int N) {
    const int64_t total = R * N;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / N;
        const int n = (int)(e - r * N);
        T* o = bbox + e * 8;
        o[0] = (T)cx[e]; o[1] = (T)cy[e]; o[2] = (T)w[e]; o[3] = (T)h[e];
        o[4] = (T)0.f; o[5] = (T)0.f; o[6] = (T)0.f; o[7] = (T)0.f;
        kpm[r * (N + 1) + 1 + n] = mask[e] ? 0 : 1;
        if (n == 0) kpm[r * (N + 1)] = 0;
    }
}
// y = x * s[0] with the factor on the device (the incoming gradient of a scalar loss: 1 or 1/world, graph-replay safe)
template <typename T>
__global__ void scale_dev_kernel(const T* __restrict__ x, const float* __restrict__ s, float* __restrict__ y, int64_t n) {
    const float sv = s[0];
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) y[e] = ld(x, e) * sv;
}
// zero fill in 16-byte stores (the flat gradient buffer at the start of a step)
__global__ void zero_kernel(uint4* __restrict__ p, int64_t n16) {
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n16; e += (int64_t)gridDim.x * 256) p[e] = z;
}
template <typename T>
__global__ void sum_all_kernel(const T* __restrict__ x, float* __restrict__ out, int64_t rows, int cols, int64_t ldx) {
    __shared__ float red[4];
    float a = 0.f;
    const int64_t total = rows * cols;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / cols;
        a += ld(x, r * ldx + (e - r * cols));
    }
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}


// ---- sequence concatenation [B, S_0 + S_1 + ..., d] of up to 4 sources [B, S_i, d], each with an optional learned scalar added
// (the flag embedding nn.Embedding(2, 1) of retrieval_augmented_autoreg.py:1022-1028), and its backward: the gradient split back
// into contiguous pieces + the scalars' gradients (sums).  One 16-byte vector per thread, no per-element index division.
struct CatSrc { const void* p; const float* scalar; float* dscalar; int rows; int off; };   // rows per batch entry, first output row
struct CatParams { CatSrc s[4]; int nsrc; int B; int total_rows; int vec_per_row; };

template <typename T, bool BWD>
__global__ __launch_bounds__(256) void concat_rows_kernel(const CatParams P, void* out_v) {
    constexpr int VEC = 16 / (int)sizeof(T);
    const int vpr = P.vec_per_row;
    const int64_t nvec = (int64_t)P.B * P.total_rows * vpr;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * 256) {
        const int64_t row = v / vpr;                   // (one division per 16-byte vector)
        const int c = (int)(v - row * vpr);
        const int b = (int)(row / P.total_rows), r = (int)(row - (int64_t)b * P.total_rows);
        int si = 0;
#pragma unroll
        for (int i = 1; i < 4; ++i) si += (i < P.nsrc && r >= P.s[i].off);
        const CatSrc& S = P.s[si];
        const int64_t sidx = (((int64_t)b * S.rows + (r - S.off)) * vpr + c) * VEC;
        const int64_t oidx = v * VEC;
        T tmp[VEC];
        if (!BWD) {
            *reinterpret_cast<uint4*>(tmp) = *reinterpret_cast<const uint4*>((const T*)S.p + sidx);
            if (S.scalar) {
                const float sv = S.scalar[0];
#pragma unroll
                for (int i = 0; i < VEC; ++i) tmp[i] = (T)((float)tmp[i] + sv);
            }
            *reinterpret_cast<uint4*>((T*)out_v + oidx) = *reinterpret_cast<uint4*>(tmp);
        } else {   // out_v = the gradient of the concatenation (read), S.p = the piece's gradient (written)
            *reinterpret_cast<uint4*>(tmp) = *reinterpret_cast<const uint4*>((const T*)out_v + oidx);
            *reinterpret_cast<uint4*>((T*)const_cast<void*>(S.p) + sidx) = *reinterpret_cast<uint4*>(tmp);
            if (S.dscalar) {
                float a = 0.f;
#pragma unroll
                for (int i = 0; i < VEC; ++i) a += (float)tmp[i];
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] += (si == i) ? a : 0.f;   // (acc[si] puts the array into scratch memory: 214 us instead of 8)
            }
        }
    }
    if (BWD) {   // one atomic per workgroup and scalar (one per WAVE was 16 k same-address atomics: 214 us for an 8 us copy)
        __shared__ float red[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float t = wave_sum(acc[i]);
            if ((threadIdx.x & 63) == 0) red[i][threadIdx.x >> 6] = t;
        }
        __syncthreads();
        if (threadIdx.x < 4 && (int)threadIdx.x < P.nsrc && P.s[threadIdx.x].dscalar)
            atomicAdd(P.s[threadIdx.x].dscalar, red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3]);
    }
}

// y = dropout_p(x * scale + pe[r % S])  (PositionalEncoding1d on a [rows, d] tensor; pe fp32 [S, d] or NULL)
template <typename T>
__global__ __launch_bounds__(256) void scale_pe_drop_kernel(const T* __restrict__ x, const float* __restrict__ pe, T* __restrict__ y, int64_t rows, int S, int d,
                                                           float scale, float p, const int64_t* __restrict__ seed, uint64_t call) {
    constexpr int VEC = 16 / (int)sizeof(T);
    const int vpr = d / VEC;
    const int64_t nvec = rows * vpr;
    const uint64_t sd = p > 0.f ? (uint64_t)seed[0] : 0;
    const uint32_t thr = drop_thr16(p);
    const float inv = 1.f / (1.f - p);
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * 256) {
        const int64_t row = v / vpr;
        const int c = (int)(v - row * vpr) * VEC;
        T tmp[VEC];
        *reinterpret_cast<uint4*>(tmp) = *reinterpret_cast<const uint4*>(x + v * VEC);
        const float* pr = pe ? pe + (row % S) * d + c : nullptr;
        uint64_t hh[VEC / 4];
        if (p > 0.f) {
#pragma unroll
            for (int g = 0; g < VEC / 4; ++g) hh[g] = drop_hash4(sd, call, (uint64_t)(v * (VEC / 4) + g));   // (one hash per 4 elements: common.h)
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            float f = (float)tmp[i] * scale + (pr ? pr[i] : 0.f);
            if (p > 0.f) f = drop_keep(hh[i / 4], i & 3, thr) ? (float)(T)f * inv : 0.f;   // mask the value as stored (ralf_dropout semantics)
            tmp[i] = (T)f;
        }
        *reinterpret_cast<uint4*>(y + v * VEC) = *reinterpret_cast<uint4*>(tmp);
    }
}

// strided 2-D copy with dtype conversion (concat / slice / cast): dst[r*ldd + c] = src[r*lds + c]
template <typename TS, typename TD>
__global__ void copy2d_kernel(const TS* __restrict__ src, TD* __restrict__ dst, int64_t rows, int cols, int64_t lds, int64_t ldd, int accumulate) {
    const int64_t total = rows * cols;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / cols;
        const int c = (int)(e - r * cols);
        float v = ld(src, r * lds + c);
        if (accumulate) v += ld(dst, r * ldd + c);
        st(dst, r * ldd + c, v);
    }
}

// 4-D permute with conversion and zero padding of the (new) innermost dim:
// out[i0][i1][i2][i3] (dims od[], innermost padded to od[3] >= source extent) = in[...] with in-stride per out-dim
template <typename TS, typename TD>
__global__ void permute4_kernel(const TS* __restrict__ in, TD* __restrict__ out, int d0, int d1, int d2, int d3, int64_t s0, int64_t s1, int64_t s2, int64_t s3, int valid3) {
    const int64_t total = (int64_t)d0 * d1 * d2 * d3;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        int64_t t = e;
        const int i3 = (int)(t % d3); t /= d3;
        const int i2 = (int)(t % d2); t /= d2;
        const int i1 = (int)(t % d1); t /= d1;
        const int i0 = (int)t;
        st(out, e, i3 < valid3 ? ld(in, i0 * s0 + i1 * s1 + i2 * s2 + i3 * s3) : 0.f);
    }
}

// the same with an innermost output dim of exactly 8 (NCHW image -> NHWC with the channels padded to 8: one pixel per thread,
// plane-wise coalesced reads, one 16-byte (bf16) / 32-byte (fp32) store instead of 8 scalar ones with index math per element)
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void permute4_pack8_kernel(const TS* __restrict__ in, TD* __restrict__ out, int d0, int d1, int d2, int64_t s0, int64_t s1, int64_t s2,
                                                              int64_t s3, int valid3) {
    const int64_t total = (int64_t)d0 * d1 * d2;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        int64_t t = e;
        const int i2 = (int)(t % d2); t /= d2;
        const int i1 = (int)(t % d1);
        const int i0 = (int)(t / d1);
        const int64_t base = i0 * s0 + i1 * s1 + i2 * s2;
        float v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {   // (unconditional load of a clamped channel plane, zero selected afterwards)
            const float t = ld(in, base + (c < valid3 ? c : valid3 - 1) * s3);
            v[c] = c < valid3 ? t : 0.f;
        }
        if constexpr (sizeof(TD) == 2) {
            typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
            bf16x8v o;
#pragma unroll
            for (int c = 0; c < 8; ++c) o[c] = (__bf16)v[c];
            *reinterpret_cast<bf16x8v*>(out + e * 8) = o;
        } else {
            *reinterpret_cast<float4*>(out + e * 8) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(out + e * 8 + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
    }
}

// many permute4 jobs in ONE launch (the per-step re-layout of every 3x3 convolution weight: ~40 launches of a few
// microseconds each otherwise); jobs[] lives in device memory, job j owns workgroups [first_block[j], first_block[j+1])
template <typename TS, typename TD>
__device__ __forceinline__ void permute4_job(const RalfPermuteJob& J, int blk, int nblk) {
    const TS* in = (const TS*)J.in;
    TD* out = (TD*)J.out;
    const int64_t total = (int64_t)J.d0 * J.d1 * J.d2 * J.d3;
    for (int64_t e = (int64_t)blk * 256 + threadIdx.x; e < total; e += (int64_t)nblk * 256) {
        int64_t t = e;
        const int i3 = (int)(t % J.d3); t /= J.d3;
        const int i2 = (int)(t % J.d2); t /= J.d2;
        const int i1 = (int)(t % J.d1); t /= J.d1;
        const int i0 = (int)t;
        st(out, e, i3 < J.valid3 ? ld(in, i0 * J.s0 + i1 * J.s1 + i2 * J.s2 + i3 * J.s3) : 0.f);
    }
}
__global__ __launch_bounds__(256) void permute4_batched_kernel(const RalfPermuteJob* __restrict__ jobs, int njobs) {
    int j = 0;
    while (j + 1 < njobs && (int)blockIdx.x >= jobs[j + 1].first_block) ++j;   // njobs is small (tens): linear scan
    const RalfPermuteJob J = jobs[j];
    const int nblk = (j + 1 < njobs ? jobs[j + 1].first_block : (int)gridDim.x) - J.first_block;
    const int blk = (int)blockIdx.x - J.first_block;
    if (J.src_dtype == RALF_F32 && J.dst_dtype == RALF_F32) permute4_job<float, float>(J, blk, nblk);
    else if (J.src_dtype == RALF_F32) permute4_job<float, bf16>(J, blk, nblk);
    else if (J.dst_dtype == RALF_F32) permute4_job<bf16, float>(J, blk, nblk);
    else permute4_job<bf16, bf16>(J, blk, nblk);
}

// ---- both GEMM layouts of a k x k convolution weight from ONE read of the fp32 OIHW master:
//   ohwi[o][t][i] (i padded with zeros to Cip: the forward's k-contiguous B operand) and ikwo[i][t][o] (the data gradient's), t = kh*kw.
// A workgroup owns a tile of 8 output channels x 64 input channels (all taps): the reads are runs of 64*KK contiguous floats per
// output channel, the writes 128-byte runs of i (ohwi) and 16-byte vectors of o (ikwo).  (The generic element-wise permute read
// with a stride and divided three times per element: 109 us per step for the 17 weights of the ResNet-50 + FPN, 45 MB.)
constexpr int RL_O = 8, RL_LS = 577;   // 8 output channels x (i-tile * KK <= 576) floats (+1: bank spread).  (32 channels per tile -- 64-byte instead of 16-byte runs in the
                                       // [i][tap][o] image, 74 KB of LDS -- ran at 127 us against 58.)
__host__ __device__ __forceinline__ int rl_itile(int KK) { return KK <= 9 ? 64 : (576 / KK > 0 ? 576 / KK : 1); }
template <typename TD>
__device__ __forceinline__ void relayout_tile(const RalfConvRelayoutJob& J, int tile, float* lds) {
    const int KK = J.KK, TI = rl_itile(KK), nti = (J.Ci + TI - 1) / TI;
    const int o0 = (tile / nti) * RL_O, i0 = (tile % nti) * TI;
    const int ni = min(TI, J.Ci - i0), no = min(RL_O, J.Co - o0);
    const float* in = (const float*)J.w;
    const int run = ni * KK;                       // contiguous floats per output channel
    // e / d for e < 32 * 576 and d <= 576 through the float reciprocal: (e + 0.5) / d stays >= 0.5 / 576 away from an integer, the product's error is
    // ~4e-6 -- exact.  (Integer divisions by run-time values were most of this kernel: six per two-byte element.)
    auto qdiv = [](int e, float inv) { return (int)(((float)e + 0.5f) * inv); };
    const float inv_run = 1.f / (float)run;
    for (int e = threadIdx.x; e < RL_O * run; e += 256) {
        const int o = qdiv(e, inv_run), r = e - o * run;
        lds[o * RL_LS + r] = o < no ? in[((int64_t)(o0 + o) * J.Ci + i0) * KK + r] : 0.f;
    }
    __syncthreads();
    TD* o1 = (TD*)J.ohwi;
    TD* o2 = (TD*)J.ikwo;
    // ohwi: [o][t][i0 + i]; channels beyond Ci (up to Cip) are zero
    const int nip = (i0 + TI >= J.Ci) ? (J.Cip - i0) : TI;     // the last i-tile also writes the padding
    const float inv_nip = 1.f / (float)nip, inv_kk = 1.f / (float)KK;
    for (int e = threadIdx.x; e < no * KK * nip; e += 256) {
        const int q = qdiv(e, inv_nip), i = e - q * nip, o = qdiv(q, inv_kk), t = q - o * KK;
        const float v = i < ni ? lds[o * RL_LS + i * KK + t] : 0.f;
        o1[((int64_t)(o0 + o) * KK + t) * J.Cip + i0 + i] = (TD)v;
    }
    // ikwo: [i0 + i][t][o0 + o]
    const float inv_no = 1.f / (float)no;
    for (int e = threadIdx.x; e < ni * KK * no; e += 256) {
        const int q = qdiv(e, inv_no), o = e - q * no, i = qdiv(q, inv_kk), t = q - i * KK;
        o2[((int64_t)(i0 + i) * KK + t) * J.Co + o0 + o] = (TD)lds[o * RL_LS + i * KK + t];
    }
    __syncthreads();
}
__global__ __launch_bounds__(256) void conv_relayout_batched_kernel(const RalfConvRelayoutJob* __restrict__ jobs, int njobs) {
    __shared__ float lds[RL_O * RL_LS];
    int j = 0;
    while (j + 1 < njobs && (int)blockIdx.x >= jobs[j + 1].first_block) ++j;
    const RalfConvRelayoutJob J = jobs[j];
    const int nblk = (j + 1 < njobs ? jobs[j + 1].first_block : (int)gridDim.x) - J.first_block;
    const int TI = rl_itile(J.KK);
    const int ntiles = ((J.Co + RL_O - 1) / RL_O) * ((J.Ci + TI - 1) / TI);
    for (int tile = (int)blockIdx.x - J.first_block; tile < ntiles; tile += nblk) {
        if (J.dst_dtype == RALF_F32) relayout_tile<float>(J, tile, lds); else relayout_tile<bf16>(J, tile, lds);
    }
}

// 3x3 stride-2 pad-1 max pooling, NHWC; arg = window position (kh*3+kw) of the FIRST maximum.
// One thread owns VEC consecutive channels of one pixel (16-byte accesses, 32-bit index math):
// the scalar one-element-per-thread version ran at 1/8 of the HBM rate.
template <typename T> struct PV;   // pooling vector: 16 bytes of channels
template <> struct PV<float> {
    static constexpr int N = 4;
    static __device__ __forceinline__ void load(const float* p, float (&v)[4]) { const float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
    static __device__ __forceinline__ void store(float* p, const float (&v)[4]) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
};
template <> struct PV<bf16> {
    static constexpr int N = 8;
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    static __device__ __forceinline__ void load(const bf16* p, float (&v)[8]) { const bf16x8 t = *reinterpret_cast<const bf16x8*>(p); for (int i = 0; i < 8; ++i) v[i] = (float)t[i]; }
    static __device__ __forceinline__ void store(bf16* p, const float (&v)[8]) { bf16x8 t; for (int i = 0; i < 8; ++i) t[i] = (bf16)v[i]; *reinterpret_cast<bf16x8*>(p) = t; }
};

template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int8_t* __restrict__ arg, int B, int H, int W, int C, int OH, int OW) {
    constexpr int N = PV<T>::N;
    const int cv = C / N;
    const int64_t total = (int64_t)B * OH * OW * cv;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % cv) * N;
        const int pix = (int)(e / cv);
        const int ow = pix % OW, t = pix / OW, oh = t % OH, b = t / OH;
        float best[N];
        __attribute__((aligned(8))) int8_t bi[N];
#pragma unroll
        for (int i = 0; i < N; ++i) { best[i] = -__builtin_inff(); bi[i] = -1; }
        // all 9 taps are loaded unconditionally from clamped coordinates (a load under `if (inside)` waits for itself before the next
        // tap's address is even computed: 9 dependent round trips per output vector), padding taps are skipped in the comparison
        float v[9][N];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int ih = min(max(oh * 2 - 1 + kh, 0), H - 1);
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int iw = min(max(ow * 2 - 1 + kw, 0), W - 1);
                PV<T>::load(x + (((int64_t)b * H + ih) * W + iw) * C + c, v[kh * 3 + kw]);
            }
        }
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int ih = oh * 2 - 1 + kh;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int iw = ow * 2 - 1 + kw;
                const bool inside = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
#pragma unroll
                for (int i = 0; i < N; ++i)
                    if (inside && (bi[i] < 0 || v[kh * 3 + kw][i] > best[i])) { best[i] = v[kh * 3 + kw][i]; bi[i] = (int8_t)(kh * 3 + kw); }
            }
        }
        const int64_t o = (int64_t)pix * C + c;
        PV<T>::store(y + o, best);
        if constexpr (N == 8) *reinterpret_cast<uint2*>(arg + o) = *reinterpret_cast<const uint2*>(bi);   // one 8-byte store, not 8 byte stores
        else *reinterpret_cast<uint32_t*>(arg + o) = *reinterpret_cast<const uint32_t*>(bi);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ dy, const int8_t* __restrict__ arg, T* __restrict__ dx, int B, int H, int W, int C, int OH, int OW) {
    constexpr int N = PV<T>::N;
    const int cv = C / N;
    const int64_t total = (int64_t)B * H * W * cv;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % cv) * N;
        const int pix = (int)(e / cv);
        const int iw = pix % W, t = pix / W, ih = t % H, b = t / H;
        float g[N];
#pragma unroll
        for (int i = 0; i < N; ++i) g[i] = 0.f;
#pragma unroll
        for (int dh = 0; dh < 2; ++dh) {          // windows with oh*2-1+kh == ih, kh in [0,2]: oh in {ih/2, (ih+1)/2}
            const int oh = (ih + dh) / 2;
            if (dh == 1 && oh == ih / 2) continue;
            const int kh = ih - (oh * 2 - 1);
            if (oh >= OH || kh < 0 || kh > 2) continue;
#pragma unroll
            for (int dw = 0; dw < 2; ++dw) {
                const int ow = (iw + dw) / 2;
                if (dw == 1 && ow == iw / 2) continue;
                const int kw = iw - (ow * 2 - 1);
                if (ow >= OW || kw < 0 || kw > 2) continue;
                const int64_t o = (((int64_t)b * OH + oh) * OW + ow) * C + c;
                float v[N];
                PV<T>::load(dy + o, v);
                int8_t a[N];
                if constexpr (N == 8) *reinterpret_cast<uint2*>(a) = *reinterpret_cast<const uint2*>(arg + o);
                else *reinterpret_cast<uint32_t*>(a) = *reinterpret_cast<const uint32_t*>(arg + o);
                const int8_t want = (int8_t)(kh * 3 + kw);
#pragma unroll
                for (int i = 0; i < N; ++i)
                    if (a[i] == want) g[i] += v[i];
            }
        }
        PV<T>::store(dx + (int64_t)pix * C + c, g);
    }
}

// nearest up-sampling (torch semantics: src = min(floor(dst * in/out), in-1)) fused with the FPN add:
//   up[b,y,x,:]  = src[b,sy,sx,:]  -> written with row stride ld_up (into the concat buffer)
//   sum[b,y,x,:] = up + lateral
template <typename T>
__global__ void upsample_add_kernel(const T* __restrict__ src, const T* __restrict__ lateral, T* __restrict__ up, int64_t ld_up, T* __restrict__ sum,
                                    int B, int IH, int IW, int OH, int OW, int C) {
    const float sy_scale = (float)IH / OH, sx_scale = (float)IW / OW;
    const int64_t total = (int64_t)B * OH * OW * C;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        int64_t t = e;
        const int c = (int)(t % C); t /= C;
        const int x = (int)(t % OW); t /= OW;
        const int y = (int)(t % OH);
        const int b = (int)(t / OH);
        const int sy = min((int)floorf(y * sy_scale), IH - 1), sx = min((int)floorf(x * sx_scale), IW - 1);
        const float v = ld(src, (((int64_t)b * IH + sy) * IW + sx) * C + c);
        st(up, (e / C) * ld_up + c, v);
        st(sum, e, v + ld(lateral, e));
    }
}
// dsrc[b,sy,sx,:] = sum over destination pixels mapping to (sy,sx) of (g1[dst] (row stride ld1) + g2[dst])
template <typename T>
__global__ void upsample_bwd_kernel(const T* __restrict__ g1, int64_t ld1, const T* __restrict__ g2, T* __restrict__ dsrc,
                                    int B, int IH, int IW, int OH, int OW, int C) {
    const float sy_scale = (float)IH / OH, sx_scale = (float)IW / OW;
    const int64_t total = (int64_t)B * IH * IW * C;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        int64_t t = e;
        const int c = (int)(t % C); t /= C;
        const int sx = (int)(t % IW); t /= IW;
        const int sy = (int)(t % IH);
        const int b = (int)(t / IH);
        float g = 0.f;
        for (int y = 0; y < OH; ++y) {
            if (min((int)floorf(y * sy_scale), IH - 1) != sy) continue;
            for (int x = 0; x < OW; ++x) {
                if (min((int)floorf(x * sx_scale), IW - 1) != sx) continue;
                const int64_t p = ((int64_t)b * OH + y) * OW + x;
                g += ld(g1, p * ld1 + c) + ld(g2, p * C + c);
            }
        }
        st(dsrc, e, g);
    }
}

// per-row decode-space mask + token choice (one wave per row of fp32 logits [B,V]): sample_core.h mask_sample_row
__global__ __launch_bounds__(256) void mask_sample_kernel(const float* __restrict__ logits, const uint8_t* __restrict__ allowed,
                                                           const int64_t* __restrict__ forced, int mode, int top_k, float temperature, float top_p,
                                                           const int64_t* __restrict__ seed, uint64_t call, int64_t* __restrict__ out, int B, int V,
                                                           int64_t* __restrict__ seq_out, int64_t seq_ld, uint8_t* __restrict__ flag_out, int64_t flag_ld, int64_t pad_id,
                                                           int row0) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    auto emit = [&](int64_t tok) {   // lane 0: the token, its column of the sequence buffer, its key-padding flag
        out[row] = tok;
        if (seq_out) seq_out[(int64_t)row * seq_ld] = tok;
        if (flag_out) flag_out[(int64_t)row * flag_ld] = tok == pad_id ? 1 : 0;
    };
    sample_core::mask_sample_row(logits + (int64_t)row * V, allowed, forced ? forced[row] : -1, mode, top_k, temperature, top_p, seed, call,
                                 (uint64_t)row + (uint64_t)row0, V, lane, emit);
}

}  // namespace

#define DISPATCH_T(dtype, ...)                                            \
    do {                                                                  \
        if ((dtype) == RALF_F32) { typedef float T; __VA_ARGS__; }        \
        else if ((dtype) == RALF_BF16) { typedef bf16 T; __VA_ARGS__; }   \
        else { ralf::set_error("bad dtype %d", (dtype)); return RALF_ERR_INVALID; } \
    } while (0)
#define ST ((hipStream_t)stream)

extern "C" int ralf_embed_fwd(int dtype, const int64_t* idx, const float* W, const float* pe, void* out, int64_t rows, int S, int d, float scale, void* stream) {
    RALF_REQUIRE(idx && W && out && rows > 0 && d > 0 && S > 0, "embed_fwd: bad arguments");
    DISPATCH_T(dtype, hipLaunchKernelGGL((embed_fwd_kernel<T>), dim3(grid_for(rows * d)), dim3(256), 0, ST, idx, W, pe, (T*)out, rows, S, d, scale));
    return ralf::check_launch("embed_fwd");
}
/* dW (fp32 [vocab, d]) is accumulated into */
extern "C" int ralf_embed_bwd(int dtype, const int64_t* idx, const void* dy, float* dW, int64_t rows, int d, float scale, void* stream) {
    RALF_REQUIRE(idx && dy && dW && rows > 0 && d > 0, "embed_bwd: bad arguments");
    DISPATCH_T(dtype, hipLaunchKernelGGL((embed_bwd_kernel<T>), dim3(grid_for(rows * d)), dim3(256), 0, ST, idx, (const T*)dy, dW, rows, d, scale));
    return ralf::check_launch("embed_bwd");
}
extern "C" int ralf_dropout(int dtype, const void* x, const void* res, void* y, int64_t n, float p, const int64_t* seed, uint64_t call_id, void* stream) {
    RALF_REQUIRE(x && y && (seed || p == 0.f) && n > 0 && p >= 0.f && p < 1.f, "dropout: bad arguments");
    DISPATCH_T(dtype, hipLaunchKernelGGL((dropout_kernel<T>), dim3(grid_for(n)), dim3(256), 0, ST, (const T*)x, (const T*)res, (T*)y, n, p, seed, call_id));
    return ralf::check_launch("dropout");
}
/* cnt_loss: fp32[2] = {number of non-ignored targets, mean loss}; dlogits (dtype) = d loss / d logits, may be NULL */
extern "C" int ralf_xent_fwd_bwd(int dtype, const float* logits, const int64_t* target, void* dlogits, float* cnt_loss, int64_t rows, int V,
                                 int ignore_index, float label_smoothing, void* stream) {
    RALF_REQUIRE(logits && target && cnt_loss && rows > 0 && V > 0, "xent: bad arguments");
    hipLaunchKernelGGL(xent_count_kernel, dim3(1), dim3(256), 0, ST, target, rows, ignore_index, cnt_loss);
    DISPATCH_T(dtype, hipLaunchKernelGGL((xent_kernel<T>), dim3(grid_for(rows, 4, 512)), dim3(256), 0, ST, logits, target, (T*)dlogits, cnt_loss, rows, V, ignore_index, label_smoothing));
    return ralf::check_launch("xent");
}
extern "C" int ralf_add_scalar(int dtype, const void* x, const float* s, void* y, int64_t rows, int cols, int64_t ldx, int64_t ldy, void* stream) {
    RALF_REQUIRE(x && s && y && rows > 0 && cols > 0, "add_scalar: bad arguments");
    DISPATCH_T(dtype, hipLaunchKernelGGL((add_scalar_kernel<T>), dim3(grid_for(rows * cols)), dim3(256), 0, ST, (const T*)x, s, (T*)y, rows, cols, ldx, ldy));
    return ralf::check_launch("add_scalar");
}
extern "C" int ralf_counter_add(void* counter, int is_int64, int64_t inc, void* stream) {
    RALF_REQUIRE(counter, "counter_add: null pointer");
    hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(64), 0, ST, counter, is_int64, (long long)inc);
    return ralf::check_launch("counter_add");
}
extern "C" int ralf_layout_pack(int dtype, const float* cx, const float* cy, const float* w, const float* h, const uint8_t* mask, void* bbox, uint8_t* kpm,
                                int64_t R, int N, void* stream) {
    RALF_REQUIRE(cx && cy && w && h && mask && bbox && kpm && R > 0 && N > 0, "layout_pack: bad arguments");
    DISPATCH_T(dtype, hipLaunchKernelGGL((layout_pack_kernel<T>), dim3(grid_for(R * N)), dim3(256), 0, ST, cx, cy, w, h, mask, (T*)bbox, kpm, R, N));
    return ralf::check_launch("layout_pack");
}
extern "C" int ralf_scale_dev(int dtype, const void* x, const float* s, float* y, int64_t n, void* stream) {
    RALF_REQUIRE(x && s && y && n > 0, "scale_dev: bad arguments");
    DISPATCH_T(dtype, hipLaunchKernelGGL((scale_dev_kernel<T>), dim3(grid_for(n)), dim3(256), 0, ST, (const T*)x, s, y, n));
    return ralf::check_launch("scale_dev");
}
extern "C" int ralf_zero(void* p, int64_t nbytes, void* stream) {
    RALF_REQUIRE(p && nbytes >= 0 && ((uintptr_t)p % 16) == 0 && nbytes % 16 == 0, "zero: 16-byte aligned buffer and size");
    if (nbytes == 0) return RALF_OK;
    hipLaunchKernelGGL(zero_kernel, dim3(grid_for(nbytes / 16, 256, 2048)), dim3(256), 0, ST, (uint4*)p, nbytes / 16);
    return ralf::check_launch("zero");
}
/* out[0] += sum of the [rows, cols] view */
extern "C" int ralf_sum_all(int dtype, const void* x, float* out, int64_t rows, int cols, int64_t ldx, void* stream) {
    RALF_REQUIRE(x && out && rows > 0 && cols > 0, "sum_all: bad arguments");
    DISPATCH_T(dtype, hipLaunchKernelGGL((sum_all_kernel<T>), dim3(grid_for(rows * cols, 256, 256)), dim3(256), 0, ST, (const T*)x, out, rows, cols, ldx));
    return ralf::check_launch("sum_all");
}

/* out[b, off_i + r, :] = src_i[b, r, :] (+ scalar_i[0])  for up to 4 sources [B, rows_i, d] -> out [B, sum rows_i, d]  (backward = 0), or
 * the reverse split of a gradient `out` into the pieces dsrc_i (+ dscalar_i[0] += sum of the piece) (backward = 1).  d * sizeof % 16 == 0. */
extern "C" int ralf_concat_rows(int dtype, int backward, int nsrc, const void* const* src, const int* rows, const float* const* scalar, float* const* dscalar,
                                void* out, int B, int d, void* stream) {
    RALF_REQUIRE(nsrc >= 1 && nsrc <= 4 && src && rows && out && B > 0 && d > 0, "concat_rows: bad arguments");
    const int vec = dtype == RALF_F32 ? 4 : 8;
    RALF_REQUIRE(d % vec == 0 && ((uintptr_t)out % 16) == 0, "concat_rows: d %% %d and 16-byte alignment", vec);
    CatParams P;
    int off = 0;
    for (int i = 0; i < 4; ++i) {
        P.s[i].p = nullptr; P.s[i].scalar = nullptr; P.s[i].dscalar = nullptr; P.s[i].rows = 0; P.s[i].off = 0;
        if (i < nsrc) {
            RALF_REQUIRE(src[i] && rows[i] > 0 && ((uintptr_t)src[i] % 16) == 0, "concat_rows: source %d", i);
            P.s[i].p = src[i]; P.s[i].rows = rows[i]; P.s[i].off = off;
            P.s[i].scalar = (!backward && scalar) ? scalar[i] : nullptr;
            P.s[i].dscalar = (backward && dscalar) ? dscalar[i] : nullptr;
            off += rows[i];
        }
    }
    P.nsrc = nsrc; P.B = B; P.total_rows = off; P.vec_per_row = d / vec;
    const dim3 g(grid_for((int64_t)B * off * P.vec_per_row, 256, backward ? 512 : 2048));   // backward: 512 atomics per scalar at most
    if (dtype == RALF_F32) { if (backward) hipLaunchKernelGGL((concat_rows_kernel<float, true>), g, dim3(256), 0, ST, P, out); else hipLaunchKernelGGL((concat_rows_kernel<float, false>), g, dim3(256), 0, ST, P, out); }
    else { if (backward) hipLaunchKernelGGL((concat_rows_kernel<bf16, true>), g, dim3(256), 0, ST, P, out); else hipLaunchKernelGGL((concat_rows_kernel<bf16, false>), g, dim3(256), 0, ST, P, out); }
    return ralf::check_launch("concat_rows");
}
/* y = dropout_p(x * scale + pe[r % S, :])  on [rows, d] (PositionalEncoding1d of the K retrieved features, common/positional_encoding.py:92-107);
 * pe fp32 [S, d] or NULL; the same call on a gradient with pe = NULL is the backward */
extern "C" int ralf_scale_pe_dropout(int dtype, const void* x, const float* pe, void* y, int64_t rows, int S, int d, float scale, float p,
                                     const int64_t* seed, uint64_t call_id, void* stream) {
    RALF_REQUIRE(x && y && rows > 0 && d > 0 && (seed || p == 0.f) && p >= 0.f && p < 1.f && (S > 0 || !pe), "scale_pe_dropout: bad arguments");
    const int vec = dtype == RALF_F32 ? 4 : 8;
    RALF_REQUIRE(d % vec == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0, "scale_pe_dropout: d %% %d and 16-byte alignment", vec);
    DISPATCH_T(dtype, hipLaunchKernelGGL((scale_pe_drop_kernel<T>), dim3(grid_for(rows * (d / vec), 256, 2048)), dim3(256), 0, ST, (const T*)x, pe, (T*)y, rows, S > 0 ? S : 1, d, scale, p, seed, call_id));
    return ralf::check_launch("scale_pe_dropout");
}
/* dst[r*ldd+c] (+)= src[r*lds+c] with conversion between RALF_F32 / RALF_BF16 */
extern "C" int ralf_copy2d(int src_dtype, int dst_dtype, const void* src, void* dst, int64_t rows, int cols, int64_t lds, int64_t ldd, int accumulate, void* stream) {
    RALF_REQUIRE(src && dst && rows > 0 && cols > 0, "copy2d: bad arguments");
    const dim3 g(grid_for(rows * cols));
    if (src_dtype == RALF_F32 && dst_dtype == RALF_F32) hipLaunchKernelGGL((copy2d_kernel<float, float>), g, dim3(256), 0, ST, (const float*)src, (float*)dst, rows, cols, lds, ldd, accumulate);
    else if (src_dtype == RALF_F32) hipLaunchKernelGGL((copy2d_kernel<float, bf16>), g, dim3(256), 0, ST, (const float*)src, (bf16*)dst, rows, cols, lds, ldd, accumulate);
    else if (dst_dtype == RALF_F32) hipLaunchKernelGGL((copy2d_kernel<bf16, float>), g, dim3(256), 0, ST, (const bf16*)src, (float*)dst, rows, cols, lds, ldd, accumulate);
    else hipLaunchKernelGGL((copy2d_kernel<bf16, bf16>), g, dim3(256), 0, ST, (const bf16*)src, (bf16*)dst, rows, cols, lds, ldd, accumulate);
    return ralf::check_launch("copy2d");
}
/* out (dims d0..d3 contiguous, dtype dst) [i0,i1,i2,i3] = in[i0*s0+i1*s1+i2*s2+i3*s3] for i3 < valid3 else 0 */
extern "C" int ralf_permute4(int src_dtype, int dst_dtype, const void* in, void* out, int d0, int d1, int d2, int d3, int64_t s0, int64_t s1, int64_t s2, int64_t s3,
                             int valid3, void* stream) {
    RALF_REQUIRE(in && out && d0 > 0 && d1 > 0 && d2 > 0 && d3 > 0, "permute4: bad arguments");
    if (d3 == 8 && (src_dtype == RALF_F32 || dst_dtype != RALF_F32) && s2 == 1 && (((uintptr_t)out) & 31) == 0) {   // pixel packing (see permute4_pack8_kernel)
        const dim3 gp(grid_for((int64_t)d0 * d1 * d2));
        if (src_dtype != RALF_F32)   // (an image batch uploaded in bf16: models/ralf.py: sample())
            hipLaunchKernelGGL((permute4_pack8_kernel<bf16, bf16>), gp, dim3(256), 0, ST, (const bf16*)in, (bf16*)out, d0, d1, d2, s0, s1, s2, s3, valid3);
        else if (dst_dtype == RALF_F32) hipLaunchKernelGGL((permute4_pack8_kernel<float, float>), gp, dim3(256), 0, ST, (const float*)in, (float*)out, d0, d1, d2, s0, s1, s2, s3, valid3);
        else hipLaunchKernelGGL((permute4_pack8_kernel<float, bf16>), gp, dim3(256), 0, ST, (const float*)in, (bf16*)out, d0, d1, d2, s0, s1, s2, s3, valid3);
        return ralf::check_launch("permute4");
    }
    const dim3 g(grid_for((int64_t)d0 * d1 * d2 * d3));
    if (src_dtype == RALF_F32 && dst_dtype == RALF_F32) hipLaunchKernelGGL((permute4_kernel<float, float>), g, dim3(256), 0, ST, (const float*)in, (float*)out, d0, d1, d2, d3, s0, s1, s2, s3, valid3);
    else if (src_dtype == RALF_F32) hipLaunchKernelGGL((permute4_kernel<float, bf16>), g, dim3(256), 0, ST, (const float*)in, (bf16*)out, d0, d1, d2, d3, s0, s1, s2, s3, valid3);
    else if (dst_dtype == RALF_F32) hipLaunchKernelGGL((permute4_kernel<bf16, float>), g, dim3(256), 0, ST, (const bf16*)in, (float*)out, d0, d1, d2, d3, s0, s1, s2, s3, valid3);
    else hipLaunchKernelGGL((permute4_kernel<bf16, bf16>), g, dim3(256), 0, ST, (const bf16*)in, (bf16*)out, d0, d1, d2, d3, s0, s1, s2, s3, valid3);
    return ralf::check_launch("permute4");
}
extern "C" int ralf_permute4_batched(const RalfPermuteJob* jobs_device, int njobs, int total_blocks, void* stream) {
    RALF_REQUIRE(jobs_device && njobs > 0 && total_blocks >= njobs, "permute4_batched: bad arguments");
    hipLaunchKernelGGL(permute4_batched_kernel, dim3(total_blocks), dim3(256), 0, ST, jobs_device, njobs);
    return ralf::check_launch("permute4_batched");
}
extern "C" int ralf_conv_relayout_batched(const RalfConvRelayoutJob* jobs_device, int njobs, int total_blocks, void* stream) {
    RALF_REQUIRE(jobs_device && njobs > 0 && total_blocks >= njobs, "conv_relayout_batched: bad arguments");
    hipLaunchKernelGGL(conv_relayout_batched_kernel, dim3(total_blocks), dim3(256), 0, ST, jobs_device, njobs);
    return ralf::check_launch("conv_relayout_batched");
}
// ---- the stem's BatchNorm + ReLU + 3x3/2 max-pool as ONE pass over the convolution output (timm resnet50: conv1 -> bn1 -> act1 -> maxpool,
// image2layout/train/models/common/image.py:39-48,66-67).  z = relu(y * scale + shift) is never stored: 134 MB written and read again at
// B = 64, 256 x 256.  Every tap is normalised, rounded to T (what ralf_bn_apply would have stored) and compared, so pooled values and argmax
// equal the ralf_bn_apply -> ralf_maxpool3x3s2_fwd chain bit for bit.
namespace {
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_maxpool_fwd_kernel(const T* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                                                                   T* __restrict__ y, int8_t* __restrict__ arg, int B, int H, int W, int C, int OH, int OW) {
    constexpr int N = PV<T>::N;
    const int cv = C / N;
    const int64_t total = (int64_t)B * OH * OW * cv;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % cv) * N;
        const int pix = (int)(e / cv);
        const int ow = pix % OW, t = pix / OW, oh = t % OH, b = t / OH;
        float sc[N], sh[N];
#pragma unroll
        for (int i = 0; i < N; ++i) { sc[i] = scale[c + i]; sh[i] = shift[c + i]; }
        float best[N];
        __attribute__((aligned(8))) int8_t bi[N];
#pragma unroll
        for (int i = 0; i < N; ++i) { best[i] = -__builtin_inff(); bi[i] = -1; }
        float v[9][N];   // (unconditional loads of clamped taps, see maxpool_fwd_kernel)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int ih = min(max(oh * 2 - 1 + kh, 0), H - 1);
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int iw = min(max(ow * 2 - 1 + kw, 0), W - 1);
                PV<T>::load(x + (((int64_t)b * H + ih) * W + iw) * C + c, v[kh * 3 + kw]);
            }
        }
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int ih = oh * 2 - 1 + kh;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int iw = ow * 2 - 1 + kw;
                const bool inside = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    const float z = (float)(T)fmaxf(__builtin_fmaf(v[kh * 3 + kw][i], sc[i], sh[i]), 0.f);
                    if (inside && (bi[i] < 0 || z > best[i])) { best[i] = z; bi[i] = (int8_t)(kh * 3 + kw); }
                }
            }
        }
        const int64_t o = (int64_t)pix * C + c;
        PV<T>::store(y + o, best);
        if constexpr (N == 8) *reinterpret_cast<uint2*>(arg + o) = *reinterpret_cast<const uint2*>(bi);
        else *reinterpret_cast<uint32_t*>(arg + o) = *reinterpret_cast<const uint32_t*>(bi);
    }
}

// the gradient that reaches z = relu(BN(y)) at an input pixel through the pooling: the sum over the <= 4 windows whose argmax it is, rounded to T (what
// ralf_maxpool3x3s2_bwd stores), zero where the ReLU was inactive (y * scale + shift <= 0).
// dz of the 2 x 2 pixel block (rows 2 k, 2 k + 1; columns 2 l, 2 l + 1) of image b: its four pixels draw on the four pooling windows (k .. k + 1) x (l .. l + 1)
// and on no other, so each window's gradient and arg-max vector is loaded ONCE for the block (a pixel at a time they were loaded 9 times per block, behind
// parity-dependent branches: the two kernels below were bound by their ~240 instructions per 16-byte element, 1.1-2.2 TB/s).  Pixel (dy, dx) sits in window
// (wy, wx) at tap (dy - 2 wy + 1, dx - 2 wx + 1) where that is inside 0 .. 2; windows are added in the order (0,0), (0,1), (1,0), (1,1) -- pooled_dz's order.
template <typename T, int N>
__device__ __forceinline__ void pooled_dz_2x2(const T* __restrict__ dpool, const int8_t* __restrict__ arg, const float (&yv)[4][N], const float (&sc)[N], const float (&sh)[N],
                                              int b, int k, int l, int c, int C, int OH, int OW, float (&g)[4][N]) {
    float wv[4][N];
    __attribute__((aligned(8))) int8_t wa[4][N];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const int oh = min(k + (w >> 1), OH - 1), ow = min(l + (w & 1), OW - 1);   // (clamped: a window beyond the edge is loaded from a valid place and not used)
        const int64_t o = (((int64_t)b * OH + oh) * OW + ow) * C + c;
        PV<T>::load(dpool + o, wv[w]);
        if constexpr (N == 8) *reinterpret_cast<uint2*>(wa[w]) = *reinterpret_cast<const uint2*>(arg + o);
        else *reinterpret_cast<uint32_t*>(wa[w]) = *reinterpret_cast<const uint32_t*>(arg + o);
    }
    const bool wok[4] = {true, l + 1 < OW, k + 1 < OH, (k + 1 < OH) && (l + 1 < OW)};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int dy = q >> 1, dx = q & 1;
#pragma unroll
        for (int i = 0; i < N; ++i) g[q][i] = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int kh = dy - 2 * (w >> 1) + 1, kw = dx - 2 * (w & 1) + 1;   // (compile-time)
            if (kh < 0 || kh > 2 || kw < 0 || kw > 2) continue;
            const int8_t want = (int8_t)(kh * 3 + kw);
            if (wok[w]) {
#pragma unroll
                for (int i = 0; i < N; ++i)
                    if (wa[w][i] == want) g[q][i] += wv[w][i];
            }
        }
#pragma unroll
        for (int i = 0; i < N; ++i) g[q][i] = __builtin_fmaf(yv[q][i], sc[i], sh[i]) > 0.f ? (float)(T)g[q][i] : 0.f;
    }
}

// backward, pass 1: per workgroup the column sums of dz and of dz * (y - mean) -> part[blockIdx.x][2][C] (the layout of RalfGemmDesc.bnb_part:
// ralf_bn_bwd_stats_from_partials finishes them).  A thread owns one channel vector and walks 2 x 2 pixel blocks (block coordinates advanced with carries).
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_maxpool_bwd_reduce_kernel(const T* __restrict__ dpool, const int8_t* __restrict__ arg, const T* __restrict__ y,
                                                                          const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ mean,
                                                                          float* __restrict__ part, int B, int H, int W, int C, int OH, int OW, FDiv fkw, FDiv fkh) {
    constexpr int N = PV<T>::N;
    __shared__ float red[2][256][N];
    const int cv = C / N;                 // channel vectors per pixel (<= 256, a power of two: C = 64)
    const int ppi = 256 / cv;             // blocks per iteration of the workgroup
    const int tx = threadIdx.x % cv, ty = threadIdx.x / cv;
    const int c = tx * N;
    float sc[N], sh[N], mu[N], a1[N], a2[N];
#pragma unroll
    for (int i = 0; i < N; ++i) { sc[i] = scale[c + i]; sh[i] = shift[c + i]; mu[i] = mean[c + i]; a1[i] = a2[i] = 0.f; }
    const int KH = (int)fkh.den, KW = (int)fkw.den, nblocks = B * KH * KW;
    const int stride = (int)gridDim.x * ppi;
    int sl, sk, sb, t0;
    fkw.divmod((uint32_t)stride, t0, sl);
    fkh.divmod((uint32_t)t0, sb, sk);
    int q = (int)blockIdx.x * ppi + ty, l, k, b;
    fkw.divmod((uint32_t)min(q, nblocks - 1), t0, l);
    fkh.divmod((uint32_t)t0, b, k);
    for (; q < nblocks; q += stride) {
        float yv[4][N], g[4][N];
        const bool rok = 2 * k + 1 < H, cok = 2 * l + 1 < W;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int ih = min(2 * k + (p >> 1), H - 1), iw = min(2 * l + (p & 1), W - 1);
            PV<T>::load(y + (((int64_t)b * H + ih) * W + iw) * C + c, yv[p]);
        }
        pooled_dz_2x2<T, N>(dpool, arg, yv, sc, sh, b, k, l, c, C, OH, OW, g);
        const bool pok[4] = {true, cok, rok, rok && cok};
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if (pok[p]) {
#pragma unroll
                for (int i = 0; i < N; ++i) { a1[i] += g[p][i]; a2[i] += g[p][i] * (yv[p][i] - mu[i]); }
            }
        }
        l += sl;
        const int cl = l >= KW;
        l -= cl ? KW : 0;
        k += sk + cl;
        const int ck = k >= KH;
        k -= ck ? KH : 0;
        b += sb + ck;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) { red[0][threadIdx.x][i] = a1[i]; red[1][threadIdx.x][i] = a2[i]; }
    __syncthreads();
    if (ty == 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            float t1 = 0.f, t2 = 0.f;
            for (int j = 0; j < ppi; ++j) { t1 += red[0][j * cv + tx][i]; t2 += red[1][j * cv + tx][i]; }
            part[((int64_t)blockIdx.x * 2 + 0) * C + c + i] = t1;
            part[((int64_t)blockIdx.x * 2 + 1) * C + c + i] = t2;
        }
    }
}

// backward, pass 2: dy = c1 * dz + c2 * y + c3 (the affine form of the BatchNorm backward apply), dz recomputed as in pass 1; a thread per (2 x 2 block, channel vector)
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_maxpool_bwd_apply_kernel(const T* __restrict__ dpool, const int8_t* __restrict__ arg, const T* __restrict__ y,
                                                                         const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ c1p,
                                                                         const float* __restrict__ c2p, const float* __restrict__ c3p, T* __restrict__ dy,
                                                                         int B, int H, int W, int C, int OH, int OW, FDiv fcv, FDiv fkw, FDiv fkh) {
    constexpr int N = PV<T>::N;
    const uint32_t total = (uint32_t)B * fkh.den * fkw.den * fcv.den;   // (< 2^31: checked at launch)
    for (uint32_t e = blockIdx.x * 256u + threadIdx.x; e < total; e += gridDim.x * 256u) {
        int q, cq, t, l, k, b;
        fcv.divmod(e, q, cq);
        fkw.divmod((uint32_t)q, t, l);
        fkh.divmod((uint32_t)t, b, k);
        const int c = cq * N;
        float sc[N], sh[N], k1[N], k2[N], k3[N], yv[4][N], g[4][N];
#pragma unroll
        for (int i = 0; i < N; ++i) { sc[i] = scale[c + i]; sh[i] = shift[c + i]; k1[i] = c1p[c + i]; k2[i] = c2p[c + i]; k3[i] = c3p[c + i]; }
        const bool rok = 2 * k + 1 < H, cok = 2 * l + 1 < W;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int ih = min(2 * k + (p >> 1), H - 1), iw = min(2 * l + (p & 1), W - 1);
            PV<T>::load(y + (((int64_t)b * H + ih) * W + iw) * C + c, yv[p]);
        }
        pooled_dz_2x2<T, N>(dpool, arg, yv, sc, sh, b, k, l, c, C, OH, OW, g);
        const bool pok[4] = {true, cok, rok, rok && cok};
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if (pok[p]) {
                float o[N];
#pragma unroll
                for (int i = 0; i < N; ++i) o[i] = __builtin_fmaf(g[p][i], k1[i], __builtin_fmaf(yv[p][i], k2[i], k3[i]));
                PV<T>::store(dy + (((int64_t)b * H + 2 * k + (p >> 1)) * W + 2 * l + (p & 1)) * C + c, o);
            }
        }
    }
}
}  // namespace

extern "C" int ralf_bn_relu_maxpool_fwd(int dtype, const void* y, const float* scale, const float* shift, void* out, int8_t* arg, int B, int H, int W, int C, void* stream) {
    RALF_REQUIRE(y && scale && shift && out && arg && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "bn_relu_maxpool_fwd: bad arguments (C %% 8 == 0)");
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    DISPATCH_T(dtype, hipLaunchKernelGGL((bn_relu_maxpool_fwd_kernel<T>), dim3(grid_for((int64_t)B * OH * OW * C / 4, 256, 1 << 20)), dim3(256), 0, ST, (const T*)y, scale, shift, (T*)out, arg, B, H, W, C, OH, OW));
    return ralf::check_launch("bn_relu_maxpool_fwd");
}
extern "C" int ralf_bn_relu_maxpool_bwd_reduce(int dtype, const void* dpool, const int8_t* arg, const void* y, const float* scale, const float* shift, const float* mean,
                                               float* part, int nblk, int B, int H, int W, int C, void* stream) {
    RALF_REQUIRE(dpool && arg && y && scale && shift && mean && part && nblk > 0 && B > 0 && H > 0 && W > 0, "bn_relu_maxpool_bwd_reduce: bad arguments");
    RALF_REQUIRE(C % 8 == 0 && C <= 1024 && (256 % (C / (dtype == RALF_F32 ? 4 : 8))) == 0, "bn_relu_maxpool_bwd_reduce: C / vector width must divide 256");
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    RALF_REQUIRE((int64_t)B * H * W * C < (1ll << 31), "bn_relu_maxpool_bwd_reduce: 32-bit element index");
    DISPATCH_T(dtype, hipLaunchKernelGGL((bn_relu_maxpool_bwd_reduce_kernel<T>), dim3(nblk), dim3(256), 0, ST, (const T*)dpool, arg, (const T*)y, scale, shift, mean, part, B, H, W, C, OH, OW,
                                         make_fdiv((uint32_t)((W + 1) / 2)), make_fdiv((uint32_t)((H + 1) / 2))));
    return ralf::check_launch("bn_relu_maxpool_bwd_reduce");
}
extern "C" int ralf_bn_relu_maxpool_bwd_apply(int dtype, const void* dpool, const int8_t* arg, const void* y, const float* scale, const float* shift, const float* c1,
                                              const float* c2, const float* c3, void* dy, int B, int H, int W, int C, void* stream) {
    RALF_REQUIRE(dpool && arg && y && scale && shift && c1 && c2 && c3 && dy && B > 0 && H > 0 && W > 0 && C % 8 == 0, "bn_relu_maxpool_bwd_apply: bad arguments");
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    RALF_REQUIRE((int64_t)B * H * W * C < (1ll << 31), "bn_relu_maxpool_bwd_apply: 32-bit element index");
    const int nvec = dtype == RALF_F32 ? 4 : 8, KH = (H + 1) / 2, KW = (W + 1) / 2;
    DISPATCH_T(dtype, hipLaunchKernelGGL((bn_relu_maxpool_bwd_apply_kernel<T>), dim3(grid_for((int64_t)B * KH * KW * (C / nvec), 256, 1 << 20)), dim3(256), 0, ST, (const T*)dpool, arg, (const T*)y, scale, shift, c1, c2, c3, (T*)dy, B, H, W, C, OH, OW,
                                         make_fdiv((uint32_t)(C / nvec)), make_fdiv((uint32_t)KW), make_fdiv((uint32_t)KH)));
    return ralf::check_launch("bn_relu_maxpool_bwd_apply");
}

extern "C" int ralf_maxpool3x3s2_fwd(int dtype, const void* x, void* y, int8_t* arg, int B, int H, int W, int C, void* stream) {
    RALF_REQUIRE(x && y && arg && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "maxpool_fwd: bad arguments (C %% 8 == 0)");
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    DISPATCH_T(dtype, hipLaunchKernelGGL((maxpool_fwd_kernel<T>), dim3(grid_for((int64_t)B * OH * OW * C / 4, 256, 1 << 20)), dim3(256), 0, ST, (const T*)x, (T*)y, arg, B, H, W, C, OH, OW));
    return ralf::check_launch("maxpool_fwd");
}
extern "C" int ralf_maxpool3x3s2_bwd(int dtype, const void* dy, const int8_t* arg, void* dx, int B, int H, int W, int C, void* stream) {
    RALF_REQUIRE(dy && dx && arg && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "maxpool_bwd: bad arguments (C %% 8 == 0)");
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    DISPATCH_T(dtype, hipLaunchKernelGGL((maxpool_bwd_kernel<T>), dim3(grid_for((int64_t)B * H * W * C / 4, 256, 1 << 20)), dim3(256), 0, ST, (const T*)dy, arg, (T*)dx, B, H, W, C, OH, OW));
    return ralf::check_launch("maxpool_bwd");
}
extern "C" int ralf_upsample_nearest_add(int dtype, const void* src, const void* lateral, void* up, int64_t ld_up, void* sum, int B, int IH, int IW, int OH, int OW, int C, void* stream) {
    RALF_REQUIRE(src && lateral && up && sum, "upsample_add: null pointer");
    DISPATCH_T(dtype, hipLaunchKernelGGL((upsample_add_kernel<T>), dim3(grid_for((int64_t)B * OH * OW * C)), dim3(256), 0, ST, (const T*)src, (const T*)lateral, (T*)up, ld_up, (T*)sum, B, IH, IW, OH, OW, C));
    return ralf::check_launch("upsample_add");
}
extern "C" int ralf_upsample_nearest_bwd(int dtype, const void* g_up, int64_t ld_up, const void* g_sum, void* dsrc, int B, int IH, int IW, int OH, int OW, int C, void* stream) {
    RALF_REQUIRE(g_up && g_sum && dsrc, "upsample_bwd: null pointer");
    DISPATCH_T(dtype, hipLaunchKernelGGL((upsample_bwd_kernel<T>), dim3(grid_for((int64_t)B * IH * IW * C)), dim3(256), 0, ST, (const T*)g_up, ld_up, (const T*)g_sum, (T*)dsrc, B, IH, IW, OH, OW, C));
    return ralf::check_launch("upsample_bwd");
}

/* decode-space restriction + sampling on the device (helpers/sampling.py:18-71, decoding_space_restriction.py):
 * logits fp32 [B,V] (V <= 1024), allowed uint8 [V] or NULL, forced int64 [B] or NULL (-1 = free),
 * mode 0 deterministic (argmax), 1 top-k multinomial with temperature; out int64 [B] */
extern "C" int ralf_mask_sample_step(const float* logits, const uint8_t* allowed, const int64_t* forced, int mode, int top_k, float temperature,
                                     const int64_t* seed, uint64_t call_id, int64_t* out, int64_t* seq_out, int64_t seq_ld, uint8_t* pad_flag_out,
                                     int64_t flag_ld, int64_t pad_id, int B, int V, float top_p, int row0, void* stream) {
    RALF_REQUIRE(logits && out && B > 0 && V > 0 && V <= 1024 && row0 >= 0, "mask_sample: bad arguments (V <= 1024)");
    RALF_REQUIRE(mode >= 0 && mode <= 4, "mask_sample: mode 0 (argmax), 1 (top_k), 2 (top_p), 3 (random), 4 (gumbel)");
    RALF_REQUIRE(mode == 0 || (seed && temperature > 0.f), "mask_sample: sampling needs a seed and T > 0");
    RALF_REQUIRE(mode != 1 || top_k >= 1, "mask_sample: top-k sampling needs k >= 1");
    RALF_REQUIRE(mode != 2 || (top_p > 0.f && top_p <= 1.f), "mask_sample: top-p sampling needs 0 < top_p <= 1");
    RALF_REQUIRE((!seq_out || seq_ld > 0) && (!pad_flag_out || flag_ld > 0), "mask_sample: output strides must be positive");
    hipLaunchKernelGGL(mask_sample_kernel, dim3((B + 3) / 4), dim3(256), 0, ST, logits, allowed, forced, mode, top_k, temperature, top_p, seed, call_id, out, B, V,
                       seq_out, seq_ld, pad_flag_out, flag_ld, pad_id, row0);
    return ralf::check_launch("mask_sample");
}
extern "C" int ralf_mask_sample(const float* logits, const uint8_t* allowed, const int64_t* forced, int mode, int top_k, float temperature,
                                const int64_t* seed, uint64_t call_id, int64_t* out, int B, int V, float top_p, void* stream) {
    return ralf_mask_sample_step(logits, allowed, forced, mode, top_k, temperature, seed, call_id, out, nullptr, 0, nullptr, 0, -1, B, V, top_p, 0, stream);
}
