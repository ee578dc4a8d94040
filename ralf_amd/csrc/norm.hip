// LayerNorm and BatchNorm2d (NHWC) forward / backward for gfx950 -- HBM-bound wave64 reduction
// kernels (no MFMA: these are byte-moving ops, sized to the memory roofline).
//
// Replaces nn.LayerNorm inside nn.TransformerEncoderLayer/DecoderLayer, FeedForward, Attention and
// BaseDecoder.head (image2layout/train/models/common/attention.py:19,41; common/common.py:38-40) and
// the nn.BatchNorm2d layers of the timm ResNet-50 body (common/image.py:39-67), train mode = batch
// statistics, momentum 0.1, eps 1e-5, unbiased running_var.
#include "common.h"

namespace {
typedef __bf16 bf16;

template <typename T> struct V4;  // 4 consecutive elements
template <> struct V4<float> {
    static __device__ __forceinline__ void load(const float* p, float (&v)[4]) { float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
    static __device__ __forceinline__ void store(float* p, const float (&v)[4]) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
};
template <> struct V4<bf16> {
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ void load(const bf16* p, float (&v)[4]) { bf16x4 t = *reinterpret_cast<const bf16x4*>(p); for (int i = 0; i < 4; ++i) v[i] = (float)t[i]; }
    static __device__ __forceinline__ void store(bf16* p, const float (&v)[4]) { bf16x4 t; for (int i = 0; i < 4; ++i) t[i] = (bf16)v[i]; *reinterpret_cast<bf16x4*>(p) = t; }
};

template <typename T> struct VL;  // 16 bytes of consecutive elements
template <> struct VL<float> {
    static constexpr int N = 4;
    static __device__ __forceinline__ void load(const float* p, float (&v)[4]) { V4<float>::load(p, v); }
};
template <> struct VL<bf16> {
    static constexpr int N = 8;
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    static __device__ __forceinline__ void load(const bf16* p, float (&v)[8]) { const bf16x8 t = *reinterpret_cast<const bf16x8*>(p); for (int i = 0; i < 8; ++i) v[i] = (float)t[i]; }
};

__device__ __forceinline__ float wave_sum(float v) {
    return wave::sum64_desc(v);   // the descending butterfly, bit for bit, without the LDS crossbar (wave_ops.h)
}

// ---------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, lane l owns columns {c*256 + 4l .. +3}; cols % 256 == 0, <= 1024
// ---------------------------------------------------------------------------------------------
template <typename T, int NCH>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      T* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd, int rows, float eps) {
    constexpr int cols = NCH * 256;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        float v[NCH][4];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            V4<T>::load(x + (int64_t)row * cols + c * 256 + lane * 4, v[c]);
            s += v[c][0] + v[c][1] + v[c][2] + v[c][3];
        }
        const float mu = wave_sum(s) * (1.f / cols);
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) { const float d = v[c][i] - mu; q = __fmaf_rn(d, d, q); }   // (explicit fma here and below: tlayer.hip writes the same bits)
        const float rs = rsqrtf(__fmaf_rn(wave_sum(q), 1.f / cols, eps));
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            float g[4], b[4], o[4];
            V4<float>::load(gamma + c * 256 + lane * 4, g);
            V4<float>::load(beta + c * 256 + lane * 4, b);
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = __fmaf_rn((v[c][i] - mu) * rs, g[i], b[i]);
            V4<T>::store(y + (int64_t)row * cols + c * 256 + lane * 4, o);
        }
        if (lane == 0 && mean) { mean[row] = mu; rstd[row] = rs; }
    }
}

// dx per row + per-workgroup partial dgamma/dbeta (registers -> LDS -> one fp32 atomic per column).
// 16 waves per workgroup, two rows in flight per wave: the kernel is latency-bound otherwise (a wave
// walking 16 rows one after the other reached 1.2 TB/s), and fat workgroups keep the atomic count at
// 2*cols per CU.
// dx_drop (optional): a second output = dropout mask (seed, call, p) applied to dx.  dx is the gradient of the residual stream in
// front of this norm; the block that produced that stream is `x_prev + dropout(f(..))`, so its backward starts by masking dx --
// here, while dx is in registers, instead of in a launch of its own.
template <typename T, int NCH, int LNB_WAVES>
__global__ __launch_bounds__(LNB_WAVES * 64) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ gamma,
                                                      const float* __restrict__ mean, const float* __restrict__ rstd, T* __restrict__ dx,
                                                      float* __restrict__ dgamma, float* __restrict__ dbeta, const T* __restrict__ skip, int rows,
                                                      T* __restrict__ dx_drop, float p_drop, const int64_t* __restrict__ seed, uint64_t call) {
    constexpr int cols = NCH * 256;
    __shared__ float red[2][LNB_WAVES][cols];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float ag[NCH][4], ab[NCH][4], g[NCH][4];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        V4<float>::load(gamma + c * 256 + lane * 4, g[c]);
#pragma unroll
        for (int i = 0; i < 4; ++i) ag[c][i] = ab[c][i] = 0.f;
    }
    for (int row0 = (blockIdx.x * LNB_WAVES + wave) * 2; row0 < rows; row0 += gridDim.x * LNB_WAVES * 2) {
        const int nr = min(2, rows - row0);
        float d[2][NCH][4], xh[2][NCH][4], sk[2][NCH][4], mu[2], rs[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int row = row0 + (r < nr ? r : 0);   // an odd tail re-reads row0 (weight 0 below)
            mu[r] = mean[row]; rs[r] = rstd[row];
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                V4<T>::load(dy + (int64_t)row * cols + c * 256 + lane * 4, d[r][c]);
                V4<T>::load(x + (int64_t)row * cols + c * 256 + lane * 4, xh[r][c]);
                if (skip) V4<T>::load(skip + (int64_t)row * cols + c * 256 + lane * 4, sk[r][c]);
            }
        }
        float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const float w = r < nr ? 1.f : 0.f;
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    xh[r][c][i] = (xh[r][c][i] - mu[r]) * rs[r];
                    ag[c][i] += w * d[r][c][i] * xh[r][c][i];
                    ab[c][i] += w * d[r][c][i];
                    const float dg = d[r][c][i] * g[c][i];
                    s1[r] = __fmaf_rn(d[r][c][i], g[c][i], s1[r]); s2[r] = __fmaf_rn(dg, xh[r][c][i], s2[r]);   // (explicit fma here and below -- the compiler may or may not contract `s1 += dg` --: tlayer.hip's backward writes the same bits)
                }
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) { s1[r] = wave::sum64_desc(s1[r]); s2[r] = wave::sum64_desc(s2[r]); }   // (four independent chains)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            if (r >= nr) break;
            const float m1 = s1[r] * (1.f / cols), m2 = s2[r] * (1.f / cols);
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                float o[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float t = __fmaf_rn(-xh[r][c][i], m2, __fmaf_rn(d[r][c][i], g[c][i], -m1));
                    o[i] = skip ? __fmaf_rn(rs[r], t, sk[r][c][i]) : rs[r] * t;   // pre-norm residual: the skip branch's gradient joins here (no separate add kernel)
                }
                V4<T>::store(dx + (int64_t)(row0 + r) * cols + c * 256 + lane * 4, o);
                if (dx_drop) {
                    const uint64_t e0 = (uint64_t)(row0 + r) * cols + c * 256 + lane * 4, sd = (uint64_t)seed[0];
                    const uint32_t thr = drop_thr16(p_drop);
                    const float inv = 1.f / (1.f - p_drop);
                    const uint64_t hh = drop_hash4(sd, call, e0 >> 2);   // (e0 is a multiple of 4: one hash for the lane's 4 columns, common.h)
                    float m[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        // (mask the value as ralf_dropout would see it: rounded to T first)
                        m[i] = drop_keep(hh, i, thr) ? (float)(T)o[i] * inv : 0.f;
                    }
                    V4<T>::store(dx_drop + (int64_t)(row0 + r) * cols + c * 256 + lane * 4, m);
                }
            }
        }
    }
    if (dgamma) {
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) { red[0][wave][c * 256 + lane * 4 + i] = ag[c][i]; red[1][wave][c * 256 + lane * 4 + i] = ab[c][i]; }
        __syncthreads();
        for (int c = threadIdx.x; c < 2 * cols; c += LNB_WAVES * 64) {
            const int which = c / cols, cc = c - which * cols;
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < LNB_WAVES; ++w) t += red[which][w][cc];
            atomicAdd((which ? dbeta : dgamma) + cc, t);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// column sums over rows of a [rows, cols] matrix (bias gradients): out[c] += sum_r x[r][c]
// ---------------------------------------------------------------------------------------------
// thread (tx, ty): one 16-byte vector of consecutive columns, rows ty, ty+RPI, ... of its row chunk; TPR = threads per row
template <typename T, int TPR>
__global__ __launch_bounds__(256) void colsum4_kernel(const T* __restrict__ x, int64_t ld, float* __restrict__ out, int rows, int cols, int rows_per_wg) {
    constexpr int RPI = 256 / TPR, N = VL<T>::N;
    __shared__ float red[RPI][TPR * N];
    const int tx = threadIdx.x % TPR, ty = threadIdx.x / TPR;
    const int c = (blockIdx.x * TPR + tx) * N;
    const int r0 = blockIdx.y * rows_per_wg, r1 = min(rows, r0 + rows_per_wg);
    float s[N];
#pragma unroll
    for (int i = 0; i < N; ++i) s[i] = 0.f;
    if (c < cols) {
        int r = r0 + ty;
        for (; r + 3 * RPI < r1; r += 4 * RPI) {   // four independent 16-byte loads in flight
            float v[4][N];
#pragma unroll
            for (int u = 0; u < 4; ++u) VL<T>::load(x + (int64_t)(r + u * RPI) * ld + c, v[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < N; ++i) s[i] += v[u][i];
        }
        for (; r < r1; r += RPI) {
            float v[N];
            VL<T>::load(x + (int64_t)r * ld + c, v);
#pragma unroll
            for (int i = 0; i < N; ++i) s[i] += v[i];
        }
    }
#pragma unroll
    for (int i = 0; i < N; ++i) red[ty][tx * N + i] = s[i];
    __syncthreads();
    for (int j = threadIdx.x; j < TPR * N; j += 256) {
        const int cc = blockIdx.x * TPR * N + j;
        if (cc < cols) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < RPI; ++w) t += red[w][j];
            atomicAdd(out + cc, t);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, int64_t ld, float* __restrict__ out, int rows, int cols, int rows_per_wg) {
    __shared__ float red[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    const int r0 = blockIdx.y * rows_per_wg, r1 = min(rows, r0 + rows_per_wg);
    float s = 0.f;
    if (c < cols)
        for (int r = r0 + ty; r < r1; r += 4) s += (float)x[(int64_t)r * ld + c];
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && c < cols) atomicAdd(out + c, red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx]);
}

// ---------------------------------------------------------------------------------------------
// BatchNorm2d on NHWC viewed as [M = B*H*W, C]; a thread owns 4 consecutive channels
// ---------------------------------------------------------------------------------------------
// 16-byte vector of a row: 8 bf16 or 4 fp32 channels
template <typename T> struct VW;
template <> struct VW<float> {
    static constexpr int N = 4;
    static __device__ __forceinline__ void load(const float* p, float (&v)[4]) { V4<float>::load(p, v); }
    static __device__ __forceinline__ void store(float* p, const float (&v)[4]) { V4<float>::store(p, v); }
};
template <> struct VW<bf16> {
    static constexpr int N = 8;
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    static __device__ __forceinline__ void store(bf16* p, const float (&v)[8]) {
        bf16x8 t;
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = (bf16)v[i];
        *reinterpret_cast<bf16x8*>(p) = t;
    }
    static __device__ __forceinline__ void load(const bf16* p, float (&v)[8]) {
        const bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (float)t[i];
    }
};

// mode 0: s1 = sum x, s2 = sum x^2          (forward statistics)
// mode 1: s1 = sum g, s2 = sum g * xhat     (backward), g = dy masked by (y > 0) when relu
// A thread owns one 16-byte channel vector and walks rows with a 4-deep unrolled stride loop (4 x 16 B of
// every operand in flight per thread: enough outstanding loads to approach the HBM rate).
// RM (MODE 1): 0 no ReLU, 1 ReLU from the 1-bit mask, 2 ReLU from y.  Compile-time: with the choice made by uniform branches INSIDE the
// unrolled row loop the compiler put an s_waitcnt vmcnt(0) behind every row's loads (one row in flight instead of four: 3.4 TB/s).
// All loads are unconditional on a clamped row; rows beyond M are masked out of the sums.
template <typename T, int MODE, int RM = 0>
__global__ __launch_bounds__(256) void bn_reduce_kernel(const T* __restrict__ x, const T* __restrict__ dy, const T* __restrict__ y,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         float* __restrict__ part, int64_t M, int C, int relu, const uint8_t* __restrict__ mask) {
    constexpr int NV = VW<T>::N, UN = 4;
    __shared__ float red[2][256][NV];
    const int cv = C / NV;                       // channel vectors per row
    const int tpr = cv < 256 ? cv : 256;         // threads per row (power of two or 256)
    const int rpi = 256 / tpr;                   // rows per iteration
    const int tx = threadIdx.x % tpr, ty = threadIdx.x / tpr;
    const int c0 = (blockIdx.y * 256 + tx) * NV;
    float a1[NV], a2[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) a1[i] = a2[i] = 0.f;
    if (c0 < C) {
        float mu[NV], rs[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) { mu[i] = MODE == 1 ? mean[c0 + i] : 0.f; rs[i] = MODE == 1 ? rstd[c0 + i] : 1.f; }
        const int64_t stride = (int64_t)gridDim.x * rpi;
        for (int64_t r0 = (int64_t)blockIdx.x * rpi + ty; r0 < M; r0 += stride * UN) {
            float xv[UN][NV], gv[UN][NV], yv[UN][NV];
            uint32_t mb[UN];   // ReLU mask bits of this vector (bit i = element i kept)
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int64_t r = r0 + u * stride, rc = r < M ? r : M - 1;
                const int64_t L = rc * C + c0;
                VW<T>::load(x + L, xv[u]);
                if (MODE == 1) VW<T>::load(dy + L, gv[u]);
                if (MODE == 1 && RM == 1) mb[u] = (uint32_t)mask[L >> 3] >> (NV == 8 ? 0 : (int)(L & 4));
                if (MODE == 1 && RM == 2) VW<T>::load(y + L, yv[u]);
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const bool valid = r0 + u * stride < M;
                uint32_t bits = 0xFFu;
                if (MODE == 1 && RM == 1) bits = mb[u];
                if (MODE == 1 && RM == 2) {
                    bits = 0u;
#pragma unroll
                    for (int i = 0; i < NV; ++i) bits |= (yv[u][i] > 0.f ? 1u : 0u) << i;
                }
                if (!valid) bits = 0u;
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    if (MODE == 0) { const float v = valid ? xv[u][i] : 0.f; a1[i] += v; a2[i] += v * v; }
                    else {
                        const float g = ((bits >> i) & 1u) ? gv[u][i] : 0.f;
                        a1[i] += g; a2[i] += g * (xv[u][i] - mu[i]) * rs[i];
                    }
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) { red[0][threadIdx.x][i] = a1[i]; red[1][threadIdx.x][i] = a2[i]; }
    __syncthreads();
    if (ty == 0 && c0 < C) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            float t1 = 0.f, t2 = 0.f;
            for (int j = 0; j < rpi; ++j) { t1 += red[0][j * tpr + tx][i]; t2 += red[1][j * tpr + tx][i]; }
            // per-workgroup partials (plain stores): same-address fp32 atomics from hundreds of workgroups
            // serialise in L2 and dominated this kernel (0.3 TB/s); a tiny second kernel sums the partials
            part[((int64_t)blockIdx.x * 2 + 0) * C + c0 + i] = t1;
            part[((int64_t)blockIdx.x * 2 + 1) * C + c0 + i] = t2;
        }
    }
}

// s1[c] += sum_b part[b][0][c], s2[c] += sum_b part[b][1][c]   (deterministic order)
// 64 channels per workgroup, the 4 waves split the partial blocks (8 independent loads in flight per thread)
__global__ __launch_bounds__(256) void bn_partial_sum_kernel(const float* __restrict__ part, int nblk, int C, float* __restrict__ s1, float* __restrict__ s2) {
    __shared__ float red[2][4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    float a = 0.f, b = 0.f;
    if (c < C) {
#pragma unroll 8
        for (int i = ty; i < nblk; i += 4) { a += part[((int64_t)i * 2) * C + c]; b += part[((int64_t)i * 2 + 1) * C + c]; }
    }
    red[0][ty][tx] = a; red[1][ty][tx] = b;
    __syncthreads();
    if (ty == 0 && c < C) {
        s1[c] += red[0][0][tx] + red[0][1][tx] + red[0][2][tx] + red[0][3][tx];
        s2[c] += red[1][0][tx] + red[1][1][tx] + red[1][2][tx] + red[1][3][tx];
    }
}

// training: batch mean / biased var -> (mean, rstd, scale, shift) and running-stat update (unbiased var)
// eval: scale/shift from the running statistics
__global__ void bn_finalize_kernel(const float* __restrict__ s1, const float* __restrict__ s2, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* __restrict__ running_mean, float* __restrict__ running_var,
                                   float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ scale, float* __restrict__ shift,
                                   int64_t M, int C, float eps, float momentum, int training) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float mu, var;
    if (training) {
        mu = s1[c] / (float)M;
        var = fmaxf(s2[c] / (float)M - mu * mu, 0.f);
        if (running_mean) {
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * var * ((float)M / (float)(M > 1 ? M - 1 : 1));
        }
    } else {
        mu = running_mean[c];
        var = running_var[c];
    }
    const float rs = rsqrtf(var + eps);
    mean[c] = mu; rstd[c] = rs;
    scale[c] = gamma[c] * rs;
    shift[c] = beta[c] - mu * gamma[c] * rs;
}

// training-mode statistics in one step after the partial sums: per-channel sum of the per-workgroup partials,
// mean / rstd / scale / shift, running-stat update and the num_batches_tracked counter (one launch instead of
// partial-sum + finalize + a host-issued counter increment, and no zero-fill of the sums)
// WAVES = 4, or 16 for a few hundred partial rows: one launch instead of fold + finalize (two dependent launches of ~5 us of work
// each, on the forward critical path of every BatchNorm of layer2 / layer3)
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void bn_partial_finalize_kernel(const float* __restrict__ part, int nblk, const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta, float* __restrict__ running_mean,
                                                                   float* __restrict__ running_var, int64_t* __restrict__ counter,
                                                                   float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ scale,
                                                                   float* __restrict__ shift, int64_t M, int C, float eps, float momentum) {
    __shared__ float red[2][WAVES][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    float a = 0.f, b = 0.f;
    if (c < C) {
#pragma unroll 8
        for (int i = ty; i < nblk; i += WAVES) { a += part[((int64_t)i * 2) * C + c]; b += part[((int64_t)i * 2 + 1) * C + c]; }
    }
    red[0][ty][tx] = a; red[1][ty][tx] = b;
    __syncthreads();
    if (ty == 0 && c < C) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) { s1 += red[0][w][tx]; s2 += red[1][w][tx]; }
        const float mu = s1 / (float)M;
        const float var = fmaxf(s2 / (float)M - mu * mu, 0.f);
        if (running_mean) {
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * var * ((float)M / (float)(M > 1 ? M - 1 : 1));
        }
        const float rs = rsqrtf(var + eps);
        mean[c] = mu; rstd[c] = rs;
        scale[c] = gamma[c] * rs;
        shift[c] = beta[c] - mu * gamma[c] * rs;
    }
    if (counter && blockIdx.x == 0 && threadIdx.x == 0) *counter += 1;
}

// backward statistics from the partial rows a data-gradient GEMM wrote (RalfGemmDesc.bnb_part: [nblk][2][C], sums of dz and of dz * (x - mean)):
// s1[c] += sum_b part[b][0][c],  s2[c] += rstd[c] * sum_b part[b][1][c]   (= sum dz * xhat); deterministic order
// coef (optional, fp32 [3][C]): the affine form of the backward apply, dx = c1 * dz + c2 * x + c3 with c1 = gamma rstd,
// c2 = -c1 rstd (sum dz xhat) / M, c3 = -c1 (sum dz) / M - c2 mean (ralf_bn_bwd_apply_affine, RalfGemmDesc.at_mode 2); needs s1 / s2 to
// hold nothing but this layer's sums (they are the zero-initialised dbeta / dgamma views)
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void bn_bwd_partial_sum_kernel(const float* __restrict__ part, int nblk, int C, const float* __restrict__ rstd,
                                                                  float* __restrict__ s1, float* __restrict__ s2, const float* __restrict__ gamma,
                                                                  const float* __restrict__ mean, float invM, float* __restrict__ coef) {
    __shared__ float red[2][WAVES][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    float a = 0.f, b = 0.f;
    if (c < C) {
#pragma unroll 8
        for (int i = ty; i < nblk; i += WAVES) { a += part[((int64_t)i * 2) * C + c]; b += part[((int64_t)i * 2 + 1) * C + c]; }
    }
    red[0][ty][tx] = a; red[1][ty][tx] = b;
    __syncthreads();
    if (ty == 0 && c < C) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) { t1 += red[0][w][tx]; t2 += red[1][w][tx]; }
        s1[c] += t1;
        s2[c] += rstd[c] * t2;
        if (coef) {
            const float rs = rstd[c], c1 = gamma[c] * rs, c2 = -c1 * rs * (rs * t2) * invM;
            coef[c] = c1; coef[C + c] = c2; coef[2 * C + c] = -c1 * t1 * invM - c2 * mean[c];
        }
    }
}

// [nrows][2][C] partial sums -> [G][2][C] (row r goes to group r % G): first stage when the GEMM wrote thousands of partial rows
__global__ __launch_bounds__(256) void bn_partial_fold_kernel(const float* __restrict__ part, int nrows, int C, int G, float* __restrict__ out) {
    __shared__ float red[2][4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx, g = blockIdx.y;
    float a = 0.f, b = 0.f;
    if (c < C) {
#pragma unroll 4
        for (int i = g + ty * G; i < nrows; i += 4 * G) { a += part[((int64_t)i * 2) * C + c]; b += part[((int64_t)i * 2 + 1) * C + c]; }
    }
    red[0][ty][tx] = a; red[1][ty][tx] = b;
    __syncthreads();
    if (ty == 0 && c < C) {
        out[((int64_t)g * 2) * C + c] = red[0][0][tx] + red[0][1][tx] + red[0][2][tx] + red[0][3][tx];
        out[((int64_t)g * 2 + 1) * C + c] = red[1][0][tx] + red[1][1][tx] + red[1][2][tx] + red[1][3][tx];
    }
}

// y = relu?( x*scale + shift (+ res) ); 8 consecutive elements per thread.  relu_mask (optional, 1 bit per element, bit i of
// byte j <-> element 8j+i): the backward kernels read it instead of y (1/16 of the bytes).
// HOIST: the grid stride (gridDim.x * 256 * 8 elements) is a multiple of C, so a thread sees the SAME 8 channels in every
// iteration and loads their coefficients once (they were 4 of the 6 vector loads per iteration: the kernel ran at the L1 rate).
template <typename T, bool HOIST>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                                                        const T* __restrict__ res, T* __restrict__ y, uint8_t* __restrict__ relu_mask,
                                                        int64_t total8, int C, int relu) {
    constexpr int NV = VW<T>::N, H = 8 / NV;
    float sc[8], sh[8];
    auto coef = [&](int c0) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            VW<float>::load(scale + c0 + 4 * q, *reinterpret_cast<float (*)[4]>(&sc[4 * q]));
            VW<float>::load(shift + c0 + 4 * q, *reinterpret_cast<float (*)[4]>(&sh[4 * q]));
        }
    };
    const int64_t e0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (HOIST && e0 < total8) coef((int)((e0 * 8) % C));
    for (int64_t e = e0; e < total8; e += (int64_t)gridDim.x * 256) {
        if (!HOIST) coef((int)((e * 8) % C));
        uint32_t bits = 0;
#pragma unroll
        for (int h = 0; h < H; ++h) {
            float xv[NV], o[NV];
            VW<T>::load(x + e * 8 + h * NV, xv);
#pragma unroll
            for (int i = 0; i < NV; ++i) o[i] = __builtin_fmaf(xv[i], sc[h * NV + i], sh[h * NV + i]);   // (the same fma as the operand-transform loader, gemm_impl.h)
            if (res) {
                float rv[NV];
                VW<T>::load(res + e * 8 + h * NV, rv);
#pragma unroll
                for (int i = 0; i < NV; ++i) o[i] += rv[i];
            }
            if (relu) {
#pragma unroll
                for (int i = 0; i < NV; ++i) { bits |= (o[i] > 0.f ? 1u : 0u) << (h * NV + i); o[i] = fmaxf(o[i], 0.f); }
            }
            VW<T>::store(y + e * 8 + h * NV, o);
        }
        if (relu && relu_mask) relu_mask[e] = (uint8_t)bits;
    }
}

// dx = gamma*rstd*(g - sum_g/M - xhat*sum_gx/M);  dres = g (gradient of the fused residual add); 8 elements per thread
template <typename T, bool HOIST>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ dy, const T* __restrict__ y,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                            const float* __restrict__ s1, const float* __restrict__ s2, T* __restrict__ dx, T* __restrict__ dres,
                                                            int64_t total8, int C, float invM, int relu, const uint8_t* __restrict__ mask) {
    constexpr int NV = VW<T>::N, H = 8 / NV;
    float mu[8], rs[8], gm[8], a1[8], a2[8];   // HOIST (see bn_apply_kernel): loaded once per thread instead of 10 vector loads per iteration
    auto coef = [&](int c0) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int c = c0 + 4 * q;
            VW<float>::load(mean + c, *reinterpret_cast<float (*)[4]>(&mu[4 * q])); VW<float>::load(rstd + c, *reinterpret_cast<float (*)[4]>(&rs[4 * q]));
            VW<float>::load(gamma + c, *reinterpret_cast<float (*)[4]>(&gm[4 * q]));
            VW<float>::load(s1 + c, *reinterpret_cast<float (*)[4]>(&a1[4 * q])); VW<float>::load(s2 + c, *reinterpret_cast<float (*)[4]>(&a2[4 * q]));
        }
    };
    const int64_t e0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (HOIST && e0 < total8) coef((int)((e0 * 8) % C));
    for (int64_t e = e0; e < total8; e += (int64_t)gridDim.x * 256) {
        if (!HOIST) coef((int)((e * 8) % C));
        const uint32_t mbyte = (relu && mask) ? (uint32_t)mask[e] : 0xFFu;
#pragma unroll
        for (int h = 0; h < H; ++h) {
            const int64_t off = e * 8 + h * NV;
            float xv[NV], g[NV], o[NV];
            VW<T>::load(x + off, xv);
            VW<T>::load(dy + off, g);
            if (relu && mask) {
#pragma unroll
                for (int i = 0; i < NV; ++i) g[i] = ((mbyte >> (h * NV + i)) & 1u) ? g[i] : 0.f;
            } else if (relu) {
                float yv[NV];
                VW<T>::load(y + off, yv);
#pragma unroll
                for (int i = 0; i < NV; ++i) g[i] = yv[i] > 0.f ? g[i] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = h * NV + i;
                const float xh = (xv[i] - mu[c]) * rs[c];
                o[i] = gm[c] * rs[c] * (g[i] - a1[c] * invM - xh * a2[c] * invM);
            }
            VW<T>::store(dx + off, o);
            if (dres) VW<T>::store(dres + off, g);
        }
    }
}

// dx = c1 * dz + (c2 * x + c3): the backward apply with the per-channel work folded into three coefficients (bn_bwd_partial_sum_kernel);
// the stand-alone form of RalfGemmDesc.at_mode 2, same arithmetic
template <typename T, bool HOIST>
__global__ __launch_bounds__(256) void bn_bwd_apply_affine_kernel(const T* __restrict__ dz, const T* __restrict__ x, const float* __restrict__ c1p,
                                                                   const float* __restrict__ c2p, const float* __restrict__ c3p, T* __restrict__ dx,
                                                                   int64_t total8, int C) {
    constexpr int NV = VW<T>::N, H = 8 / NV;
    float c1[8], c2[8], c3[8];
    auto coef = [&](int c0) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            VW<float>::load(c1p + c0 + 4 * q, *reinterpret_cast<float (*)[4]>(&c1[4 * q]));
            VW<float>::load(c2p + c0 + 4 * q, *reinterpret_cast<float (*)[4]>(&c2[4 * q]));
            VW<float>::load(c3p + c0 + 4 * q, *reinterpret_cast<float (*)[4]>(&c3[4 * q]));
        }
    };
    const int64_t e0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (HOIST && e0 < total8) coef((int)((e0 * 8) % C));
    for (int64_t e = e0; e < total8; e += (int64_t)gridDim.x * 256) {
        if (!HOIST) coef((int)((e * 8) % C));
#pragma unroll
        for (int h = 0; h < H; ++h) {
            const int64_t off = e * 8 + h * NV;
            float xv[NV], g[NV], o[NV];
            VW<T>::load(x + off, xv);
            VW<T>::load(dz + off, g);
#pragma unroll
            for (int i = 0; i < NV; ++i) o[i] = __builtin_fmaf(g[i], c1[h * NV + i], __builtin_fmaf(xv[i], c2[h * NV + i], c3[h * NV + i]));
            VW<T>::store(dx + off, o);
        }
    }
}

// grid of a flat 8-elements-per-thread kernel: `hoist` = its stride is a multiple of C (per-thread channels loop-invariant)
inline int flat_grid(int64_t total8, int C, bool* hoist) {
    const int64_t need = (total8 + 255) / 256;
    const int grid = (int)(need < 1 ? 1 : (need > 4096 ? 4096 : need));
    *hoist = need <= 4096 || ((int64_t)grid * 2048) % C == 0;
    return grid;
}

inline int grid_for(int64_t work_items, int per_block, int cap = 2048) {
    int64_t b = (work_items + per_block - 1) / per_block;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

#define DISPATCH_T(dtype, ...)                                            \
    do {                                                                  \
        if ((dtype) == RALF_F32) { typedef float T; __VA_ARGS__; }        \
        else if ((dtype) == RALF_BF16) { typedef bf16 T; __VA_ARGS__; }   \
        else { ralf::set_error("bad dtype %d", (dtype)); return RALF_ERR_INVALID; } \
    } while (0)

extern "C" int ralf_layernorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                                  int rows, int cols, float eps, void* stream) {
    RALF_REQUIRE(x && gamma && beta && y, "layernorm_fwd: null pointer");
    RALF_REQUIRE(rows > 0 && cols % 256 == 0 && cols <= 1024, "layernorm_fwd: cols=%d must be a multiple of 256, <= 1024", cols);
    hipStream_t st = (hipStream_t)stream;
    const int grid = grid_for(rows, 4);
    DISPATCH_T(dtype, switch (cols / 256) {
        case 1: hipLaunchKernelGGL((ln_fwd_kernel<T, 1>), dim3(grid), dim3(256), 0, st, (const T*)x, gamma, beta, (T*)y, mean, rstd, rows, eps); break;
        case 2: hipLaunchKernelGGL((ln_fwd_kernel<T, 2>), dim3(grid), dim3(256), 0, st, (const T*)x, gamma, beta, (T*)y, mean, rstd, rows, eps); break;
        case 3: hipLaunchKernelGGL((ln_fwd_kernel<T, 3>), dim3(grid), dim3(256), 0, st, (const T*)x, gamma, beta, (T*)y, mean, rstd, rows, eps); break;
        default: hipLaunchKernelGGL((ln_fwd_kernel<T, 4>), dim3(grid), dim3(256), 0, st, (const T*)x, gamma, beta, (T*)y, mean, rstd, rows, eps); break;
    });
    return ralf::check_launch("layernorm_fwd");
}

/* dgamma/dbeta (fp32 [cols]) are ACCUMULATED into (caller zeroes or carries gradients); may be NULL.
 * skip (dtype [rows, cols], may be NULL) is added to dx: gradient of the residual branch x -> (LN(x), x) */
extern "C" int ralf_layernorm_bwd(int dtype, const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                                  void* dx, float* dgamma, float* dbeta, const void* skip, int rows, int cols,
                                  void* dx_drop, float p_drop, const int64_t* seed, uint64_t call_id, void* stream) {
    RALF_REQUIRE(dy && x && gamma && mean && rstd && dx, "layernorm_bwd: null pointer");
    RALF_REQUIRE(!dx_drop || (seed && p_drop > 0.f && p_drop < 1.f), "layernorm_bwd: dx_drop needs a seed and 0 < p < 1");
    RALF_REQUIRE(rows > 0 && cols % 256 == 0 && cols <= 512, "layernorm_bwd: cols=%d must be 256 or 512", cols);
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype, if (cols == 256)
                   hipLaunchKernelGGL((ln_bwd_kernel<T, 1, 16>), dim3(grid_for(rows, 32, 256)), dim3(1024), 0, st, (const T*)dy, (const T*)x, gamma, mean, rstd, (T*)dx, dgamma, dbeta, (const T*)skip, rows, (T*)dx_drop, p_drop, seed, call_id);
               else
                   hipLaunchKernelGGL((ln_bwd_kernel<T, 2, 8>), dim3(grid_for(rows, 16, 512)), dim3(512), 0, st, (const T*)dy, (const T*)x, gamma, mean, rstd, (T*)dx, dgamma, dbeta, (const T*)skip, rows, (T*)dx_drop, p_drop, seed, call_id););
    return ralf::check_launch("layernorm_bwd");
}

/* out[c] += sum_r x[r*ld + c]   (fp32 accumulate into out) */
// many column sums in one launch: job j = (matrix, rows, cols % 256 == 0); workgroup <-> (job, 256-column block, row chunk)
struct CSJob { const bf16* x; float* out; int64_t ld; int rows, rpw, cb, first; };
constexpr int CS_MAX = 64;
struct CSParams { int njobs; int pad[3]; CSJob j[CS_MAX]; };
__global__ __launch_bounds__(256) void colsum_grouped_kernel(const CSParams G) {
    constexpr int TPR = 32, RPI = 8, N = 8;
    __shared__ float red[RPI][TPR * N];
    const int b = (int)blockIdx.x;
    int ji = 0;
    while (ji + 1 < G.njobs && b >= G.j[ji + 1].first) ++ji;
    const CSJob& J = G.j[ji];
    const int lb = b - J.first, cbi = lb % J.cb, rci = lb / J.cb;
    const int tx = threadIdx.x % TPR, ty = threadIdx.x / TPR;
    const int c = (cbi * TPR + tx) * N;
    const int r0 = rci * J.rpw, r1 = min(J.rows, r0 + J.rpw);
    float s[N];
#pragma unroll
    for (int i = 0; i < N; ++i) s[i] = 0.f;
    int r = r0 + ty;
    for (; r + 3 * RPI < r1; r += 4 * RPI) {
        float v[4][N];
#pragma unroll
        for (int u = 0; u < 4; ++u) VL<bf16>::load(J.x + (int64_t)(r + u * RPI) * J.ld + c, v[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < N; ++i) s[i] += v[u][i];
    }
    for (; r < r1; r += RPI) {
        float v[N];
        VL<bf16>::load(J.x + (int64_t)r * J.ld + c, v);
#pragma unroll
        for (int i = 0; i < N; ++i) s[i] += v[i];
    }
#pragma unroll
    for (int i = 0; i < N; ++i) red[ty][tx * N + i] = s[i];
    __syncthreads();
    {
        const int j = threadIdx.x;   // 256 columns, 256 threads
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < RPI; ++w) t += red[w][j];
        atomicAdd(J.out + cbi * TPR * N + j, t);
    }
}

extern "C" int ralf_colsum_grouped(const RalfColsumJob* jobs, int njobs, int dtype, void* stream) {
    RALF_REQUIRE(jobs && njobs > 0 && dtype == RALF_BF16, "colsum_grouped: bf16 jobs only");
    for (int j0 = 0; j0 < njobs; j0 += CS_MAX) {
        const int n = njobs - j0 < CS_MAX ? njobs - j0 : CS_MAX;
        CSParams G;
        G.njobs = n;
        int first = 0;
        for (int i = 0; i < n; ++i) {
            const RalfColsumJob& w = jobs[j0 + i];
            RALF_REQUIRE(w.x && w.out && w.rows > 0 && w.cols > 0 && w.cols % 256 == 0 && w.ld % 8 == 0 && ((uintptr_t)w.x % 16) == 0,
                         "colsum_grouped: job %d: cols %% 256, ld %% 8, 16-byte alignment", j0 + i);
            CSJob& J = G.j[i];
            J.x = (const bf16*)w.x; J.out = w.out; J.ld = w.ld; J.rows = w.rows; J.cb = w.cols / 256;
            int rchunks = 32768 / w.cols;   // <= ~32 k same-address atomics per job (see ralf_colsum)
            rchunks = rchunks < 16 ? 16 : (rchunks > 256 ? 256 : rchunks);
            int rpw = ceil_div(w.rows, rchunks);
            if (rpw < 32) rpw = 32;
            J.rpw = rpw;
            J.first = first;
            first += J.cb * ceil_div(w.rows, rpw);
        }
        hipLaunchKernelGGL(colsum_grouped_kernel, dim3(first), dim3(256), 0, (hipStream_t)stream, G);
    }
    return ralf::check_launch("colsum_grouped");
}

extern "C" int ralf_colsum(int dtype, const void* x, int64_t ld, float* out, int rows, int cols, void* stream) {
    RALF_REQUIRE(x && out && rows > 0 && cols > 0, "colsum: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int esz = dtype == RALF_F32 ? 4 : 2;
    const int vw = 16 / esz;
    if (cols % vw == 0 && ld % vw == 0 && ((uintptr_t)x % 16) == 0) {   // vector path: 16 bytes of columns per thread
        const int cv = cols / vw;
        const int tpr = cv >= 64 ? 64 : cv >= 32 ? 32 : 16;
        const int cb = ceil_div(cv, tpr), rpi = 256 / tpr;
        // one fp32 atomic per column per workgroup: same-address atomics run at ~19 G/s, so the row chunks are sized for
        // <= ~32 k atomics in total (1024 chunks on a [16384, 1024] input spent 14 us in atomics alone)
        int rchunks = 32768 / (cols > 0 ? cols : 1);
        rchunks = rchunks < 16 ? 16 : (rchunks > 256 ? 256 : rchunks);
        int rpw = ceil_div(rows, rchunks);
        if (rpw < rpi * 4) rpw = rpi * 4;
        rchunks = ceil_div(rows, rpw);
        DISPATCH_T(dtype, switch (tpr) {
            case 64: hipLaunchKernelGGL((colsum4_kernel<T, 64>), dim3(cb, rchunks), dim3(256), 0, st, (const T*)x, ld, out, rows, cols, rpw); break;
            case 32: hipLaunchKernelGGL((colsum4_kernel<T, 32>), dim3(cb, rchunks), dim3(256), 0, st, (const T*)x, ld, out, rows, cols, rpw); break;
            default: hipLaunchKernelGGL((colsum4_kernel<T, 16>), dim3(cb, rchunks), dim3(256), 0, st, (const T*)x, ld, out, rows, cols, rpw); break;
        });
        return ralf::check_launch("colsum");
    }
    const int cb = ceil_div(cols, 64);
    int rchunks = 1024 / cb;
    if (rchunks < 1) rchunks = 1;
    int rpw = ceil_div(rows, rchunks);
    if (rpw < 64) rpw = 64;
    rchunks = ceil_div(rows, rpw);
    DISPATCH_T(dtype, hipLaunchKernelGGL((colsum_kernel<T>), dim3(cb, rchunks), dim3(256), 0, st, (const T*)x, ld, out, rows, cols, rpw));
    return ralf::check_launch("colsum");
}

/* sums s1,s2 (fp32 [C]) must be zero on entry. */
static const int BN_ROW_GROUPS = [] { const char* e = getenv("RALF_BN_GROUPS"); int v = e ? atoi(e) : 512; return v < 1 ? 1 : (v > RALF_BN_MAX_PARTIALS ? RALF_BN_MAX_PARTIALS : v); }();   // tuning knob
static int bn_reduce_geom(int dtype, int64_t M, int C, int* gx, int* gy) {
    const int nv = dtype == RALF_F32 ? 4 : 8;
    if (C % nv) return -1;
    const int cv = C / nv;
    if (!(((cv & (cv - 1)) == 0 && cv <= 256) || cv % 256 == 0)) return -1;
    const int tpr = cv < 256 ? cv : 256, rpi = 256 / tpr;
    *gy = ceil_div(cv, 256);
    *gx = grid_for(M, rpi * 16, BN_ROW_GROUPS);   // workgroups along rows (2 per CU), 4 x 16 B of every operand per thread in flight
    return 0;
}

/* s1 += sum x, s2 += sum x^2 (fp32 [C]); workspace: RALF_BN_MAX_PARTIALS * 2 * C floats */
extern "C" int ralf_bn_stats(int dtype, const void* x, float* s1, float* s2, int64_t M, int C, float* workspace, void* stream) {
    RALF_REQUIRE(x && s1 && s2 && workspace && M > 0, "bn_stats: bad arguments");
    int gx, gy;
    RALF_REQUIRE(bn_reduce_geom(dtype, M, C, &gx, &gy) == 0, "bn_stats: C=%d unsupported (needs C/vec a power of two <= 256 or a multiple of 256)", C);
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype, hipLaunchKernelGGL((bn_reduce_kernel<T, 0>), dim3(gx, gy), dim3(256), 0, st, (const T*)x, nullptr, nullptr, nullptr, nullptr, workspace, M, C, 0, nullptr));
    hipLaunchKernelGGL(bn_partial_sum_kernel, dim3(ceil_div(C, 64)), dim3(256), 0, st, workspace, gx, C, s1, s2);
    return ralf::check_launch("bn_stats");
}

// eval-mode scale / shift of many BatchNorm layers: one workgroup per layer
__global__ __launch_bounds__(256) void bn_fold_batched_kernel(const RalfBnFoldJob* __restrict__ jobs, float eps) {
    const RalfBnFoldJob j = jobs[blockIdx.x];
    for (int c = threadIdx.x; c < j.C; c += 256) {
        const float sc = j.gamma[c] * rsqrtf(j.var[c] + eps);
        j.scale[c] = sc;
        j.shift[c] = j.beta[c] - j.mean[c] * sc;
    }
}

extern "C" int ralf_bn_fold_batched(const RalfBnFoldJob* jobs_device, int njobs, float eps, void* stream) {
    RALF_REQUIRE(jobs_device && njobs > 0, "bn_fold_batched: bad arguments");
    hipLaunchKernelGGL(bn_fold_batched_kernel, dim3(njobs), dim3(256), 0, (hipStream_t)stream, jobs_device, eps);
    return ralf::check_launch("bn_fold_batched");
}

extern "C" int ralf_bn_finalize(const float* s1, const float* s2, const float* gamma, const float* beta, float* running_mean, float* running_var,
                                float* mean, float* rstd, float* scale, float* shift, int64_t M, int C, float eps, float momentum, int training, void* stream) {
    RALF_REQUIRE(gamma && beta && mean && rstd && scale && shift, "bn_finalize: null pointer");
    RALF_REQUIRE(training ? (s1 && s2) : (running_mean && running_var), "bn_finalize: missing statistics");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, (hipStream_t)stream, s1, s2, gamma, beta, running_mean, running_var, mean, rstd, scale, shift, M, C, eps, momentum, training);
    return ralf::check_launch("bn_finalize");
}

/* training-mode forward statistics of x [M, C]: (mean, rstd, scale, shift) + running-stat update (+ counter += 1) */
extern "C" int ralf_bn_batch_stats(int dtype, const void* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                                   int64_t* num_batches_tracked, float* mean, float* rstd, float* scale, float* shift, int64_t M, int C,
                                   float eps, float momentum, float* workspace, void* stream) {
    RALF_REQUIRE(x && gamma && beta && mean && rstd && scale && shift && workspace && M > 0, "bn_batch_stats: bad arguments");
    int gx, gy;
    RALF_REQUIRE(bn_reduce_geom(dtype, M, C, &gx, &gy) == 0, "bn_batch_stats: C=%d unsupported (needs C/vec a power of two <= 256 or a multiple of 256)", C);
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype, hipLaunchKernelGGL((bn_reduce_kernel<T, 0>), dim3(gx, gy), dim3(256), 0, st, (const T*)x, nullptr, nullptr, nullptr, nullptr, workspace, M, C, 0, nullptr));
    hipLaunchKernelGGL(bn_partial_finalize_kernel<4>, dim3(ceil_div(C, 64)), dim3(256), 0, st, workspace, gx, gamma, beta, running_mean, running_var,
                       num_batches_tracked, mean, rstd, scale, shift, M, C, eps, momentum);
    return ralf::check_launch("bn_batch_stats");
}

extern "C" int ralf_bn_stats_from_partials(const float* partials, int nrows, const float* gamma, const float* beta, float* running_mean, float* running_var,
                                           int64_t* num_batches_tracked, float* mean, float* rstd, float* scale, float* shift, int64_t M, int C,
                                           float eps, float momentum, float* workspace, void* stream) {
    RALF_REQUIRE(partials && nrows > 0 && gamma && beta && mean && rstd && scale && shift && workspace && M > 0 && C > 0, "bn_stats_from_partials: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const float* src = partials;
    int n = nrows;
    if (nrows > 64 && nrows <= 1024) {   // a few hundred rows: 16 waves walk them in one launch
        hipLaunchKernelGGL(bn_partial_finalize_kernel<16>, dim3(ceil_div(C, 64)), dim3(1024), 0, st, src, n, gamma, beta, running_mean, running_var,
                           num_batches_tracked, mean, rstd, scale, shift, M, C, eps, momentum);
        return ralf::check_launch("bn_stats_from_partials");
    }
    if (nrows > 1024) {   // thousands of rows: fold to 128 rows with a wide grid first (the finalize kernel walks its rows serially)
        const int G = 128;
        hipLaunchKernelGGL(bn_partial_fold_kernel, dim3(ceil_div(C, 64), G), dim3(256), 0, st, partials, nrows, C, G, workspace);
        src = workspace; n = G;
    }
    hipLaunchKernelGGL(bn_partial_finalize_kernel<4>, dim3(ceil_div(C, 64)), dim3(256), 0, st, src, n, gamma, beta, running_mean, running_var,
                       num_batches_tracked, mean, rstd, scale, shift, M, C, eps, momentum);
    return ralf::check_launch("bn_stats_from_partials");
}

extern "C" int ralf_bn_bwd_stats_from_partials(const float* partials, int nrows, const float* rstd, float* s1, float* s2, int C, float* workspace,
                                               const float* gamma, const float* mean, int64_t M, float* coef, void* stream) {
    RALF_REQUIRE(partials && nrows > 0 && rstd && s1 && s2 && workspace && C > 0, "bn_bwd_stats_from_partials: bad arguments");
    RALF_REQUIRE(!coef || (gamma && mean && M > 0), "bn_bwd_stats_from_partials: the coefficient output needs gamma, mean and M");
    const float invM = coef ? 1.f / (float)M : 0.f;
    hipStream_t st = (hipStream_t)stream;
    const float* src = partials;
    int n = nrows;
    if (nrows > 1024) {   // thousands of rows: fold to 128 rows with a wide grid first
        const int G = 128;
        hipLaunchKernelGGL(bn_partial_fold_kernel, dim3(ceil_div(C, 64), G), dim3(256), 0, st, partials, nrows, C, G, workspace);
        src = workspace; n = G;
    }
    if (n > 64) hipLaunchKernelGGL(bn_bwd_partial_sum_kernel<16>, dim3(ceil_div(C, 64)), dim3(1024), 0, st, src, n, C, rstd, s1, s2, gamma, mean, invM, coef);
    else hipLaunchKernelGGL(bn_bwd_partial_sum_kernel<4>, dim3(ceil_div(C, 64)), dim3(256), 0, st, src, n, C, rstd, s1, s2, gamma, mean, invM, coef);
    return ralf::check_launch("bn_bwd_stats_from_partials");
}

extern "C" int ralf_bn_bwd_apply_affine(int dtype, const void* dz, const void* x, const float* c1, const float* c2, const float* c3, void* dx, int64_t M, int C, void* stream) {
    RALF_REQUIRE(dz && x && c1 && c2 && c3 && dx && C % 8 == 0, "bn_bwd_apply_affine: bad arguments (C %% 8 == 0)");
    const int64_t total8 = M * C / 8;
    bool hoist;
    const int grid = flat_grid(total8, C, &hoist);
    if (hoist) DISPATCH_T(dtype, hipLaunchKernelGGL((bn_bwd_apply_affine_kernel<T, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const T*)dz, (const T*)x, c1, c2, c3, (T*)dx, total8, C));
    else DISPATCH_T(dtype, hipLaunchKernelGGL((bn_bwd_apply_affine_kernel<T, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const T*)dz, (const T*)x, c1, c2, c3, (T*)dx, total8, C));
    return ralf::check_launch("bn_bwd_apply_affine");
}

extern "C" int ralf_bn_apply(int dtype, const void* x, const float* scale, const float* shift, const void* res, void* y, uint8_t* relu_mask,
                             int64_t M, int C, int relu, void* stream) {
    RALF_REQUIRE(x && scale && shift && y && C % 8 == 0, "bn_apply: bad arguments (C %% 8 == 0)");
    const int64_t total8 = M * C / 8;
    bool hoist;
    const int grid = flat_grid(total8, C, &hoist);
    if (hoist) DISPATCH_T(dtype, hipLaunchKernelGGL((bn_apply_kernel<T, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const T*)x, scale, shift, (const T*)res, (T*)y, relu_mask, total8, C, relu));
    else DISPATCH_T(dtype, hipLaunchKernelGGL((bn_apply_kernel<T, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const T*)x, scale, shift, (const T*)res, (T*)y, relu_mask, total8, C, relu));
    return ralf::check_launch("bn_apply");
}

/* backward reductions: s1 += sum g, s2 += sum g*xhat; g = dy * (y>0) when relu; workspace as ralf_bn_stats */
extern "C" int ralf_bn_bwd_reduce(int dtype, const void* x, const void* dy, const void* y, const uint8_t* relu_mask, const float* mean, const float* rstd,
                                  float* s1, float* s2, int64_t M, int C, int relu, float* workspace, void* stream) {
    RALF_REQUIRE(x && dy && mean && rstd && s1 && s2 && workspace && (!relu || y || relu_mask), "bn_bwd_reduce: bad arguments");
    int gx, gy;
    RALF_REQUIRE(bn_reduce_geom(dtype, M, C, &gx, &gy) == 0, "bn_bwd_reduce: C=%d unsupported", C);
    hipStream_t st = (hipStream_t)stream;
    const int rm = !relu ? 0 : (relu_mask ? 1 : 2);
    DISPATCH_T(dtype, {
        if (rm == 0) hipLaunchKernelGGL((bn_reduce_kernel<T, 1, 0>), dim3(gx, gy), dim3(256), 0, st, (const T*)x, (const T*)dy, (const T*)y, mean, rstd, workspace, M, C, relu, relu_mask);
        else if (rm == 1) hipLaunchKernelGGL((bn_reduce_kernel<T, 1, 1>), dim3(gx, gy), dim3(256), 0, st, (const T*)x, (const T*)dy, (const T*)y, mean, rstd, workspace, M, C, relu, relu_mask);
        else hipLaunchKernelGGL((bn_reduce_kernel<T, 1, 2>), dim3(gx, gy), dim3(256), 0, st, (const T*)x, (const T*)dy, (const T*)y, mean, rstd, workspace, M, C, relu, relu_mask);
    });
    hipLaunchKernelGGL(bn_partial_sum_kernel, dim3(ceil_div(C, 64)), dim3(256), 0, st, workspace, gx, C, s1, s2);
    return ralf::check_launch("bn_bwd_reduce");
}

extern "C" int ralf_bn_bwd_apply(int dtype, const void* x, const void* dy, const void* y, const uint8_t* relu_mask, const float* mean, const float* rstd,
                                 const float* gamma, const float* s1, const float* s2, void* dx, void* dres, int64_t M, int C, int relu, void* stream) {
    RALF_REQUIRE(x && dy && mean && rstd && gamma && s1 && s2 && dx && (!relu || y || relu_mask), "bn_bwd_apply: bad arguments");
    RALF_REQUIRE(C % 8 == 0, "bn_bwd_apply: C %% 8 == 0");
    const int64_t total8 = M * C / 8;
    bool hoist;
    const int grid = flat_grid(total8, C, &hoist);
    if (hoist) DISPATCH_T(dtype, hipLaunchKernelGGL((bn_bwd_apply_kernel<T, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const T*)x, (const T*)dy, (const T*)y, mean, rstd, gamma, s1, s2, (T*)dx, (T*)dres, total8, C, 1.f / (float)M, relu, relu_mask));
    else DISPATCH_T(dtype, hipLaunchKernelGGL((bn_bwd_apply_kernel<T, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const T*)x, (const T*)dy, (const T*)y, mean, rstd, gamma, s1, s2, (T*)dx, (T*)dres, total8, C, 1.f / (float)M, relu, relu_mask));
    return ralf::check_launch("bn_bwd_apply");
}
