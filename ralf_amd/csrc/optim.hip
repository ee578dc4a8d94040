// Fused optimizer step for gfx950: global grad-norm (sum of squares), clip coefficient on device
// (no host sync), AdamW update on flat fp32 buffers with an optional bf16 shadow copy of the
// weights for the next forward.
//
// Replaces torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW.step as driven by
// image2layout/train/train.py:212-230,449-454 (AdamW groups from BaseModel.optim_groups,
// image2layout/train/models/common/base_model.py:207-347; max_norm = 0.1).
#include "common.h"

namespace {
typedef __bf16 bf16;

__device__ __forceinline__ float wave_sum(float v) {
    return wave::sum64_desc(v);   // the descending butterfly, bit for bit, without the LDS crossbar (wave_ops.h)
}

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ out) {
    __shared__ float red[4];
    float a = 0.f;
    const int64_t n4 = n >> 2;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n4; e += (int64_t)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(g)[e];
        a += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = g[(n4 << 2) + threadIdx.x]; a += v * v; }
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

// deterministic form: partials[b] = sum of squares of block b's elements (fixed assignment), summed in a fixed order by
// clip_coef_partials_kernel.  With the atomic form the clip coefficient differed in its last bit from run to run -- and between
// data-parallel RANKS holding bit-identical averaged gradients, whose weights then drifted apart at rounding level (found by the
// two-ranks-on-one-GPU test); torch's clip_grad_norm_ gives every rank the same coefficient.
__global__ __launch_bounds__(256) void sumsq_partials_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ partials) {
    __shared__ float red[4];
    float a = 0.f;
    const int64_t n4 = n >> 2;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n4; e += (int64_t)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(g)[e];
        a += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = g[(n4 << 2) + threadIdx.x]; a += v * v; }
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void clip_coef_partials_kernel(const float* __restrict__ partials, int nparts, float max_norm,
                                                                  float* __restrict__ coef, float* __restrict__ norm_out) {
    __shared__ float red[256];
    float a = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) a += partials[i];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float nrm = sqrtf(red[0]);
        if (norm_out) norm_out[0] = nrm;
        coef[0] = max_norm > 0.f ? fminf(1.f, max_norm / (nrm + 1e-6f)) : 1.f;
    }
}

// coef = min(1, max_norm / (sqrt(sumsq) + 1e-6))  (torch.nn.utils.clip_grad_norm_); norm_out = sqrt(sumsq)
__global__ void clip_coef_kernel(const float* __restrict__ sumsq, float max_norm, float* __restrict__ coef, float* __restrict__ norm_out) {
    const float nrm = sqrtf(sumsq[0]);
    if (norm_out) norm_out[0] = nrm;
    coef[0] = max_norm > 0.f ? fminf(1.f, max_norm / (nrm + 1e-6f)) : 1.f;
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                     bf16* __restrict__ shadow, int64_t n, float lr, float b1, float b2, float eps, float wd,
                                                     float bc1, float bc2_sqrt, const float* __restrict__ coef, const int* __restrict__ step_dev,
                                                     const float* __restrict__ lr_scale) {
    const float c = coef ? coef[0] : 1.f;
    if (lr_scale) lr *= lr_scale[0];   // learning-rate schedule factor on the device: a captured graph follows the scheduler
    if (step_dev) {  // step counter lives on the device (captured graphs): bias corrections follow it
        const float t = (float)step_dev[0];
        bc1 = 1.f - powf(b1, t);
        bc2_sqrt = sqrtf(1.f - powf(b2, t));
    }
    const float step = lr / bc1;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const float gr = g[e] * c;
        float pv = p[e] * (1.f - lr * wd);
        const float mv = b1 * m[e] + (1.f - b1) * gr;
        const float vv = b2 * v[e] + (1.f - b2) * gr * gr;
        pv -= step * mv / (sqrtf(vv) / bc2_sqrt + eps);
        p[e] = pv; m[e] = mv; v[e] = vv;
        if (shadow) shadow[e] = (bf16)pv;
    }
}
inline int grid_for(int64_t n) { int64_t b = (n + 255) / 256; return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b)); }
}  // namespace

/* out[0] += sum g^2 */
extern "C" int ralf_sumsq(const float* g, int64_t n, float* out, void* stream) {
    RALF_REQUIRE(g && out && n > 0, "sumsq: bad arguments");
    RALF_REQUIRE(((uintptr_t)g & 15) == 0, "sumsq: buffer must be 16-byte aligned");
    hipLaunchKernelGGL(sumsq_kernel, dim3(grid_for(n / 4 + 1) > 1024 ? 1024 : grid_for(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, g, n, out);
    return ralf::check_launch("sumsq");
}
extern "C" int ralf_sumsq_partials(const float* g, int64_t n, float* partials, void* stream) {
    RALF_REQUIRE(g && partials && n > 0, "sumsq_partials: bad arguments");
    RALF_REQUIRE(((uintptr_t)g & 15) == 0, "sumsq_partials: buffer must be 16-byte aligned");
    hipLaunchKernelGGL(sumsq_partials_kernel, dim3(RALF_SUMSQ_PARTS), dim3(256), 0, (hipStream_t)stream, g, n, partials);
    return ralf::check_launch("sumsq_partials");
}
extern "C" int ralf_clip_coef_partials(const float* partials, float max_norm, float* coef, float* norm_out, void* stream) {
    RALF_REQUIRE(partials && coef, "clip_coef_partials: null pointer");
    hipLaunchKernelGGL(clip_coef_partials_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, RALF_SUMSQ_PARTS, max_norm, coef, norm_out);
    return ralf::check_launch("clip_coef_partials");
}
extern "C" int ralf_clip_coef(const float* sumsq, float max_norm, float* coef, float* norm_out, void* stream) {
    RALF_REQUIRE(sumsq && coef, "clip_coef: null pointer");
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, sumsq, max_norm, coef, norm_out);
    return ralf::check_launch("clip_coef");
}
/* torch.optim.AdamW semantics (decoupled decay, bias correction with step count `step` >= 1, or the
 * device-resident counter step_dev[0] when given); gradients are multiplied by coef[0] (device scalar,
 * may be NULL); shadow = optional bf16 copy of p */
extern "C" int ralf_adamw(float* p, const float* g, float* m, float* v, void* shadow_bf16, int64_t n, float lr, float beta1, float beta2, float eps,
                          float weight_decay, int step, const float* coef, const int* step_dev, const float* lr_scale, void* stream) {
    RALF_REQUIRE(p && g && m && v && n > 0 && (step >= 1 || step_dev), "adamw: bad arguments");
    const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (bf16*)shadow_bf16, n, lr, beta1, beta2, eps, weight_decay, bc1, sqrtf(bc2), coef, step_dev, lr_scale);
    return ralf::check_launch("adamw");
}

extern "C" int ralf_stream_create(void** out_stream) {
    if (!out_stream) { ralf::set_error("stream_create: null output"); return RALF_ERR_INVALID; }
    hipStream_t s = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (e != hipSuccess) { ralf::set_error("stream_create: %s", hipGetErrorString(e)); return RALF_ERR_LAUNCH; }
    *out_stream = (void*)s;
    return RALF_OK;
}

extern "C" int ralf_stream_create_priority(void** out_stream, int high) {
    if (!out_stream) { ralf::set_error("stream_create_priority: null output"); return RALF_ERR_INVALID; }
    int least = 0, greatest = 0;
    hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
    hipStream_t s = nullptr;
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, high ? greatest : least);
    if (e != hipSuccess) { ralf::set_error("stream_create_priority: %s", hipGetErrorString(e)); return RALF_ERR_LAUNCH; }
    *out_stream = (void*)s;
    return RALF_OK;
}

extern "C" int ralf_stream_destroy(void* stream) {
    if (!stream) return RALF_OK;
    hipError_t e = hipStreamDestroy((hipStream_t)stream);
    if (e != hipSuccess) { ralf::set_error("stream_destroy: %s", hipGetErrorString(e)); return RALF_ERR_LAUNCH; }
    return RALF_OK;
}
