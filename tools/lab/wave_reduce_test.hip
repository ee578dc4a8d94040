// butterfly reductions without the LDS crossbar: DPP (quad_perm, row_half_mirror, row_mirror) inside a 16-lane row,
// v_permlane16_swap / v_permlane32_swap across rows -- checked bit for bit against the __shfl_xor (ds_bpermute_b32) butterflies
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../../ralf_amd/csrc/wave_ops.h"
__device__ __forceinline__ float ref_sum_asc(float v) { for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o); return v; }
__device__ __forceinline__ float ref_xor_sum(float v) { v += __shfl_xor(v, 16); return v + __shfl_xor(v, 32); }
__device__ __forceinline__ float ref_xor_max(float v) { v = fmaxf(v, __shfl_xor(v, 16)); return fmaxf(v, __shfl_xor(v, 32)); }
__device__ __forceinline__ float ref_sum_desc(float v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; }
__device__ __forceinline__ float ref_max(float v) { for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o)); return v; }
constexpr int NT = 6;
__global__ void k(const float* p, float* out) {
  const float v = p[blockIdx.x * 64 + threadIdx.x];
  float* o = out + (blockIdx.x * 64 + threadIdx.x) * 2 * NT;
  o[0] = wave::sum64(v); o[1] = ref_sum_asc(v);
  o[2] = wave::sum_x16_x32(v); o[3] = ref_xor_sum(v);
  o[4] = wave::max_x16_x32(v); o[5] = ref_xor_max(v);
  o[6] = wave::sum64_desc(v); o[7] = ref_sum_desc(v);
  o[8] = wave::max64(v); o[9] = ref_max(v);
  o[10] = wave::xor4(v); o[11] = __shfl_xor(v, 4);
}
int main() {
  const int nb = 1024, n = nb * 64;
  float* h = (float*)malloc(n * 4); float* ho = (float*)malloc(n * 8 * NT);
  srand(1); for (int i = 0; i < n; ++i) h[i] = (float)rand() / RAND_MAX * 200.f - 100.f;
  float *d, *dout; hipMalloc(&d, n * 4); hipMalloc(&dout, n * 8 * NT);
  hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
  k<<<nb, 64>>>(d, dout); hipMemcpy(ho, dout, n * 8 * NT, hipMemcpyDeviceToHost);
  int bad[NT] = {0}, tot = 0;
  for (int i = 0; i < n; ++i) for (int j = 0; j < NT; ++j) if (memcmp(&ho[(i * NT + j) * 2], &ho[(i * NT + j) * 2 + 1], 4)) { ++bad[j]; ++tot; }
  printf("mismatches of %d: sum64 %d  sum_x16_x32 %d  max_x16_x32 %d  sum64_desc %d  max64 %d  xor4 %d\n", n, bad[0], bad[1], bad[2], bad[3], bad[4], bad[5]);
  return tot ? 1 : 0;
}
