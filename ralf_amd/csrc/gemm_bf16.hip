// bf16 instantiations of the GEMM kernel (see gemm_impl.h); split from gemm.hip so the two element types build in parallel
#include "gemm_impl.h"

int ralf_gemm_dispatch_bf16(void* kparams, int nbatch, hipStream_t st) { return dispatch<bf16>(*(KParams*)kparams, nbatch, st); }
int ralf_gemm_reduce_bf16(void* kparams, int nbatch, int blocks, hipStream_t st) {
    hipLaunchKernelGGL((splitk_reduce_kernel<bf16>), dim3(blocks), dim3(256), 0, st, *(KParams*)kparams, nbatch);
    return ralf::check_launch("gemm splitk reduce");
}
