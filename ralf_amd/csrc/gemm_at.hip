// bf16 GEMM kernels with an A-operand transform (RalfGemmDesc.at_*, gemm_impl.h GATHER 7 / 9 / 8): the BatchNorm apply (+ residual + ReLU) of a
// 1x1 convolution's input and the BatchNorm backward apply of its output gradient run inside the operand loader, with write-through of the
// transformed operand.  Replaces ralf_bn_apply / ralf_bn_bwd_apply in front of the 1x1 convolutions of the ResNet bottlenecks
// (image2layout/train/models/common/image.py:39-48: timm Bottleneck conv -> bn -> relu chains).  Own translation unit: builds in parallel.
#include "gemm_impl.h"

int ralf_gemm_dispatch_at(void* kparams, int nbatch, hipStream_t st) {
    KParams& P = *(KParams*)kparams;
    const RalfGemmDesc& d = P.d;
    if (d.at_mode == 1) {   // forward products: weights [N][K]
        if (!d.b_kcontig) { ralf::set_error("gemm: at_mode 1 needs B k-contiguous ([N][K] weights)"); return RALF_ERR_INVALID; }
        return d.at_a2 ? launch_at<bf16, true, 7>(P, nbatch, st) : launch_at<bf16, true, 9>(P, nbatch, st);
    }
    if (d.b_kcontig) { ralf::set_error("gemm: at_mode 2 needs B row-contiguous ([K][N]: the data gradient's weights)"); return RALF_ERR_INVALID; }
    return launch_at<bf16, false, 8>(P, nbatch, st);
}
