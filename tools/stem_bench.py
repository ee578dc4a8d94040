"""the stem convolution at B = 64, 256 x 256: direct form (ralf_stem7x7_fwd) against the implicit-GEMM form"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import _time_gpu  # noqa: E402
from ralf_amd import ops  # noqa: E402

B, H = 64, 256
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.zeros(B, H, H, 8, device="cuda"); x[..., :4] = torch.rand(B, H, H, 4, device="cuda", generator=g); x = x.to(torch.bfloat16)
w = torch.zeros(64, 7, 7, 8, device="cuda"); w[..., :4] = torch.randn(64, 7, 7, 4, device="cuda", generator=g) * 0.1; w = w.to(torch.bfloat16)
M = B * 128 * 128
geom = dict(RH=128, RW=128, SH=H, SW=H, SC=8, KH=7, KW=7, stride=2, pad=3, mode=0)
cst = ops.colstats_buffer(M, 64, x.device)
to = _time_gpu(lambda: ops.gemm(x, w, M, 64, 392, conv=geom, gather=1, colstats=cst), 20, 3)
tn = _time_gpu(lambda: ops.stem7x7_fwd(x, w), 20, 3)
print(f"stem forward: implicit-GEMM {to * 1e6:.1f} us, direct {tn * 1e6:.1f} us ({(67.1 + 134.2) / tn / 1e6:.2f} TB/s of 201 MB)")

dy = torch.randn(B, 128, 128, 64, device="cuda", generator=g).to(torch.bfloat16)
from ralf_amd.functional import _splitk_for  # noqa: E402
out = torch.empty(64, 4, 7, 7, device="cuda")


def old_w():
    gg = ops.gemm(dy.view(M, 64), x, 64, 392, M, a_kcontig=False, b_kcontig=False, conv=geom, gather=2, out_dtype=torch.float32, splitk=_splitk_for(64, 392, M))
    return ops.permute4(gg, (64, 4, 7, 7), (392, 1, 56, 8), 7, torch.float32, out=out)


a = old_w().clone()
b = ops.stem7x7_wgrad(x, dy).clone()
print("wgrad max rel diff", ((a - b).abs().max() / a.abs().max()).item())
to, tn = _time_gpu(old_w, 20, 3), _time_gpu(lambda: ops.stem7x7_wgrad(x, dy, out=out), 20, 3)
print(f"stem weight gradient: implicit-GEMM {to * 1e6:.1f} us, direct {tn * 1e6:.1f} us")
