"""the backbone's 3 x 3 / stride-1 convolutions (timm Bottleneck.conv2 of layer1..3, B = 64, 256 x 256 canvases), forward with column statistics and data
gradient with the BatchNorm-backward epilogue:
    RALF_GEMM_PATCH=0 python tools/conv3x3_patch_bench.py [save.pt]    vs    python tools/conv3x3_patch_bench.py [save.pt]
(the patch form, gemm_impl.h GATHER 15).  With a file name the outputs of the first run are saved and the second run compares bit for bit."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ralf_amd import ops  # noqa: E402
from tools.gemm_bench import timeit  # noqa: E402

dt = torch.bfloat16
save = sys.argv[1] if len(sys.argv) > 1 else None
prev = torch.load(save) if save and os.path.exists(save) else None
outs = {}
print("RALF_GEMM_PATCH =", os.environ.get("RALF_GEMM_PATCH", "1"))
for (H, C) in [(64, 64), (32, 128), (16, 256), (8, 512)]:
    B = 64
    M = B * H * H
    g = torch.Generator(device="cuda").manual_seed(H)
    x = torch.randn(M, C, device="cuda", generator=g).to(dt)
    w = (torch.randn(C, 3, 3, C, device="cuda", generator=g) * 0.05).to(dt)
    st = torch.empty((M + 63) // 64, 2, C, device="cuda")
    xa = torch.randn(M, C, device="cuda", generator=g).to(dt)
    mean = torch.randn(C, device="cuda", generator=g)
    bits = torch.randint(0, 256, (M * C // 8,), device="cuda", dtype=torch.uint8, generator=g)
    part = torch.empty((M + 63) // 64, 2, C, device="cuda")
    out = torch.empty(M, C, device="cuda", dtype=dt)
    gf = dict(RH=H, RW=H, SH=H, SW=H, SC=C, KH=3, KW=3, stride=1, pad=1, mode=0)
    gd = dict(gf, mode=1)
    sk = 4 if H == 8 else 1
    kw = dict(splitk=sk) if sk > 1 else {}
    f_fwd = (lambda: ops.gemm(x, w, M, C, 9 * C, conv=gf, gather=1, colstats=st, out=out)) if sk == 1 else (lambda: ops.gemm(x, w, M, C, 9 * C, conv=gf, gather=1, out=out, **kw))
    f_bwd = (lambda: ops.gemm(x, w, M, C, 9 * C, conv=gd, gather=1, bnb=(xa, bits, mean, part), out=out)) if sk == 1 else (lambda: ops.gemm(x, w, M, C, 9 * C, conv=gd, gather=1, out=out, **kw))
    f_fwd(); outs[f"f{H}"] = out.clone(); outs[f"fs{H}"] = st.clone() if sk == 1 else None
    f_bwd(); outs[f"b{H}"] = out.clone(); outs[f"bs{H}"] = part.clone() if sk == 1 else None
    t0, t1 = timeit(f_fwd), timeit(f_bwd)
    fl = 2.0 * M * C * 9 * C
    same = ""
    if prev is not None:
        same = "   bit-identical to the saved run: " + str(all(torch.equal(outs[k], prev[k]) for k in (f"f{H}", f"fs{H}", f"b{H}", f"bs{H}") if outs[k] is not None))
    print(f"3x3 s1 {C:4d} ch at {H:2d}x{H:2d}: forward + statistics {t0 * 1e6:6.1f} us ({fl / t0 / 1e12:5.0f} TFLOP/s)   data gradient + BatchNorm backward {t1 * 1e6:6.1f} us ({fl / t1 / 1e12:5.0f} TFLOP/s){same}")
if save and prev is None:
    torch.save(outs, save)
