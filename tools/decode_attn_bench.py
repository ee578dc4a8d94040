"""The cross-attention block of a decode step (ralf_decode_attn, self_ = 0) at B = 256 over a 532-row memory, six caches in rotation as in
the decoder (6 x 139 MB: nothing stays in the 256 MB infinity cache between calls).   python tools/decode_attn_bench.py"""
import sys
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import gemm_bench  # noqa: F401  (RALF_LIB=path runs another build of the library)
from ralf_amd import ops
B, M, d, H = 256, 532, 256, 8
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(B, d, device="cuda", generator=g).bfloat16()
W = (torch.randn(3 * d, d, device="cuda", generator=g) * 0.05).bfloat16()
bias = torch.randn(3 * d, device="cuda", generator=g)
lg, lb = torch.ones(d, device="cuda"), torch.zeros(d, device="cuda")
caches = [torch.randn(B, M, 2 * d, device="cuda", generator=g).bfloat16() for _ in range(6)]
packed = [c.view(B, M, 2, H // 2, 64).permute(0, 2, 3, 1, 4).contiguous() for c in caches]   # [B, k | v, head pair, rows, 64]
for c, pc in zip(caches, packed):
    assert torch.equal(ops.decode_attn(x, lg, lb, W, bias, c, M, H, False), ops.decode_attn(x, lg, lb, W, bias, pc, M, H, False, packed_rows=M))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 50
e0.record()
for _ in range(reps):
    for c in caches:
        ops.decode_attn(x, lg, lb, W, bias, c, M, H, False)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / (reps * 6)
byts = B * M * 2 * d * 2
print(f"cross-attention decode block, B={B}, {M} keys: {us:.1f} us per call = {byts / us / 1e6:.2f} TB/s of K/V stream ({byts / us / 1e6 / 8:.3f} of 8 TB/s)")
e0.record()
for _ in range(reps):
    for c in packed:
        ops.decode_attn(x, lg, lb, W, bias, c, M, H, False, packed_rows=M)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / (reps * 6)
print(f"  head-pair-major cache (one contiguous block per workgroup): {us:.1f} us per call = {byts / us / 1e6:.2f} TB/s ({byts / us / 1e6 / 8:.3f} of 8 TB/s)")
