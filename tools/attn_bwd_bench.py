"""One-pass attention backward alone (the image encoder's shape: B = 64, H = 8, 256 x 256 tokens, dh = 32, packed QKV): microseconds per call
(HIP events, back-to-back launches), with and without dropout.  RALF_ATTN_BWD_FUSED16 / RALF_ATTN_BWD_FUSED select the kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ralf_amd import ops  # noqa: E402
from bench import _time_gpu  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(11)
B, H, dh = 64, 8, 32
CASES = [(256, 0.1), (256, 0.0), (128, 0.1)]
if len(sys.argv) > 1:   # one case only (counter passes): S p
    CASES = [(int(sys.argv[1]), float(sys.argv[2]))]
for S, p in CASES:
    seed = torch.tensor([99], dtype=torch.int64, device="cuda")
    qkv = torch.randn(B, S, 3 * H * dh, device="cuda", generator=g).bfloat16()
    offs = (0, H * dh, 2 * H * dh)
    o, lse = ops.attention_fwd(qkv, qkv, qkv, B, H, S, S, dh, *offs, p_drop=p, seed=seed, call_id=5)
    do = torch.randn(B, S, H * dh, device="cuda", generator=g).bfloat16()
    da = torch.zeros_like(qkv)
    fwd = _time_gpu(lambda: ops.attention_fwd(qkv, qkv, qkv, B, H, S, S, dh, *offs, p_drop=p, seed=seed, call_id=5), 50, 5)
    bwd = _time_gpu(lambda: ops.attention_bwd(do, qkv, qkv, qkv, o, lse, da, da, da, B, H, S, S, dh, *offs, *offs, p_drop=p, seed=seed, call_id=5), 50, 5)
    scores = B * H * S * S
    print(f"S={S} p={p}: forward {fwd * 1e6:6.1f} us, backward {bwd * 1e6:6.1f} us  ({scores / bwd / 1e12:.2f} T scores/s backward)")
