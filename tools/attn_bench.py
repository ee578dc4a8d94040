"""Micro-benchmark of the MFMA attention kernels on the RALF shapes (bf16), with and without probability dropout."""
import sys
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tools")
from ralf_amd import ops  # noqa: E402
from gemm_bench import timeit  # noqa: E402

dt = torch.bfloat16
seed = torch.tensor([1234], dtype=torch.int64, device="cuda")
for (name, B, H, Sq, Sk, dh, causal) in [("enc self", 64, 8, 256, 256, 32, False), ("dec self", 64, 8, 50, 50, 32, True), ("dec cross", 64, 8, 50, 532, 32, False),
                                          ("fuse", 64, 8, 256, 16, 64, False), ("layout enc", 1024, 4, 11, 11, 64, False)]:
    d = H * dh
    qkv = torch.randn(B, max(Sq, Sk), 3 * d, device="cuda").to(dt)
    q = qkv[:, :Sq].contiguous() if Sq != Sk else qkv
    for p in (0.0, 0.1):
        kw = dict(q_off=0, k_off=d, v_off=2 * d, causal=causal, p_drop=p, seed=seed, call_id=3)
        o, lse = ops.attention_fwd(q, qkv, qkv, B, H, Sq, Sk, dh, **kw)
        tf = timeit(lambda: ops.attention_fwd(q, qkv, qkv, B, H, Sq, Sk, dh, **kw))
        do = torch.randn_like(o)
        dq, dkv = torch.empty_like(q), torch.empty_like(qkv)
        bw = dict(q_off=0, k_off=d, v_off=2 * d, dq_off=0, dk_off=d, dv_off=2 * d, causal=causal, p_drop=p, seed=seed, call_id=3)
        tb = timeit(lambda: ops.attention_bwd(do, q, qkv, qkv, o, lse, dq, dkv, dkv, B, H, Sq, Sk, dh, **bw))
        fl = 4.0 * B * H * Sq * Sk * dh
        print(f"{name:10s} B={B:4d} H={H} Sq={Sq:3d} Sk={Sk:3d} dh={dh} p={p}: fwd {tf*1e6:7.1f} us ({fl/tf/1e12:6.1f} TF/s)   bwd {tb*1e6:7.1f} us ({2.5*fl/tb/1e12:6.1f} TF/s)")
