"""two gloo ranks on ONE GPU: engine.GradExchange with the HIP pack / unpack kernels against the expected sums, range by range
(python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/debug_two_ranks.py)"""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ralf_amd.engine import GradExchange, complement_ranges  # noqa: E402

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
n = 5_000_000
g = torch.Generator().manual_seed(3)
local = [torch.randn(n, generator=g) for _ in range(world)]
late = [(0, 640 * 64), (1_920_000, 2_048_000)]
early = complement_ranges(late, n)
for wire in ("fp32", "bf16"):
    for mode in ("allreduce", "rs_ag"):
        for asyn in (False, True):
            flat = (local[rank] / world).to(dev)
            ex = GradExchange(flat, world, None, wire, mode=mode)
            tok = ex.start(early, asyn)
            ex.finish(tok)
            ex.run(late)
            torch.cuda.synchronize()
            if wire == "fp32":
                want = sum(x / world for x in local)
            else:
                want = sum((x / world).bfloat16().float() for x in local).bfloat16().float()
            got = flat.cpu()
            bad = (got != want).nonzero().flatten()
            rel = ((got - want).norm() / want.norm()).item()
            if rank == 0:
                print(f"{wire:5s} {mode:9s} async={asyn}: mismatching elements {bad.numel()} of {n}, rel err {rel:.3e}, |got| {got.norm():.4f} |want| {want.norm():.4f}"
                      + (f", first bad index {int(bad[0])}" if bad.numel() else ""), flush=True)
dist.barrier()
dist.destroy_process_group()
