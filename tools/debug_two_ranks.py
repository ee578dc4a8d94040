"""two gloo ranks on ONE GPU: the gradient exchange INSIDE engine.TrainStep checked against the expected sum of the ranks' local gradients
(python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/debug_two_ranks.py [graph])"""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ralf_amd.engine import TrainStep  # noqa: E402
from ralf_amd.synthetic import make_batch, to_device  # noqa: E402

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
use_graph = len(sys.argv) > 1 and sys.argv[1] == "graph"
for wire, mode in (("fp32", "allreduce"), ("bf16", "allreduce"), ("bf16", "rs_ag")):
    torch.manual_seed(100)
    model = bench.build_model(dev, 10, "bfloat16")
    inputs, tgt = model.preprocess(make_batch(2, 10, seed=7 + rank))
    inputs, tgt = to_device(inputs, dev), to_device(tgt, dev)
    inputs["retrieved"] = {k: v for k, v in inputs["retrieved"].items() if k != "image"}
    step = TrainStep(model, use_graph=use_graph, grad_wire=wire, grad_exchange=mode)
    ex = step.exchange
    log = {}
    real_start, real_finish = ex.start, ex.finish

    def start(ranges, async_op, _rs=real_start, _log=log, _ex=ex):
        torch.cuda.synchronize()
        for a, b in ranges:
            _log[(a, b)] = _ex.flat[a:b].clone()
        return _rs(ranges, async_op)
    ex.start = start
    for it in range(2):
        log.clear()
        step(inputs, tgt)
        torch.cuda.synchronize()
        worst, n_el = 0.0, 0
        for (a, b), mine in log.items():
            both = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(both, mine)
            want = both[0] + both[1] if wire == "fp32" else (both[0].bfloat16().float() + both[1].bfloat16().float()).bfloat16().float()
            got = ex.flat[a:b]
            worst = max(worst, ((got - want).norm() / want.norm().clamp_min(1e-30)).item())
            n_el += b - a
        if rank == 0:
            print(f"{wire} {mode} graph={use_graph} step {it}: {len(log)} ranges, {n_el} of {ex.flat.numel()} elements exchanged, worst rel err vs expected {worst:.3e}, "
                  f"grad norm {float(step.opt.grad_norm):.6f}, early {step._early[:2]}... late {len(step._late)} ranges", flush=True)
    del step, model
dist.barrier()
dist.destroy_process_group()
