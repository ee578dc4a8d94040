"""two ranks on one GPU over gloo: where do the replicas diverge?  python tools/debug_two_ranks.py"""
import os
import sys

import torch


def _rank(rank, world, port, use_graph, staged):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    sys.path.insert(0, root)
    import bench
    from ralf_amd.engine import TrainStep
    from ralf_amd.synthetic import make_batch, to_device

    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    torch.manual_seed(100 + rank)
    model = bench.build_model(dev, 10, "bfloat16")
    inputs, tgt = model.preprocess(make_batch(2, 10, seed=7 + rank))
    inputs, tgt = to_device(inputs, dev), to_device(tgt, dev)
    inputs["retrieved"] = {k: v for k, v in inputs["retrieved"].items() if k != "image"}
    step = TrainStep(model, use_graph=use_graph, grad_wire="fp32", overlap_allreduce=staged)

    def diff(t, what):
        g = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(g, t.contiguous())
        d = (g[0].double() - g[1].double()).abs()
        if rank == 0:
            nz = int((d > 0).sum())
            print(f"  {what}: max |r0 - r1| = {d.max().item():.3e}, differing elements {nz} of {d.numel()}" + (f", first at {int((d > 0).nonzero()[0])}" if nz else ""), flush=True)

    diff(step.opt.P, "P after sync")
    for n in range(3):
        loss = float(step(inputs, tgt))
        torch.cuda.synchronize()
        if rank == 0:
            print(f" step {n}: loss r0 {loss:.4f}", flush=True)
        diff(step.opt.G, "G after exchange")
        diff(step.opt.coef, "clip coef")
        diff(step.opt.M, "M")
        diff(step.opt.P, "P")
        if step.opt.P16 is not None:
            diff(step.opt.P16.float(), "P16")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp

    for use_graph, staged in ((False, False), (False, True), (True, True)):
        print(f"== use_graph={use_graph} staged={staged}", flush=True)
        mp.spawn(_rank, args=(2, 29700 + os.getpid() % 200 + (2 * use_graph + staged), use_graph, staged), nprocs=2, join=True)
