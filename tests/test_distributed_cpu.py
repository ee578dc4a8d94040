"""CPU, world_size 2, gloo: the data-parallel gradient exchange of the engine (one sum all-reduce over the
flat gradient buffer, backward pre-scaled by 1/world) yields the cross-rank mean on every rank."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ralf_amd.engine import average_gradients

    torch.manual_seed(100 + rank)
    local = torch.randn(10007)                       # this rank's gradient of its own mean loss
    flat = local / world                             # what backward(1/world) leaves in the flat buffer
    average_gradients(flat, world)
    gathered = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    want = torch.stack(gathered).mean(0)
    q.put((rank, torch.allclose(flat, want, atol=1e-6), float((flat - want).abs().max())))
    dist.destroy_process_group()


def test_flat_gradient_average_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res


def test_single_process_is_a_noop():
    from ralf_amd.engine import average_gradients

    g = torch.arange(8.0)
    average_gradients(g, 1)
    assert torch.equal(g, torch.arange(8.0))
