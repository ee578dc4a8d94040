"""CPU, world_size 2, gloo: the data-parallel gradient exchange of the engine (one sum all-reduce over the
flat gradient buffer, backward pre-scaled by 1/world) yields the cross-rank mean on every rank."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import free_port


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
    port = free_port()
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


def test_flat_ranges_and_complement():
    """the staged exchange addresses the flat gradient buffer by ranges: parameter views -> merged ranges (alignment padding
    bridged), complement = everything else; together they tile the buffer exactly once."""
    from ralf_amd.engine import complement_ranges, flat_ranges

    flat = torch.zeros(64 * 10)
    sizes = [(0, 50), (64, 64), (128, 7), (320, 100), (448, 64), (512, 30)]   # 64-aligned starts, as FlatAdamW lays them out
    params = []
    for a, n in sizes:
        p = torch.nn.Parameter(torch.zeros(n))
        p.grad = flat[a:a + n]
        params.append(p)
    r = flat_ranges(params[::-1], flat, align=64)
    assert r == [(0, 135), (320, 542)]
    c = complement_ranges(r, flat.numel())
    assert c == [(135, 320), (542, 640)]
    cover = torch.zeros(flat.numel())
    for a, b in r + c:
        cover[a:b] += 1
    assert bool((cover == 1).all())
    assert complement_ranges([], 10) == [(0, 10)] and complement_ranges([(0, 10)], 10) == []


def _worker_ranges(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ralf_amd.engine import complement_ranges

    torch.manual_seed(7 + rank)
    flat = torch.randn(4096)
    whole = flat.clone()
    dist.all_reduce(whole)
    late = [(0, 700), (1500, 1564)]
    early = complement_ranges(late, flat.numel())
    works = [dist.all_reduce(flat[a:b], async_op=True) for a, b in early]      # TrainStep._exchange_around
    for w in works:
        w.wait()
    for a, b in late:
        dist.all_reduce(flat[a:b])
    q.put((rank, torch.equal(flat, whole), 0.0))
    dist.destroy_process_group()


def test_ranged_exchange_equals_whole_buffer_exchange_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker_ranges, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res


def _worker_exchange(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ralf_amd.engine import GradExchange, complement_ranges

    def cp(src, dst):      # stands in for the HIP cast kernel (ralf_copy2d) on this CPU-only box
        dst.copy_(src)

    torch.manual_seed(11 + rank)
    local = torch.randn(5000)
    gathered = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    ok = True
    # (the second set of ranges has odd lengths: rs_ag falls back to all_reduce for those)
    for wire, mode, late in (("fp32", "allreduce", [(0, 640), (1920, 2048)]), ("bf16", "allreduce", [(0, 640), (1920, 2048)]),
                             ("fp32", "rs_ag", [(0, 640), (1920, 2048)]), ("bf16", "rs_ag", [(0, 640), (1920, 2048)]),
                             ("fp32", "rs_ag", [(0, 641), (1920, 2048)])):
        early = complement_ranges(late, local.numel())
        flat = local / world                                   # the backward was seeded with 1/world
        ex = GradExchange(flat, world, None, wire, pack=cp, unpack=cp, mode=mode)
        token = ex.start(early, True)                          # TrainStep._exchange_around: early ranges overlap stage 2 ...
        ex.finish(token)
        ex.run(late)                                           # ... the rest afterwards
        if wire == "fp32":
            want = torch.stack(gathered).mean(0)
            ok &= torch.allclose(flat, want, atol=1e-6)
        else:   # each rank's contribution is rounded to bf16, the sum is carried in bf16
            want = sum((g / world).bfloat16() for g in gathered).float()
            ok &= torch.allclose(flat, want, atol=0, rtol=2 ** -7)
            ok &= (flat - torch.stack(gathered).mean(0)).abs().max().item() < 2e-2
            ok &= ex.bytes_on_wire(early + late) == 2 * local.numel()
            # replica identity: both ranks hold the same bits; the clip norm moves by rounding only
            both = [torch.empty_like(flat) for _ in range(world)]
            dist.all_gather(both, flat)
            ok &= torch.equal(both[0], both[1])
            n_bf16, n_fp32 = flat.double().norm().item(), torch.stack(gathered).mean(0).double().norm().item()
            ok &= abs(n_bf16 - n_fp32) < 2.0 ** -8 * n_fp32
    q.put((rank, bool(ok), 0.0))
    dist.destroy_process_group()


def test_grad_exchange_fp32_and_bf16_wire_world2():
    """engine.GradExchange over two gloo ranks: ranged start/finish + run tile the buffer, as all_reduce and as reduce-scatter +
    all-gather; the (opt-in) bf16 wire halves the bytes"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker_exchange, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res


def _numpy_scan(X):
    """exact flat-IP top-k with the scan's tie order (score desc, index asc); pads with (-inf, -1) when k > rows"""
    import numpy as np

    def search(Q, k):
        S = (np.asarray(Q, np.float64) @ X.T.astype(np.float64)).astype(np.float32)
        order = np.lexsort((np.broadcast_to(np.arange(S.shape[1]), S.shape), -S), axis=1)[:, :k]
        s, i = np.take_along_axis(S, order, 1), order.astype(np.int64)
        if s.shape[1] < k:
            pad = k - s.shape[1]
            s = np.concatenate([s, np.full((s.shape[0], pad), -np.inf, np.float32)], 1)
            i = np.concatenate([i, np.full((i.shape[0], pad), -1, np.int64)], 1)
        return torch.from_numpy(s), torch.from_numpy(i)
    return search


def _worker_knn(rank, world, port, q):
    import numpy as np

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ralf_amd.retrieval.sharded import query_block, search_index_sharded, search_query_sharded

    rng = np.random.default_rng(5)
    X = rng.standard_normal((301, 16)).astype(np.float32)
    X[40] = X[7]; X[200] = X[7]                                   # exact ties across shards
    Q = torch.from_numpy(np.concatenate([X[[7, 40]], rng.standard_normal((9, 16)).astype(np.float32)]))
    want_s, want_i = _numpy_scan(X)(Q, 17)
    ok = True
    s, i = search_query_sharded(_numpy_scan(X), Q, 17)            # replicas: rank r scans its block of the queries
    ok &= torch.equal(i, want_i) and torch.equal(s, want_s)
    blk = query_block(X.shape[0], rank, world)                    # index shards: rank r owns a block of the rows
    s, i = search_index_sharded(_numpy_scan(X[blk]), blk.start, Q, 17)
    ok &= torch.equal(i, want_i) and torch.equal(s, want_s)
    tiny = query_block(5, rank, world)                            # shards smaller than k: padded lists still merge
    s, i = search_index_sharded(_numpy_scan(X[:5][tiny]), tiny.start, Q, 4)
    ws, wi = _numpy_scan(X[:5])(Q, 4)
    ok &= torch.equal(i, wi) and torch.equal(s, ws)
    q.put((rank, bool(ok), 0.0))
    dist.destroy_process_group()


def test_sharded_knn_world2():
    """query-sharded replicas and index shards (all_gather of the local top-k + merge by score desc, index asc)"""
    from ralf_amd.retrieval.sharded import query_block

    assert [query_block(11, r, 4) for r in range(4)] == [slice(0, 3), slice(3, 6), slice(6, 9), slice(9, 11)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker_knn, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res


def test_staged_ranges_are_rounded_so_that_every_range_keeps_reduce_scatter_form():
    """the staged exchange's ranges end where a parameter ends; rounded outwards to world * 64 elements (and the complement
    taken afterwards) every range is divisible by the world size, late parameters stay inside late ranges, nothing is exchanged twice"""
    from ralf_amd.engine import _round_ranges, complement_ranges

    total = 1024 * 9
    late = [(64, 1000), (1100, 1160), (5000, 5003)]
    for world in (2, 4, 8):
        q = 64 * world
        r = _round_ranges(late, q, total)
        early = complement_ranges(r, total)
        assert all(a % q == 0 and b % q == 0 for a, b in r + early)
        assert all(any(ra <= a and b <= rb for ra, rb in r) for a, b in late)
        cover = sorted(r + early)
        assert cover[0][0] == 0 and cover[-1][1] == total and all(x[1] == y[0] for x, y in zip(cover, cover[1:]))
