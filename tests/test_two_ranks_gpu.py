"""GPU: TWO data-parallel ranks on ONE GPU (gloo carries the device tensors; RCCL refuses two ranks on one device): replica sync,
rank-dependent dropout seeds, the staged three-graph step with the ranged gradient exchange -- after N steps both ranks hold the
same weights, and they differ from what either rank would have reached alone."""
import os
import sys

import pytest
from conftest import free_port
import torch

pytestmark = pytest.mark.gpu


def _rank(rank, world, port, out_dir, wire="fp32", mode="allreduce"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    from ralf_amd.engine import TrainStep
    from ralf_amd.synthetic import make_batch, to_device

    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    try:
        t = torch.ones(4, device=dev) * (rank + 1)
        dist.all_reduce(t)
        ok = bool((t == 3).all())
    except Exception as e:   # a torch build whose gloo cannot carry device tensors
        torch.save({"skip": repr(e)}, os.path.join(out_dir, f"r{rank}.pt"))
        return
    assert ok
    if wire == "bf16":
        try:   # the (opt-in) bf16 wire reduces bf16 DEVICE tensors: probe the transport itself at the size of a real range first
            g = torch.Generator().manual_seed(5)
            both = [torch.randn(1 << 20, generator=g).bfloat16() for _ in range(world)]
            want = (both[0].float() + both[1].float()).bfloat16()
            t = both[rank].to(dev)
            if mode == "rs_ag":
                n = t.numel() // world
                dist.reduce_scatter_tensor(t[rank * n:(rank + 1) * n], t)
                dist.all_gather_into_tensor(t, t[rank * n:(rank + 1) * n])
            else:
                dist.all_reduce(t)
            assert torch.equal(t.cpu(), want), "gloo does not sum bf16 device tensors of this size correctly"
        except Exception as e:
            torch.save({"skip": repr(e)}, os.path.join(out_dir, f"r{rank}.pt"))
            return
    torch.manual_seed(100 + rank)                                   # DIFFERENT initial weights per rank: the step must sync them
    model = bench.build_model(dev, 10, "bfloat16")
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.01 * (rank + 1))
    inputs, tgt = model.preprocess(make_batch(2, 10, seed=7 + rank))   # different data per rank
    inputs, tgt = to_device(inputs, dev), to_device(tgt, dev)
    inputs["retrieved"] = {k: v for k, v in inputs["retrieved"].items() if k != "image"}
    step = TrainStep(model, use_graph=True, grad_wire=wire, grad_exchange=mode)
    assert step.world == 2 and step.staged and step.exchange.active and step.exchange.wire == wire and step.exchange.mode == mode
    seed0 = int(model.rt._seed_host)
    losses = [float(step(inputs, tgt))]
    gnorm1 = float(step.opt.grad_norm)      # clip norm of the FIRST step: same weights in every run, so the runs differ by the wire alone
    losses += [float(step(inputs, tgt)) for _ in range(2)]
    torch.cuda.synchronize()
    P = step.opt.P.detach().clone()
    gathered = [torch.empty_like(P) for _ in range(world)]
    dist.all_gather(gathered, P)
    same = bool(torch.equal(gathered[0], gathered[1]))
    torch.save({"losses": losses, "same": same, "seed": seed0, "pnorm": float(P.double().norm()), "gnorm": float(step.opt.grad_norm), "gnorm1": gnorm1},
               os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


_FP32_RUN = {}


@pytest.mark.parametrize("wire,mode", [("fp32", "allreduce"), ("fp32", "rs_ag"), ("bf16", "allreduce"), ("bf16", "rs_ag")])
def test_two_ranks_on_one_gpu_keep_identical_weights(tmp_path, wire, mode):
    """every form of the gradient exchange (all_reduce / reduce-scatter + all-gather, fp32 wire / the opt-in bf16 wire with its HIP
    pack and unpack kernels over the staged ranges): replicas stay bit-identical; reduce-scatter + all-gather sums in the same
    pairs as all_reduce on two ranks (same weights and clip norm); the bf16 wire moves the clip norm by rounding only"""
    import torch.multiprocessing as mp

    port = free_port()
    mp.spawn(_rank, args=(2, port, str(tmp_path), wire, mode), nprocs=2, join=True)
    r = [torch.load(os.path.join(str(tmp_path), f"r{i}.pt")) for i in range(2)]
    if "skip" in r[0]:
        pytest.skip("gloo cannot reduce these device tensors in this build: " + r[0]["skip"])
    assert r[0]["same"] and r[1]["same"], "ranks diverged"
    assert r[0]["seed"] != r[1]["seed"], "ranks drew the same dropout masks"
    assert all(x == x for x in r[0]["losses"] + r[1]["losses"])
    assert abs(r[0]["pnorm"] - r[1]["pnorm"]) == 0.0
    assert r[0]["gnorm"] == r[1]["gnorm"] and r[0]["gnorm"] > 0      # the clip coefficient is the same bit pattern on both ranks
    if (wire, mode) == ("fp32", "allreduce"):
        _FP32_RUN.update(r[0])
    elif _FP32_RUN:
        base = _FP32_RUN
        if wire == "fp32":
            # two runs of the same step agree to the order of the fp32 atomics in the column sums (bias / LayerNorm / embedding gradients:
            # one ulp, tools/determinism_probe.py -- every matrix and filter gradient is bit-identical): the FIRST step's clip norm to 1e-6;
            # by the third step AdamW's normalised update has amplified those ulps on this 2-sample batch (measured 2.4e-5)
            assert abs(r[0]["gnorm1"] - base["gnorm1"]) < 1e-6 * base["gnorm1"], (r[0]["gnorm1"], base["gnorm1"])
            assert abs(r[0]["gnorm"] - base["gnorm"]) < 2e-4 * base["gnorm"] and abs(r[0]["pnorm"] - base["pnorm"]) < 1e-6 * base["pnorm"]
        else:
            # bf16 on the wire: every gradient element rounded to 8 bits once per rank -> the FIRST step's norm moves by << 2^-8 relative
            # (measured 4e-5).  Later steps are not comparable: AdamW's normalised update turns the rounding of a near-zero gradient
            # into a full-size step, so the two trajectories separate (step 2: 0.9 %, step 3: tens of percent on this tiny batch) --
            # which is why the bf16 wire is opt-in
            assert abs(r[0]["gnorm1"] - base["gnorm1"]) < 2.0 ** -8 * base["gnorm1"], (r[0]["gnorm1"], base["gnorm1"])


def _rank_shapes(rank, world, port, out_dir):
    """rank 0 meets a NEW batch shape (the ragged last batch) at a step where rank 1 replays a cached one, and vice versa"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import datetime

    import torch.distributed as dist

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    from ralf_amd.engine import TrainStep
    from ralf_amd.synthetic import make_batch, to_device

    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=240))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    try:
        t = torch.ones(4, device=dev) * (rank + 1)
        dist.all_reduce(t)
    except Exception as e:
        torch.save({"skip": repr(e)}, os.path.join(out_dir, f"s{rank}.pt"))
        return
    torch.manual_seed(3)
    model = bench.build_model(dev, 10, "bfloat16")

    def batch(n, seed):
        i, t = model.preprocess(make_batch(n, 10, seed=seed))
        i, t = to_device(i, dev), to_device(t, dev)
        i["retrieved"] = {k: v for k, v in i["retrieved"].items() if k != "image"}
        return i, t
    full, ragged = batch(3, 11 + rank), batch(2, 21 + rank)
    # step:      0     1       2       3     4
    seq = [[full, ragged, full, full, ragged], [full, full, full, ragged, ragged]][rank]
    step = TrainStep(model, use_graph=True)
    rounds0 = step.exchange.rounds
    losses = [float(step(*b)) for b in seq]
    torch.cuda.synchronize()
    P = step.opt.P.detach().clone()
    gathered = [torch.empty_like(P) for _ in range(world)]
    dist.all_gather(gathered, P)
    torch.save({"losses": losses, "same": bool(torch.equal(gathered[0], gathered[1])), "captures": step.captures,
                "rounds": step.exchange.rounds - rounds0}, os.path.join(out_dir, f"s{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_ranks_with_different_batch_shape_sequences_stay_in_step(tmp_path):
    """ADVICE r5 (engine.py capture): a capture triggered by ONE rank's batch shape must not issue an extra round of the gradient exchange --
    the other rank replays a cached shape and issues one.  Both ranks run five steps with two shapes met at different steps: the same number
    of exchange rounds on both (the first capture's warm-up + one per step), identical weights at the end, no hang."""
    import torch.multiprocessing as mp

    mp.spawn(_rank_shapes, args=(2, free_port(), str(tmp_path)), nprocs=2, join=True)
    r = [torch.load(os.path.join(str(tmp_path), f"s{i}.pt")) for i in range(2)]
    if "skip" in r[0]:
        pytest.skip("gloo cannot reduce device tensors in this build: " + r[0]["skip"])
    assert r[0]["captures"] == 2 and r[1]["captures"] == 2
    assert r[0]["rounds"] == r[1]["rounds"], (r[0]["rounds"], r[1]["rounds"])
    assert r[0]["same"] and r[1]["same"], "ranks diverged"
    assert all(x == x for x in r[0]["losses"] + r[1]["losses"])


def test_bench_two_ranks_on_one_gpu_prints_one_json_line(tmp_path):
    """the N > 1 flow of bench.py itself (launcher environment, replica sync, barriers, max-over-ranks timing, rank 0's line, the
    data_parallel block) with two ranks on this one GPU (RALF_BENCH_ONE_DEVICE: gloo instead of RCCL, both ranks on device 0)"""
    import json
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RALF_BENCH_ONE_DEVICE="1", OMP_NUM_THREADS="4")
    port = free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "8"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]                 # stdout carries exactly the one JSON line of rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 3 and d["warmup"] == 1
    dp = d["config"]["data_parallel"]
    assert dp["rccl_ranks"] == 2 and dp["staged_backward"] and dp["bytes_on_wire_per_step"] > 0 and dp["allreduce_ms_standalone"] > 0
    assert dp["replicas_bit_identical"]
    assert d["config"]["parallelism"] == "dp2" and d["value"] > 0
    # whole-job throughput: both ranks' tokens over the slowest rank's time
    assert abs(d["value"] - 2 * d["config"]["tokens_per_step_per_gpu"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert "knn" not in d and "cpu_baseline" not in d        # single-GPU legs only run at N = 1


def test_bench_two_ranks_on_real_rccl_when_two_gpus_are_visible():
    """BASELINE config 3's transport itself: `bench.py --gpus 2` exactly as the driver launches it (torch.distributed.run, one rank per GPU,
    backend nccl = RCCL over xGMI) -- runs only where a second GPU is visible, so that the first multi-GPU node does not meet untested code
    (image2layout/train/helpers/distrubuted.py:10-31, train/train.py:52-61).  The gradient exchange must run on an RCCL group of two ranks and
    leave both replicas with the same bits."""
    import json
    import subprocess

    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible: the two-rank RCCL run needs two")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k != "RALF_BENCH_ONE_DEVICE"}
    env.update(OMP_NUM_THREADS="4", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    dp = d["config"]["data_parallel"]
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["scaling"] == "weak"
    assert dp["rccl_ranks"] == 2 and dp["backend"] == "nccl" and dp["replicas_bit_identical"] and dp["bytes_on_wire_per_step"] > 0


def _rank_knn(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from ralf_amd.retrieval import FlatIPIndex, knn_topk_ip, query_block, search_index_sharded, search_query_sharded

    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    g = torch.Generator().manual_seed(21)
    X = torch.randn(5003, 64, generator=g)
    X[4000] = X[17]; X[2501] = X[17]                      # exact ties across the two shards
    Q = torch.cat([X[[17, 4000]], torch.randn(21, 64, generator=g)])
    want_s, want_i = knn_topk_ip(X.to(dev), Q.to(dev), 16)
    ok = {}
    s, i = search_query_sharded(FlatIPIndex(X, device="cuda").search, Q.to(dev), 16)      # replicas: rank r scans its block of the queries
    ok["query_sharded"] = bool(torch.equal(i, want_i) and torch.equal(s, want_s))
    blk = query_block(X.shape[0], rank, world)                                            # index shards: rank r owns a block of the rows
    s, i = search_index_sharded(FlatIPIndex(X[blk], device="cuda").search, blk.start, Q.to(dev), 16)
    ok["index_sharded"] = bool(torch.equal(i, want_i) and torch.equal(s, want_s))
    torch.save(ok, os.path.join(out_dir, f"k{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_knn_two_ranks_on_the_hip_scan(tmp_path):
    """SURVEY 8e with TWO ranks on the real scan: query-sharded replicas and index shards (local top-k with global indices
    all-gathered, merged by score descending / index ascending) reproduce the single-GPU result bit for bit, ties across shards
    included"""
    import torch.multiprocessing as mp

    port = free_port()
    mp.spawn(_rank_knn, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        ok = torch.load(os.path.join(str(tmp_path), f"k{r}.pt"))
        assert ok == {"query_sharded": True, "index_sharded": True}, (r, ok)


def test_dp_selftest_is_as_fast_with_the_users_queue_setting_as_with_the_default():
    """the staged data-parallel step (three graphs around RCCL) depended on how its graph branches alias onto the hardware queues: 8 queues
    ran it at 35.6 ms against 16.2 with ROCclr's default of 4 (HISTORY.md section 5).  The package sets GPU_MAX_HW_QUEUES=4 when it is unset and
    respects an exported value; bench.py -- a launcher of its own -- forces 4 for itself (RALF_FORCE_HW_QUEUES): `bench.py --dp-selftest`
    (1-rank RCCL group, staged backward, overlapped exchange) started with GPU_MAX_HW_QUEUES=8 in the environment must run like the one
    started without it"""
    import json
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ms = {}
    for name, extra in (("default", {}), ("user_sets_8", {"GPU_MAX_HW_QUEUES": "8"})):
        env = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "RALF_FORCE_HW_QUEUES")}
        env.update(extra, MASTER_PORT=str(free_port()), PYTHONWARNINGS="ignore")
        cmd = [sys.executable, os.path.join(root, "bench.py"), "--dp-selftest", "--steps", "10", "--warmup", "3", "--skip-cpu", "--skip-knn", "--skip-split",
               "--skip-decode", "--skip-variants"]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        d = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][-1])
        assert d["config"]["data_parallel"]["staged_backward"] and d["config"]["data_parallel"]["rccl_ranks"] == 1
        ms[name] = d["ms_per_step"]
    assert ms["user_sets_8"] < 1.15 * ms["default"] and ms["default"] < 20.0, ms
