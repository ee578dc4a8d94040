"""where the host time of the reference's unchanged loop body goes (train/train.py:432-454 with engine.GraphedAdamW): per-phase wall clock"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ralf_amd.engine import GraphedAdamW  # noqa: E402
from ralf_amd.synthetic import make_batch  # noqa: E402

dev = torch.device("cuda", 0)
B, N = 64, 10
lag = int(sys.argv[1]) if len(sys.argv) > 1 else 1
model = bench.build_model(dev, N, "bfloat16")
opt = GraphedAdamW(params=model.optim_groups(base_lr=1e-4, weight_decay=0.01, custom_lr={"encoder.extractor.body": 1e-5}), max_norm=0.1, loss_lag=lag)
batches = [make_batch(B, N, seed=21 + i) for i in range(3)]
acc = {}
# finer: where train_loss spends its host time
import ralf_amd.engine as E  # noqa: E402
_orig_copy, _orig_replay = E._copy_tree, torch.cuda.CUDAGraph.replay


def _timed(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        acc["  " + name] = acc.get("  " + name, 0.0) + time.perf_counter() - t0
        return r
    return w


_depth = [0]


def _copy(dst, src, stager=None):
    if _depth[0]:
        return _orig_copy(dst, src, stager)
    _depth[0] = 1
    t0 = time.perf_counter()
    try:
        return _orig_copy(dst, src, stager)
    finally:
        _depth[0] = 0
        acc["  copy_tree"] = acc.get("  copy_tree", 0.0) + time.perf_counter() - t0


_orig_step = E.TrainStep._step
_evs = []


def _step(self, inputs, targets):
    e0, e1, e2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = _orig_step(self, inputs, targets)
    e1.record()
    _evs.append((e0, e1))
    return r


E.TrainStep._step = _step
E._copy_tree = _copy
torch.cuda.CUDAGraph.replay = _timed("graph.replay", _orig_replay)
E._HostStager.begin = _timed("stager.begin", E._HostStager.begin)


def tick(name, t0):
    t1 = time.perf_counter()
    acc[name] = acc.get(name, 0.0) + (t1 - t0)
    return t1


import gc  # noqa: E402
if os.environ.get("LOOP_GC") == "freeze":
    gc.collect(); gc.freeze()
elif os.environ.get("LOOP_GC") == "off":
    gc.disable()
for it in range(13):
    if it == 3:
        acc.clear()
        torch.cuda.synchronize()
        tall = time.perf_counter()
    t = time.perf_counter()
    inputs, targets = model.preprocess(batches[it % 3])
    t = tick("preprocess", t)
    if os.environ.get("LOOP_SYNC_UPLOAD"):
        torch.cuda.synchronize()
        print(f"it {it}: upload done {1e3 * (time.perf_counter() - t):.2f} ms after preprocess returned; image ptr {inputs['image'].data_ptr():x}")
        t = time.perf_counter()
    inputs = {k: v.to(dev) if torch.is_tensor(v) else v for (k, v) in inputs.items()}
    targets = {k: v.to(dev) if torch.is_tensor(v) else v for (k, v) in targets.items()}
    t = tick("to(device)", t)
    model.zero_grad()
    t = tick("zero_grad", t)
    out, losses = model.train_loss(inputs, targets)
    t = tick("train_loss", t)
    loss = sum(losses.values())
    loss.backward()
    t = tick("backward", t)
    torch.nn.utils.clip_grad_norm_(model.parameters(), 0.1)
    t = tick("clip", t)
    opt.step()
    t = tick("opt.step", t)
    v = loss.cpu().item()
    t = tick("loss.item", t)
torch.cuda.synchronize()
gpu = [a.elapsed_time(b) for a, b in _evs[-10:]]
gaps = [_evs[i][1].elapsed_time(_evs[i + 1][0]) for i in range(len(_evs) - 10, len(_evs) - 1)]
print("GPU time of the step (events on its stream), last 10:", " ".join(f"{g:.1f}" for g in gpu))
print("GPU idle/other time between steps:", " ".join(f"{g:.1f}" for g in gaps))
total = (time.perf_counter() - tall) / 10 * 1e3
print(f"loss_lag={lag}: {total:.2f} ms per iteration; " + ", ".join(f"{k} {v / 10 * 1e3:.2f}" for k, v in acc.items()))
