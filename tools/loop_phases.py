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


def tick(name, t0):
    t1 = time.perf_counter()
    acc[name] = acc.get(name, 0.0) + (t1 - t0)
    return t1


for it in range(13):
    if it == 3:
        acc.clear()
        torch.cuda.synchronize()
        tall = time.perf_counter()
    t = time.perf_counter()
    inputs, targets = model.preprocess(batches[it % 3])
    t = tick("preprocess", t)
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
total = (time.perf_counter() - tall) / 10 * 1e3
print(f"loss_lag={lag}: {total:.2f} ms per iteration; " + ", ".join(f"{k} {v / 10 * 1e3:.2f}" for k, v in acc.items()))
