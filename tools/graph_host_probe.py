"""host time of ONE replay call of the encoder-decoder-only step (no synchronisation inside the timed region) against its device time:
is the hipGraph launch itself (per-node enqueue on the host) what delays the start of the graph's parallel branches?"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ralf_amd.engine import TrainStep  # noqa: E402
from ralf_amd.synthetic import make_batch, to_device  # noqa: E402

dev = torch.device("cuda", 0)
B, N = 64, 10
model = bench.build_model(dev, N, "bfloat16")
model.encoder = bench._BackboneStandIn(B, 256, 256, dev, model.rt.dtype)
inputs, targets = model.preprocess(make_batch(B, N, seed=1))
inputs, targets = to_device(inputs, dev), to_device(targets, dev)
inputs["retrieved"] = {k: v for k, v in inputs["retrieved"].items() if k != "image"}
step = TrainStep(model, lr=1e-4, weight_decay=1e-4, max_norm=0.1, use_graph=True, overlap_wgrad=True)
step(inputs, targets)
inputs, targets = step.static_batch()
for _ in range(3):
    step(inputs, targets)
torch.cuda.synchronize()
host, total = [], []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step(inputs, targets)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3)
    total.append((t2 - t0) * 1e3)
print("host ms per replay call:", " ".join(f"{h:.2f}" for h in host))
print("call + sync ms         :", " ".join(f"{h:.2f}" for h in total))
