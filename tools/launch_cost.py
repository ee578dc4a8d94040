"""CPU time of one graph-replayed train step (the host-side hipGraphLaunch calls, no synchronisation) vs its device time"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ralf_amd.engine import TrainStep  # noqa: E402
from ralf_amd.synthetic import make_batch, to_device  # noqa: E402

dev = torch.device("cuda", 0)
model = bench.build_model(dev, 10, "bfloat16")
inputs, targets = model.preprocess(make_batch(64, 10, seed=1))
inputs, targets = to_device(inputs, dev), to_device(targets, dev)
inputs["retrieved"] = {k: v for k, v in inputs["retrieved"].items() if k != "image"}
step = TrainStep(model, lr=1e-4, weight_decay=1e-4, max_norm=0.1)
step(inputs, targets)
inputs, targets = step.static_batch()
for _ in range(5):
    step(inputs, targets)
torch.cuda.synchronize()
for _ in range(5):
    t0 = time.perf_counter()
    step(inputs, targets)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"host call {1e3 * (t1 - t0):.2f} ms, until the device is done {1e3 * (t2 - t0):.2f} ms")
