"""B = 256 constrained decode (task c, argmax) a few times -- for rocprofv3 timelines: python3 tools/decode_once.py [reps] [bfloat16|float32]"""
import os
import sys
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from ralf_amd.engine import GraphedDecode  # noqa: E402
from ralf_amd.helpers.task import get_condition  # noqa: E402
from ralf_amd.synthetic import make_batch  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda", 0)
dtype = sys.argv[2] if len(sys.argv) > 2 else "bfloat16"
model = bench.build_model(dev, 10, dtype, "c").eval()
cond, _ = get_condition(make_batch(256, 10, seed=9), "c", model.tokenizer)
cond.retrieved = {k: v for k, v in cond.retrieved.items() if k != "image"}
cfg = {"name": "deterministic"}
dec = GraphedDecode(model, "c", cfg, True)
for _ in range(reps):
    res = model.sample(cond=cond, sampling_cfg=cfg, cond_type="c", decoder=dec)
torch.cuda.synchronize()
time.sleep(0.25)   # (an idle gap in front of the last replay: tools/prof_sequence.py lists the dispatches behind it)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
dec._graph.replay()
e1.record()
torch.cuda.synchronize()
print("graph replay ms", e0.elapsed_time(e1))
print("token checksum", int(sum((res[k].long() * (1 + i)).sum() for i, k in enumerate(("label", "mask"))).item()), float(res["center_x"].double().sum() + res["width"].double().sum()))
