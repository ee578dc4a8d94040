"""which host call sites issue the small torch kernels (copy_/fill_/add) inside one eager train step."""
import os, sys, collections, traceback
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_model
from ralf_amd.engine import TrainStep
from ralf_amd.synthetic import make_batch, to_device
from torch.utils._python_dispatch import TorchDispatchMode

dev = torch.device("cuda", 0)
model = build_model(dev, 10, "bfloat16")
inputs, targets = model.preprocess(make_batch(64, 10, seed=1))
inputs, targets = to_device(inputs, dev), to_device(targets, dev)
inputs["retrieved"] = {k: v for k, v in inputs["retrieved"].items() if k != "image"}
step = TrainStep(model, use_graph=False)
step(inputs, targets)

sites = collections.Counter()
class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if any(s in name for s in ("copy_", "clone", "fill_", "zero_", "zeros", "add", "cat", "contiguous", "_to_copy", "mul", "ones")):
            st = [f for f in traceback.extract_stack() if "/ralf_amd/" in f.filename or "bench" in f.filename]
            where = " < ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in st[-3:][::-1])
            sites[(name, where)] += 1
        return func(*args, **(kwargs or {}))
with Spy():
    step(inputs, targets)
torch.cuda.synchronize()
for (name, where), n in sites.most_common(60):
    print(f"{n:5d}  {name:32s} {where}")
