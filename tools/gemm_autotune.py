"""Record every distinct ralf_gemm call of one eager train step and time each on the current tile choice
(RALF_GEMM_TILE=11 / 22 pins one): python tools/gemm_autotune.py out.json"""
import json
import sys

import torch

sys.path.insert(0, "."); sys.path.insert(0, "tools")
import bench
from gemm_bench import timeit
from ralf_amd import ops
from ralf_amd.engine import TrainStep
from ralf_amd.synthetic import make_batch, to_device

dev = torch.device("cuda")
model = bench.build_model(dev, 10, "bfloat16")
inputs, targets = model.preprocess(make_batch(64, 10, seed=1))
inputs, targets = to_device(inputs, dev), to_device(targets, dev)
inputs["retrieved"] = {k: v for k, v in inputs["retrieved"].items() if k != "image"}
step = TrainStep(model, use_graph=False, overlap_wgrad=False)
step(inputs, targets)
calls, orig = {}, ops.gemm


def rec(A, B, M, N, K, **kw):
    conv = kw.get("conv")
    key = (M, N, K, int(kw.get("a_kcontig", True)), int(kw.get("b_kcontig", True)), int(kw.get("gather", 0) or 0), int(kw.get("splitk", 1) or 1),
           int(kw.get("colstats") is not None), tuple(sorted(conv.items())) if conv else (), tuple(kw.get("batch", (1, 1))))
    out = orig(A, B, M, N, K, **kw)
    if key not in calls:
        kw2 = dict(kw)
        if kw2.get("out") is None:
            kw2["out"] = out
        calls[key] = [0, (A, B, M, N, K, kw2)]
    calls[key][0] += 1
    return out


ops.gemm = rec
step(inputs, targets)
ops.gemm = orig
torch.cuda.synchronize()
res = []
for key, (cnt, (A, B, M, N, K, kw)) in calls.items():
    kw = dict(kw)
    kw.pop("accumulate", None)   # timing only
    try:
        t = timeit(lambda: orig(A, B, M, N, K, **kw), iters=20)
    except Exception as e:  # noqa: BLE001
        t = float("nan")
        print("skip", key, e)
    res.append({"key": [list(k) if isinstance(k, tuple) else k for k in key], "count": cnt, "us": t * 1e6})
res.sort(key=lambda r: -r["us"] * r["count"])
json.dump(res, open(sys.argv[1], "w"))
tot = sum(r["us"] * r["count"] for r in res if r["us"] == r["us"])
print(f"{len(res)} distinct products, {sum(r['count'] for r in res)} calls, {tot / 1e3:.2f} ms per step if run back to back")
