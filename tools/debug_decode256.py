"""diagnose tests/test_fullsize_gpu.py::test_b256_decode_rows_equal_reference_rows: per-row mismatches at several batch sizes"""
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: E402
from test_model_cpu import build  # noqa: E402
from test_model_gpu import FeatStandIn, load_det  # noqa: E402
from ralf_amd.engine import GraphedDecode  # noqa: E402
from ralf_amd.helpers.task import RetrievalAugmentedConditionalInputsForDiscreteLayout as Cond  # noqa: E402
import numpy as np  # noqa: E402


task = sys.argv[1] if len(sys.argv) > 1 else "c"
r = conftest.Golden("sample.npz").sub(task)
for B in (3, 6, 48, 64, 96, 128, 256):
    reps = (B + 2) // 3
    tile = lambda t: t.repeat((reps,) + (1,) * (t.dim() - 1))[:B]   # noqa: E731
    model = load_det(build(task=task), "ralf_state_shapes.json").eval()
    model.encoder = FeatStandIn(tile(r["feat"]).cuda())
    cond = Cond(image=torch.zeros(B, 4, 8, 8), task=task, seq=tile(r["cond_seq"]), mask=None, retrieved={k: tile(v) for k, v in r["retrieved"].items()})
    model._create_encoder_inputs = lambda c: ({"image": c.image, "retrieved": c.retrieved, "seq_layout_const": tile(r["seq_layout_const"]),
                                               "seq_layout_const_pad_mask": tile(r["seq_layout_const_pad_mask"])}, None)
    graphed = GraphedDecode(model, task, {"name": "deterministic"})
    for name, dec in (("eager", None), ("graph", graphed), ("graph2", graphed)):
        out = model.sample(cond=cond, sampling_cfg={"name": "deterministic"}, cond_type=task, return_violation=False, use_kv_cache=True, decoder=dec)
        bad = {}
        for k in ("label", "mask", "center_x", "center_y", "width", "height"):
            want = tile(r["result"][k])
            ne = (out[k] != want)
            rows = ne.view(B, -1).any(1).nonzero().flatten().tolist()
            if rows:
                bad[k] = (len(rows), rows[:12])
        print(f"B={B:4d} {name:7s} mismatching rows per key: {bad if bad else 'none'}", flush=True)
