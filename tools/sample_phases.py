"""where the time of one `sample()` call goes at B = 256 (task c, argmax, bf16): the steps of models/ralf.py: sample() one by one, wall clock with a device
synchronisation after each:  python3 tools/sample_phases.py [reps]"""
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

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda", 0)
model = bench.build_model(dev, 10, "bfloat16", "c").eval()
cond, _ = get_condition(make_batch(256, 10, seed=9), "c", model.tokenizer)
cond.retrieved = {k: v for k, v in cond.retrieved.items() if k != "image"}
cfg = {"name": "deterministic"}
dec = GraphedDecode(model, "c", cfg, True)
for _ in range(3):
    model.sample(cond=cond, sampling_cfg=cfg, cond_type="c", decoder=dec)
torch.cuda.synchronize()
T = {}


def lap(name, t0):
    torch.cuda.synchronize()
    T.setdefault(name, []).append((time.perf_counter() - t0) * 1e3)
    return time.perf_counter()


whole_only = len(sys.argv) > 2 and sys.argv[2] == "whole"
for _ in range(reps):
    if whole_only:
        t0 = time.perf_counter()
        res = model.sample(cond=cond, sampling_cfg=cfg, cond_type="c", decoder=dec)
        lap("whole sample() call", t0)
        continue
    t = time.perf_counter()
    piped = dec.upload_image(cond.image)
    up = None if piped else model._start_image_upload(cond.image)
    T.setdefault("image upload issued (host staging copy incl.)", []).append((time.perf_counter() - t) * 1e3)
    t = time.perf_counter()
    enc_in, seqc = model._create_encoder_inputs(cond)
    T.setdefault("_create_encoder_inputs (host, beside the copy)", []).append((time.perf_counter() - t) * 1e3)
    t = time.perf_counter()
    if piped:
        enc_in = dict(enc_in, image=dec.static_image())
    elif up is not None:
        torch.cuda.current_stream().wait_event(up[1])
        enc_in = dict(enc_in, image=up[0])
    enc_dev = {k: ({kk: vv.to(dev) for kk, vv in v.items() if torch.is_tensor(vv)} if isinstance(v, dict) else (v.to(dev) if torch.is_tensor(v) else v)) for k, v in enc_in.items()}
    cond_seq = cond.seq.to(dev) if cond.seq is not None else None
    t = lap("rest of the image copy + the small host -> device copies", t)
    tokens = dec(enc_dev, cond_seq)
    t = lap("copies into the graph's buffers + replay", t)
    out = tokens.cpu()
    t = lap("tokens.cpu()", t)
    res = model.postprocess({"seq": out})
    t = lap("postprocess (host)", t)
    t0 = time.perf_counter()
    model.sample(cond=cond, sampling_cfg=cfg, cond_type="c", decoder=dec)
    lap("whole sample() call", t0)
piped = piped if not whole_only else None
enc_dev = enc_dev if not whole_only else {"image": dec.static_image()}
print("RALF_DECODE_GATES =", os.environ.get("RALF_DECODE_GATES", "2"), " piped:", piped, " RALF_UPLOAD_LP =", os.environ.get("RALF_UPLOAD_LP", "0"), " image on the device:", tuple(enc_dev["image"].shape), enc_dev["image"].dtype, f"({enc_dev['image'].numel() * enc_dev['image'].element_size() / 1e6:.0f} MB over the host link)")
print("copy stream picked:", dec._cs_pick, {k: [round(x, 2) for x in v] for k, v in dec._cs_ms.items()})
print("token checksum", int(sum((res[k].long() * (1 + i)).sum() for i, k in enumerate(("label", "mask"))).item()), float(res["center_x"].double().sum() + res["width"].double().sum()))
for k, v in T.items():
    v = sorted(v)
    print(f"{k:58s} median {v[len(v) // 2]:7.2f} ms   min {v[0]:7.2f}")
