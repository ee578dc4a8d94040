"""every ralf_gemm call of one eager train step (B = 64), each signature timed alone (HIP events, 20 back-to-back launches):
time, TFLOP/s, minimal operand bytes and GB/s.  python tools/gemm_table.py [filter: conv|all]"""
import os
import sys
from collections import OrderedDict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ralf_amd import ops  # noqa: E402
from ralf_amd.engine import TrainStep  # noqa: E402
from ralf_amd.synthetic import make_batch, to_device  # noqa: E402

dev = torch.device("cuda", 0)
B, N = 64, 10
model = bench.build_model(dev, N, "bfloat16")
inputs, targets = model.preprocess(make_batch(B, N, seed=1))
inputs, targets = to_device(inputs, dev), to_device(targets, dev)
inputs["retrieved"] = {k: v for k, v in inputs["retrieved"].items() if k != "image"}
step = TrainStep(model, lr=1e-4, weight_decay=1e-4, max_norm=0.1, use_graph=False)
step(inputs, targets)

calls = []
real = ops.gemm


def rec(A, Bm, M, Nn, K, **kw):
    calls.append((A, Bm, M, Nn, K, dict(kw)))
    return real(A, Bm, M, Nn, K, **kw)


ops.gemm = rec
step(inputs, targets)
torch.cuda.synchronize()
ops.gemm = real


def sig(c):
    A, Bm, M, Nn, K, kw = c
    g = kw.get("conv")
    gk = (g["KH"], g["stride"], g["SH"], g["SC"], g["mode"]) if g else None
    return (M, Nn, K, kw.get("a_kcontig", True), kw.get("b_kcontig", True), kw.get("gather", 0), kw.get("splitk", 1), gk, kw.get("batch", (1, 1)),
            kw.get("res") is not None, kw.get("colstats") is not None, kw.get("accumulate", False), kw.get("out2") is not None, kw.get("aux") is not None,
            str(A.dtype).replace("torch.", ""))


groups = OrderedDict()
for c in calls:
    groups.setdefault(sig(c), []).append(c)

rows = []
for s, cs in groups.items():
    A, Bm, M, Nn, K, kw = cs[0]
    kw = dict(kw)
    if kw.get("out") is None:
        kw["out"] = real(A, Bm, M, Nn, K, **kw)
    fn = lambda: real(A, Bm, M, Nn, K, **kw)
    t = bench._time_gpu(fn, iters=20, warm=3)
    nb = s[8][0] * s[8][1]
    es = A.element_size()
    g = kw.get("conv")
    if g and s[5] in (1, 4):
        a_bytes = (M // (g["RH"] * g["RW"])) * g["SH"] * g["SW"] * g["SC"] * es if g["mode"] == 0 else M * K // (g["KH"] * g["KW"]) * es
    elif g and s[5] == 2:
        a_bytes = K * M * es
    else:
        a_bytes = M * K * es * nb
    b_bytes = Nn * K * es * (nb if kw.get("sB", (0, 0)) != (0, 0) else 1)
    if g and s[5] == 2:
        b_bytes = K * Nn * es // (g["KH"] * g["KW"])
    c_bytes = M * Nn * kw["out"].element_size() * nb * (2 if s[11] else 1)
    r_bytes = M * Nn * es * nb if s[9] else 0
    by = a_bytes + b_bytes + c_bytes + r_bytes
    fl = 2.0 * M * Nn * K * nb
    rows.append((len(cs) * t, len(cs), t, s, fl, by))

rows.sort(key=lambda r: -r[0])
tot = sum(r[0] for r in rows)
print(f"{len(calls)} gemm calls per step, {len(rows)} signatures, sum of isolated times {tot * 1e3:.2f} ms")
print(f"{'n':>3s} {'us':>7s} {'tot_us':>8s} {'TF/s':>6s} {'GB/s':>6s} {'us@5TB/s':>8s} {'us@1.2PF':>8s}  M,N,K | aK bK gather splitk conv(k,stride,SH,SC,mode) batch res stats acc out2 aux dtype")
for tt, n, t, s, fl, by in rows:
    print(f"{n:3d} {t * 1e6:7.1f} {tt * 1e6:8.1f} {fl / t / 1e12:6.0f} {by / t / 1e9:6.0f} {by / 5e12 * 1e6:8.1f} {fl / 1.2e15 * 1e6:8.1f}  {s[0]},{s[1]},{s[2]} | "
          f"{int(s[3])} {int(s[4])} g{s[5]} sk{s[6]} {s[7]} {s[8]} {int(s[9])} {int(s[10])} {int(s[11])} {int(s[12])} {int(s[13])} {s[14]}")
