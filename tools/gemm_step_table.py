"""The heaviest matrix products of ONE train step (B = 64, N = 10, 256 x 256, bf16), each timed in ISOLATION against the bound that applies to it
(VERDICT r5 item 1):  python tools/gemm_step_table.py [out.txt] [top=24]

An eager step is run with ops.gemm / ops.conv1x1_k64 / ops.conv3x3_wgrad recorded (the grouped weight-gradient launches and the fused transformer-
layer kernels are not `gemm` calls and are listed by the kernel profile instead); every distinct signature is then re-issued on the recorded
operands from a hipGraph of back-to-back launches (tools/gemm_bench.timeit).
  flop   = 2 M N K (stride-2 data gradients: a quarter of that is real work -- column `useful`)
  bytes  = the operands and results the call must move once: A (the SOURCE tensor of a gathered operand), B, C, residual, BatchNorm-backward reads
  bound  = max(flop / 2.5 PFLOP/s, bytes / 6.3 TB/s)  (dense bf16 MFMA peak; achievable HBM rate, MI355X_MICROARCH.md); frac = bound / measured
(inside the step most operands of the small products come from the infinity cache, so `bytes / 6.3 TB/s` is a pessimistic bound there)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench  # noqa: E402
from gemm_bench import timeit  # noqa: E402
from ralf_amd import ops  # noqa: E402
from ralf_amd.engine import TrainStep  # noqa: E402
from ralf_amd.synthetic import make_batch, to_device  # noqa: E402

PEAK, HBM = 2.5e15, 6.3e12


def nbytes(t):
    return 0 if t is None else t.numel() * t.element_size()


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else None
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    dev = torch.device("cuda", 0)
    B, N = 64, 10
    model = bench.build_model(dev, N, "bfloat16")
    inputs, targets = model.preprocess(make_batch(B, N, seed=1))
    inputs, targets = to_device(inputs, dev), to_device(targets, dev)
    inputs["retrieved"] = {k: v for k, v in inputs["retrieved"].items() if k != "image"}
    step = TrainStep(model, lr=1e-4, weight_decay=1e-4, max_norm=0.1, use_graph=False)
    step(inputs, targets)                       # warm-up (allocator, lazy initialisation)
    torch.cuda.synchronize()
    calls = {}
    real = {"gemm": ops.gemm, "conv1x1_k64": ops.conv1x1_k64, "conv3x3_wgrad": ops.conv3x3_wgrad}

    def rec(kind, key, fl, by, note, fn):
        e = calls.setdefault((kind,) + key, {"n": 0, "flop": fl, "bytes": by, "note": note, "fn": fn})
        e["n"] += 1

    def gemm(A, Bm, M, Nn, K, **kw):
        out = real["gemm"](A, Bm, M, Nn, K, **kw)
        conv, bnb, nb = kw.get("conv"), kw.get("bnb"), kw.get("batch", (1, 1))
        nbat = nb[0] * nb[1]
        g = (conv["KH"], conv["stride"], conv["SH"], conv["SC"], conv["mode"]) if conv else None
        flags = "".join(c for c, on in (("b", kw.get("bias") is not None), ("r", kw.get("res") is not None), ("s", kw.get("colstats") is not None), ("n", bnb is not None),
                                        ("a", kw.get("accumulate", False)), ("d", kw.get("drop_p", 0.0) > 0), ("x", kw.get("aux") is not None), ("t", kw.get("at") is not None),
                                        ("2", kw.get("out2") is not None)) if on)
        key = (M, Nn, K, int(kw.get("a_kcontig", True)), int(kw.get("b_kcontig", True)), kw.get("gather", 0), kw.get("splitk", 1), g, nbat, flags, str(out.dtype)[6:])
        by = nbytes(A) + nbytes(Bm) + M * Nn * nbat * out.element_size() * (2 if kw.get("accumulate") else 1) + nbytes(kw.get("res")) + nbytes(kw.get("aux")) + nbytes(kw.get("out2"))
        if bnb is not None:
            by += nbytes(bnb[0]) + nbytes(bnb[1])
        if kw.get("at") is not None:
            by += nbytes(kw["at"].get("a2")) + nbytes(kw["at"].get("out"))
        useful = 0.25 if (conv and conv["mode"] == 1 and conv["stride"] == 2) else 1.0
        kw2 = dict(kw, out=out)
        rec("gemm", key, 2.0 * M * Nn * K * nbat, by, f"useful {useful:g}" if useful < 1 else "", lambda: real["gemm"](A, Bm, M, Nn, K, **kw2))
        return out

    def conv1x1_k64(x2d, W, colstats=None, scale=None, shift=None, res=None, relu=0, out=None):
        y = real["conv1x1_k64"](x2d, W, colstats, scale, shift, res, relu, out)
        M, K, Nn = x2d.shape[0], x2d.shape[1], W.shape[0]
        rec("conv1x1_k64", (M, Nn, K, "s" if colstats is not None else ""), 2.0 * M * Nn * K, nbytes(x2d) + nbytes(W) + nbytes(y) + nbytes(res), "",
            lambda: real["conv1x1_k64"](x2d, W, colstats, scale, shift, res, relu, y))
        return y

    def conv3x3_wgrad(dy, x, out=None, stride=1):
        r = real["conv3x3_wgrad"](dy, x, out=out, stride=stride)
        Bn, H, W_, Co = dy.shape
        Ci = x.shape[3]
        rec("conv3x3_wgrad", (Bn * H * W_, Co, Ci, stride), 2.0 * Bn * H * W_ * Co * 9 * Ci, nbytes(dy) + nbytes(x) + Co * Ci * 9 * 4, "direct form + its reduce",
            lambda: real["conv3x3_wgrad"](dy, x, out=r, stride=stride))
        return r
    ops.gemm, ops.conv1x1_k64, ops.conv3x3_wgrad = gemm, conv1x1_k64, conv3x3_wgrad
    try:
        step(inputs, targets)
        torch.cuda.synchronize()
    finally:
        ops.gemm, ops.conv1x1_k64, ops.conv3x3_wgrad = real["gemm"], real["conv1x1_k64"], real["conv3x3_wgrad"]
    rows = []
    for key, e in calls.items():
        t = timeit(e["fn"], iters=20)
        bound = max(e["flop"] / PEAK, e["bytes"] / HBM)
        rows.append((e["n"] * t, e["n"], t, e["flop"], e["bytes"], bound, key, e["note"]))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    lines = ["# python tools/gemm_step_table.py (MI355X; B = 64, N = 10, 256 x 256, bf16: the `gemm` / conv1x1_k64 / conv3x3_wgrad calls of one train step, isolated times)",
             f"# {sum(r[1] for r in rows)} calls per step, {len(rows)} signatures, sum of isolated times {tot * 1e3:.2f} ms; bound = max(flop / 2.5 PFLOP/s, bytes / 6.3 TB/s)",
             f"{'n':>3s} {'us':>7s} {'tot_us':>8s} {'TFLOP/s':>8s} {'GB/s':>6s} {'bound_us':>8s} {'frac':>5s} {'by':>4s}  signature (kind, M, N, K, aK, bK, gather, splitk, conv(k, stride, SH, SC, mode), batch, flags, out)"]
    for tot_t, n, t, fl, by, bound, key, note in rows[:top]:
        lines.append(f"{n:3d} {t * 1e6:7.1f} {tot_t * 1e6:8.1f} {fl / t / 1e12:8.0f} {by / t / 1e9:6.0f} {bound * 1e6:8.1f} {bound / t:5.2f} {'mfma' if fl / PEAK >= by / HBM else 'hbm':>4s}  {key} {note}")
    rest = rows[top:]
    lines.append(f"... {len(rest)} more signatures, {sum(r[0] for r in rest) * 1e3:.2f} ms per step together")
    print("\n".join(lines))
    if out_path:
        with open(out_path, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
