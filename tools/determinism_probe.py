"""Run-to-run reproducibility of one train step's gradients: the same weights and batch twice (and across the direct-form kernels'
switches); lists the parameters whose gradients are not bit-identical.
    python tools/determinism_probe.py [B]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import ralf_amd  # noqa: E402,F401
import bench  # noqa: E402
from ralf_amd.synthetic import make_batch, to_device  # noqa: E402


def grads(model, inputs, tgt):
    for p in model.parameters():
        p.grad = None
    model.rt.to(next(model.parameters()).device).begin_step()
    _, losses = model._train_loss(inputs, tgt)
    losses["nll_loss"].backward()
    torch.cuda.synchronize()
    return {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}, float(losses["nll_loss"])


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    dev = torch.device("cuda:0")
    model = bench.build_model(dev, 10, "bfloat16")
    inputs, tgt = model.preprocess(make_batch(B, 10, seed=7))
    inputs, tgt = to_device(inputs, dev), to_device(tgt, dev)
    inputs["retrieved"] = {k: v for k, v in inputs["retrieved"].items() if k != "image"}

    def run():   # (an eager pass does not advance the device seed: both runs draw the same dropout masks)
        return grads(model, inputs, tgt)
    for flags in ({}, {"conv_wgrad_direct": False}, {"stem_direct": False}, {"fused_stem": False}):
        for k, v in flags.items():
            setattr(model.rt, k, v)
        (a, la), (b, lb) = run(), run()
        diff = [(n, float((a[n].float() - b[n].float()).abs().max()), float(a[n].float().abs().max())) for n in a if not torch.equal(a[n], b[n])]
        print(f"flags {flags or 'default'}: loss {la!r} / {lb!r}; {len(diff)} of {len(a)} gradients differ between two runs")
        mats = [x for x in diff if a[x[0]].dim() > 1]
        print(f"    of them {len(mats)} are matrices / filters (the rest: biases, LayerNorm / BatchNorm parameters -- column sums through fp32 atomics)")
        for n, d, m in mats[:12] + [x for x in diff if a[x[0]].dim() <= 1][:3]:
            print(f"    {n} {tuple(a[n].shape)}: max |diff| {d:.3e} (max |g| {m:.3e})")
        for k in flags:
            setattr(model.rt, k, True)


if __name__ == "__main__":
    main()
