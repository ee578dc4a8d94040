"""why the train step runs slower inside the reference's loop than in bench.py's back-to-back replays: one factor at a time (ms per step, wall clock
around 10 steps with a final synchronize; modes see the table printed)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ralf_amd.engine import GraphedAdamW, TrainStep  # noqa: E402
from ralf_amd.synthetic import make_batch, to_device  # noqa: E402

dev = torch.device("cuda", 0)
B, N = 64, 10


def timed(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def fresh(model, seed=1):
    inputs, targets = model.preprocess(make_batch(B, N, seed=seed))
    return to_device(inputs, dev), to_device(targets, dev)


res = {}
for mode in sys.argv[1:] or ["A", "B", "C", "E", "F"]:
    model = bench.build_model(dev, N, "bfloat16")
    inputs, targets = fresh(model)
    if mode in "ABCG":
        step = TrainStep(model, lr=1e-4, weight_decay=1e-4, max_norm=0.1, own_stream=(mode == "G"))
        step(inputs, targets)
        si, st = step.static_batch()
        alt = [fresh(model, 2), fresh(model, 3)]
        torch.cuda.synchronize()
        if mode == "A":      # bench.py: the static batch, back to back
            res["A static batch, back to back"] = timed(lambda: step(si, st))
        elif mode == "B":    # fresh device tensors every step (copied into the static buffers)
            it = [0]
            def f():
                it[0] += 1
                step(*alt[it[0] % 2])
            res["B fresh device batch per step"] = timed(f)
        elif mode == "C":    # + host sync every step
            def f():
                step(si, st).item()
            res["C static batch, loss.item() every step"] = timed(f)
        elif mode == "G":
            res["G static batch, own stream"] = timed(lambda: step(si, st))
    else:
        lag = 1 if mode in "F" else 0
        opt = GraphedAdamW(params=model.optim_groups(base_lr=1e-4, weight_decay=1e-4, custom_lr={"encoder.extractor.body": 1e-5}), max_norm=0.1, loss_lag=lag,
                           own_stream=(mode != "H"))
        def f():
            out, losses = model.train_loss(inputs, targets)
            return losses["nll_loss"]
        if mode == "E":
            res["E adapter train_loss only, lag 0, device inputs, .item()"] = timed(lambda: f().item())
        elif mode == "F":
            res["F adapter train_loss only, lag 1"] = timed(f)
        elif mode == "H":
            res["H adapter, default stream, lag 0, .item()"] = timed(lambda: f().item())
        elif mode in "IJKL":
            batches = [make_batch(B, N, seed=21 + i) for i in range(3)]
            it = [0]
            if mode == "J":
                model._engine_saved, model._engine = model._engine, None     # preprocess without the upload: host tensors, the loop's own .to(device)
            def g():
                it[0] += 1
                if mode == "J":
                    model._engine = None
                i2, t2 = model.preprocess(batches[it[0] % 3])
                if mode == "J":
                    model._engine = model._engine_saved
                i2 = {k: v.to(dev) if torch.is_tensor(v) else v for k, v in i2.items()}
                t2 = {k: v.to(dev) if torch.is_tensor(v) else v for k, v in t2.items()}
                if mode in "KL":
                    model.zero_grad()
                out, losses = model.train_loss(i2, t2)
                loss = sum(losses.values())
                if mode in "KL":
                    loss.backward()
                if mode == "L":
                    torch.nn.utils.clip_grad_norm_(model.parameters(), 0.1)
                    opt.step()
                return loss.cpu().item()
            name = {"I": "I adapter + preprocess(upload on the priority stream) + .item()", "J": "J adapter + preprocess(host) + loop .to(device) + .item()",
                    "K": "K = I + zero_grad + backward", "L": "L = K + clip_grad_norm_ + opt.step (the whole loop body)"}[mode]
            res[name] = timed(g)
    del model
for k, v in res.items():
    print(f"{k:60s} {v:7.2f} ms")
