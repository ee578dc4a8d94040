"""The batch-1 decoder step of sample_relation's sequential loop: per-position graphs fed by pageable copies (_StepGraphs) against ONE graph with
per-element positions fed from pinned mirrors (_LockstepStep at B = 1), and the graph replay alone.
    python tools/relation_step_probe.py [dtype]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import ralf_amd  # noqa: E402,F401
import bench  # noqa: E402
from ralf_amd import nn as RN  # noqa: E402


def main():
    dtype = sys.argv[1] if len(sys.argv) > 1 else "bfloat16"
    dev = torch.device("cuda:0")
    model = bench.build_model(dev, 10, dtype, "relation" if False else "c").eval()
    T = model.tokenizer.max_token_length
    with torch.no_grad():
        model.rt.to(dev).begin_step()
        memory = torch.randn(1, 532, 256, device=dev).to(model.rt.dtype)
        cache = RN.decoder_init_cache(model.decoder, memory, model.rt, T)
        old = model._StepGraphs(model, T, dev)
        old.bind(cache)
        new = model._LockstepStep(model, T, dev, 1)
        new.bind(RN.decoder_init_cache(model.decoder, memory, model.rt, T))
        seq = torch.randint(0, 100, (1, T + 1))
        for _ in range(2):   # capture
            for L in range(1, T + 1):
                old(seq[:, L - 1].to(dev), L - 1, torch.zeros(1, L, dtype=torch.uint8).to(dev)).float().cpu()
                new.tok_h[0], new.pos_h[0] = seq[0, L - 1], L - 1
                new.kpm_h[0, :L] = 0
                new()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for rep in range(4):
            for L in range(1, T + 1):
                old(seq[:, L - 1].to(dev), L - 1, torch.zeros(1, L, dtype=torch.uint8).to(dev)).float().cpu()
        t_old = (time.perf_counter() - t0) / (4 * T)
        t0 = time.perf_counter()
        for rep in range(4):
            for L in range(1, T + 1):
                new.tok_h[0], new.pos_h[0] = seq[0, L - 1], L - 1
                new.kpm_h[0, :L] = 0
                new.kpm_h[0, L:] = 1
                new()
        t_new = (time.perf_counter() - t0) / (4 * T)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            new.graph.replay()
        e1.record()
        torch.cuda.synchronize()
        print(f"{dtype}: per-position graphs + pageable copies {t_old * 1e6:.0f} us per step; one graph + pinned mirrors {t_new * 1e6:.0f} us per step; "
              f"graph replay alone {e0.elapsed_time(e1) * 10:.0f} us")


if __name__ == "__main__":
    main()
