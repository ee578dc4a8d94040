"""Where the relation decode's time goes (bench_relation's workload at a smaller batch): decoder steps per sample, device step
(graph replay + the logits' way back) against the host-side masks, and how early a sample needs its first `random` draw.
    python tools/relation_probe.py [B] [head scale] [lockstep 0|1]"""
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import ralf_amd  # noqa: E402,F401
import bench  # noqa: E402
from ralf_amd.helpers import relation_restriction as RR  # noqa: E402
from ralf_amd.models import ralf as M  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    sharpen = float(sys.argv[2]) if len(sys.argv) > 2 else None
    lockstep = {"1": True, "0": False}.get(sys.argv[3]) if len(sys.argv) > 3 else None
    dev = torch.device("cuda:0")
    acc = {"step_s": 0.0, "steps": 0, "con_s": 0.0, "draws": 0, "first_draw_step": []}
    SG = M._GeneratorBase._StepGraphs
    call0 = SG.__call__

    def timed_call(self, token, pos, kpm):
        t0 = time.perf_counter()
        out = call0(self, token, pos, kpm).float().cpu()   # the caller's .float().cpu() is then free
        acc["step_s"] += time.perf_counter() - t0
        acc["steps"] += 1
        return out
    SG.__call__ = timed_call
    bind0 = SG.bind

    def bind(self, cache):
        acc["sample_start_step"] = acc["steps"]
        acc["sample_drew"] = False
        return bind0(self, cache)
    SG.bind = bind
    rc0 = RR.RelationConstraint.__call__

    def rc(self, seq, rel):
        t0 = time.perf_counter()
        out = rc0(self, seq, rel)
        acc["con_s"] += time.perf_counter() - t0
        return out
    RR.RelationConstraint.__call__ = rc
    ri0 = random.randint

    def ri(a, b):
        acc["draws"] += 1
        if not acc.get("sample_drew"):
            acc["sample_drew"] = True
            acc["first_draw_step"].append(acc["steps"] - acc.get("sample_start_step", 0))
        return ri0(a, b)
    random.randint = ri
    t0 = time.perf_counter()
    out = bench.bench_relation(dev, 10, B, sharpen=sharpen, lockstep=lockstep)
    wall = time.perf_counter() - t0
    n = max(acc["steps"], 1)
    print(f"B={B} sharpen={sharpen} lockstep={lockstep}: {out['ms_per_sample']:.2f} ms per sample, violated {out['relations_violated']} of {out['relations_checked']}; {acc['steps']} decoder steps incl. warm-up ({acc['steps'] / (B + 4):.0f} per sample)")
    print(f"device step + logits to the host: {acc['step_s'] / n * 1e6:.0f} us per step ({acc['step_s']:.2f} s); relation masks: {acc['con_s'] / n * 1e6:.0f} us per step "
          f"({acc['con_s']:.2f} s); wall {wall:.2f} s incl. model build")
    fd = sorted(acc["first_draw_step"])
    print(f"random draws: {acc['draws']}; samples that draw at all: {len(fd)} of {B + 4}; first draw after steps (sorted): {fd[:8]} ... median {fd[len(fd) // 2] if fd else None}")


if __name__ == "__main__":
    main()
