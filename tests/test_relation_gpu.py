"""GPU: relation-constrained decoding with back-tracking (model.sample(cond_type="relation")) reproduces the tokens the
reference's sample_relation produced for the same weights, condition, relationship table and `random` seeds
(tests/golden/relation.npz "sample", deterministic sampling)."""
import random

import pytest
import torch

from test_model_gpu import FeatStandIn, build, load_det
from test_relation_cpu import LABELS, load_table, make_batch

pytestmark = pytest.mark.gpu


def test_sample_relation_matches_reference(golden):
    from ralf_amd.helpers.layout_tokenizer import LayoutSequenceTokenizer
    from ralf_amd.helpers.task import get_condition

    g = golden("relation.npz")
    r = g.sub("sample")
    table, _ = load_table()
    random.seed(6)                                         # the constructor shuffles the table with `random` (like the reference)
    model = load_det(build(task="relation", relation_table=table), "ralf_state_shapes.json").eval()
    model.encoder = FeatStandIn(r["feat"].cuda())
    tok = LayoutSequenceTokenizer(LABELS, 10)
    batch = make_batch(g.sub("compute_relation"))
    batch["retrieved"] = [dict(r["retrieved"], image=torch.zeros(r["feat"].shape[0], 16, 4, 1, 1))]
    random.seed(8)
    torch.manual_seed(8)
    cond, _ = get_condition(batch, "relation", tok)
    random.seed(10)
    torch.manual_seed(10)
    out, vio = model.sample(cond=cond, sampling_cfg={"name": "deterministic", "temperature": 1.0}, cond_type="relation", return_violation=True,
                            use_backtrack=True, RELATION_SIZE=30)
    assert torch.equal(cond.seq, r["cond_seq"])           # (the constraint serialisation rewrites eos -> pad in place, like the reference)
    for k in ("label", "mask", "center_x", "center_y", "width", "height"):
        assert torch.equal(out[k], r["result"][k]), k
    assert (vio["total"], vio["viorated"]) == (int(r["violation"]["total"]), int(r["violation"]["viorated"]))
    # without back-tracking only the label order is enforced; relations are scored afterwards
    random.seed(10)
    torch.manual_seed(10)
    out2, vio2 = model.sample(cond=cond, sampling_cfg={"name": "deterministic"}, cond_type="relation", return_violation=True, use_backtrack=False)
    n = out["mask"].sum(1)
    assert torch.equal(out2["mask"].sum(1), n) and vio2["total"] > 0


@pytest.mark.parametrize("sharpen", [None, 6.0])
def test_lockstep_exact_mode_equals_the_sequential_order_where_samples_draw(sharpen, monkeypatch):
    """the exact-order relation decode in LOCK-STEP (the default) on the benchmark's synthetic relationship workload, where
    every sample draws from `random` (~120 draws per sample; a third of the samples draw from ranges of more than one value and wait for the
    samples before them; the others' randint(2, 2) draws are deferred and replayed in sample order).  Yardstick: the same decode with ONLY the
    lowest unfinished sample stepping (RALF_RELATION_SERIAL=1) -- the sequential loop's order on the same batched decoder step, so the
    comparison is free of the last-bit differences between the batch-1 and the batched kernels that flip near-tied logits of a random-init
    model (the batch-1 loop itself is compared on the golden model: tests/test_configs_gpu.py, and the order logic against the
    sample-after-sample loop on a fake decoder: tests/test_relation_cpu.py).  Same tokens, same randint calls with the same arguments and
    results in the same order, `random` left in the same state; the memo leaves the device a small fraction of the steps."""
    import os
    import sys
    import time

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    B = 64
    model, cond, sub = bench.relation_workload(torch.device("cuda", 0), 10, B, "bfloat16", sharpen)
    cfg = {"name": "deterministic", "temperature": 1.0}
    real = random.randint
    runs = {}
    for mode in ("serial", "lockstep"):
        calls = []

        def counting(a, b, _calls=calls):
            v = real(a, b)
            _calls.append((a, b, v))
            return v
        random.seed(77)
        torch.manual_seed(77)                # (the constraint serialisation spends torch.randperm draws, like the reference)
        monkeypatch.setenv("RALF_RELATION_SERIAL", "1" if mode == "serial" else "0")
        random.randint = counting
        t0 = time.perf_counter()
        try:
            out, vio = model.sample(cond=sub(cond, B), sampling_cfg=cfg, cond_type="relation", return_violation=True, use_backtrack=True)
        finally:
            random.randint = real
        runs[mode] = (out, vio, calls, random.getstate(), dict(model.relation_stats), time.perf_counter() - t0)
    (o1, v1, c1, s1, st1, t1), (o2, v2, c2, s2, st2, t2) = runs["serial"], runs["lockstep"]
    wide = sum(1 for a, b, _ in c1 if b > a)
    print(f"sharpen={sharpen}: {len(c1)} draws in {B} samples, {wide} of them with more than one possible value; violations {v1}; "
          f"serial {t1:.2f} s {st1}; lock-step {t2:.2f} s {st2}")
    assert len(c1) >= 0.1 * B and wide > 0           # a batch where samples draw, also from real ranges
    assert c1 == c2 and s1 == s2 and v1 == v2
    for k in ("label", "mask", "center_x", "center_y", "width", "height"):
        assert torch.equal(o1[k], o2[k]), k
    assert st2["device_steps"] < 0.2 * st2["iterations"] and st2["device_steps"] < st1["device_steps"]
