"""Host path of the `relation` task against vectors recorded from the reference (tests/golden/relation.{npz,json},
make_golden.py golden_relation): compute_relation, RelationshipPreprocessor, and every per-step vocabulary mask /
back-track target of the relation-constrained decoder, including history rewinds."""
import json
import os
import random

import numpy as np
import pytest
import torch

from ralf_amd.helpers.layout_tokenizer import LayoutSequenceTokenizer
from ralf_amd.helpers.relation_restriction import RelationConstraint
from ralf_amd.helpers.relationships import (RelElement, RelLoc, RelSize, compute_relation, detect_loc_relation_between_elements,
                                            detect_size_relation, relationship_table)
from ralf_amd.helpers.task import get_condition
from ralf_amd.helpers.task_preprocessor import PREPROCESSOR

HERE = os.path.dirname(os.path.abspath(__file__))
LABELS = ["text", "logo", "underlay"]


def _rel(v: int):
    return RelSize(v) if v <= 3 else RelLoc(v)


def load_table(suffix=""):
    with open(os.path.join(HERE, "golden", "relation.json")) as f:
        js = json.load(f)
    table = {k: [[e[0], RelElement(e[1]), _rel(e[2]), e[3], (e[4] if isinstance(e[4], str) else RelElement(e[4]))] for e in v] for k, v in js["table" + suffix].items()}
    return table, js["constraints" + suffix]


def make_batch(r):
    b = {k: r["batch"][k].clone() for k in ("label", "mask", "center_x", "center_y", "width", "height")}
    B = b["label"].size(0)
    b["image"], b["saliency"] = torch.zeros(B, 3, 8, 8), torch.zeros(B, 1, 8, 8)
    b["id"] = [str(1000 + i) for i in range(B)]
    return b


def test_compute_relation_matches_reference(golden):
    r = golden("relation.npz").sub("compute_relation")
    random.seed(5)
    out = compute_relation(make_batch(r), edge_ratio=0.5)
    assert torch.equal(out["edge_indexes"], r["edge_indexes"]) and torch.equal(out["edge_attributes"], r["edge_attributes"])


def test_relationship_table_rules(golden):
    """the synthesiser of the (unshipped) table reproduces the table the fixtures were made from"""
    r = golden("relation.npz").sub("compute_relation")
    table, _ = load_table()
    assert relationship_table(make_batch(r), LABELS) == table
    # detectors: known answers
    assert detect_size_relation([0, 0, 1.0, 1.0], [0, 0, 1.0, 1.05]) == RelSize.EQUAL
    assert detect_size_relation([0, 0, 1.0, 1.0], [0, 0, 2.0, 1.0]) == RelSize.LARGER
    assert detect_loc_relation_between_elements([0.5, 0.8, 0.2, 0.2], [0.5, 0.2, 0.2, 0.2]) == RelLoc.TOP
    assert detect_loc_relation_between_elements([0.2, 0.5, 0.2, 0.6], [0.8, 0.5, 0.2, 0.6]) == RelLoc.RIGHT
    assert detect_loc_relation_between_elements([0.5, 0.5, 0.4, 0.4], [0.6, 0.6, 0.4, 0.4]) == RelLoc.CENTER


@pytest.fixture()
def prepared(golden):
    """same seeding protocol as the generator: table shuffle at construction, then condition + serialisation"""
    g = golden("relation.npz")
    table, cons = load_table()
    tok = LayoutSequenceTokenizer(LABELS, 10)
    random.seed(6)
    pre = PREPROCESSOR["relation"](tokenizer=tok, table=table)
    random.seed(7)
    torch.manual_seed(7)
    cond, _ = get_condition(make_batch(g.sub("compute_relation")), "relation", tok)
    pre.set_relation_size(30)
    seqc = pre(cond)
    return g, tok, pre, cond, seqc, cons


def test_relationship_preprocessor_matches_reference(prepared):
    g, tok, pre, cond, seqc, _ = prepared
    r = g.sub("preprocessor")
    assert torch.equal(cond.seq, r["cond_seq"]) and torch.equal(cond.mask, r["cond_mask"])
    assert torch.equal(cond.edge_indexes, r["edge_indexes"]) and torch.equal(cond.edge_attributes, r["edge_attributes"])
    assert torch.equal(seqc["seq"], r["seq"]) and torch.equal(seqc["pad_mask"], r["pad_mask"])
    assert pre.N_total == tok.N_total + 7 + 3 + 10 + 6 + 4
    assert pre.id_to_name(pre.name_to_id(RelLoc.TOP)) == RelLoc.TOP and pre.id_to_name(pre.name_to_id("canvas")) == "canvas"


def _replay(g, tok, pre, cons, key_pre, key_steps):
    r = g.sub(key_pre)
    rc = RelationConstraint(pre)
    steps = g.sub(key_steps)
    kinds = set()
    for b in range(r["seq"].size(0)):
        rel = rc.prepare(r["seq"][b])
        enc = [[["canvas", int(a)] if k == "canvas" else [int(k), int(a)] for k, a in lst] for lst in rel]
        assert enc == cons[str(b)], f"sample {b}: prepared constraints differ"
        s = steps[f"s{b}"]
        seq = torch.full((1, 1), tok.name_to_id("bos"))
        n_checked = 0
        for it in range(s["mask"].size(0)):
            # rebuild the prefix the generator had at this step (it may have been cut back)
            n_dec = int(s["n_decoded"][it])
            assert seq.size(1) - 1 >= n_dec
            seq = seq[:, : n_dec + 1]
            mask, back = rc(seq, rel)
            assert torch.equal(mask, s["mask"][it]), f"sample {b} step {it} (decoded {n_dec}): mask differs"
            assert (-1 if back is None else back) == int(s["back"][it]), f"sample {b} step {it}: back-track target differs"
            t = int(s["token"][it])
            if t >= 0:
                seq = torch.cat([seq, torch.tensor([[t]])], dim=1)
            n_checked += 1
        assert n_checked >= 5
        kinds |= {k for lst in rel for k, _ in lst}
    return kinds


def test_constraint_masks_match_reference(prepared):
    g, tok, pre, cond, seqc, cons = prepared
    _replay(g, tok, pre, cons, "preprocessor", "steps")


def test_constraint_masks_dense_layouts(golden):
    """6 layouts with 6-9 elements, 60 % of all relations: every relation kind, hundreds of steps with infeasible cuts"""
    g = golden("relation.npz")
    table, cons = load_table("2")
    tok = LayoutSequenceTokenizer(LABELS, 10)
    random.seed(16)
    pre = PREPROCESSOR["relation"](tokenizer=tok, table=table)
    r = g.sub("preprocessor2")
    assert relationship_table(make_batch(r), LABELS) == table
    random.seed(17)
    torch.manual_seed(17)
    cond, _ = get_condition(make_batch(r), "relation", tok)
    pre.set_relation_size(60)
    seqc = pre(cond)
    assert torch.equal(cond.seq, r["cond_seq"]) and torch.equal(seqc["seq"], r["seq"]) and torch.equal(seqc["pad_mask"], r["pad_mask"])
    kinds = _replay(g, tok, pre, cons, "preprocessor2", "steps2")
    assert {"canvas", RelSize.SMALLER, RelSize.LARGER, RelLoc.TOP, RelLoc.BOTTOM, RelLoc.LEFT, RelLoc.RIGHT, RelLoc.CENTER} <= kinds


def test_relation_task_without_table_fails_loudly():
    tok = LayoutSequenceTokenizer(LABELS, 10)
    with pytest.raises(Exception):
        PREPROCESSOR["relation"](tokenizer=tok)   # no cache/pku_cgl_relationships_dic_...pt here


def test_long_layouts_keep_every_other_task():
    """<= 32-element layouts (BASELINE north star): the reference cannot build its constraint vocabulary beyond 11 elements
    (11 RelElement markers); here only the `relation` task is unavailable there."""
    tok = LayoutSequenceTokenizer(LABELS, 32)
    pre = PREPROCESSOR["c"](tokenizer=tok)
    assert pre.N_total == tok.N_total + 7 + 3 + 32 + 6 + 4
    assert pre.name_to_id(RelLoc.UNKNOWN) == tok.N_total + 10 + 32
    with pytest.raises(ValueError):
        PREPROCESSOR["relation"](tokenizer=tok, table={})


def _fake_logits(b, prefix, V, scale):
    """a deterministic stand-in for the decoder: logits of sample b after the tokens `prefix` -- a function of what the KEY / VALUE CACHE holds at
    the positions below the current one, so a cache written by another branch gives other logits (as on the device)"""
    h = int(b) * 1000003
    for t in prefix:
        h = (h * 1000003 + int(t) + 7) & 0x7FFFFFFF
    g = np.random.default_rng(h)
    return torch.from_numpy((g.standard_normal(V) * scale).astype(np.float32))


@pytest.mark.parametrize("scale", [0.25, 1.0, 3.0])
def test_lockstep_exact_mode_consumes_random_like_the_sequential_loop(golden, monkeypatch, scale):
    """the ORDER logic of the exact-mode lock-step decode (models/ralf.py _relation_lockstep_batched, shared=True) on a fake decoder, no GPU:
    range-one draws deferred and replayed in sample order, wide draws parked until their sample is the lowest unfinished one -- the randint
    calls (arguments and results, in order), the tokens and the final state of `random` are those of the sample-after-sample loop.  The scale of
    the fake logits moves the mix: below the gate almost every step back-tracks (range-one draws), above it back-tracks come from the relation
    constraints (wide draws after three failures at one step)."""
    import ralf_amd.models.ralf as R
    import ralf_amd.nn as RN
    from test_model_cpu import build

    g = golden("relation.npz")
    table, _ = load_table()
    random.seed(6)
    model = build(task="relation", relation_table=table).eval()
    tok = LayoutSequenceTokenizer(LABELS, 10)
    batch = make_batch(g.sub("compute_relation"))
    n0 = batch["label"].size(0)
    rep = 6
    batch = {k: (v.repeat((rep,) + (1,) * (v.dim() - 1)) if torch.is_tensor(v) else v * rep) for k, v in batch.items()}
    B = n0 * rep
    batch["id"] = [str(1000 + i % n0) for i in range(B)]
    r = g.sub("sample")
    batch["retrieved"] = [{k: v.repeat((rep,) + (1,) * (v.dim() - 1)) for k, v in dict(r["retrieved"], image=torch.zeros(n0, 16, 4, 1, 1)).items()}]
    random.seed(8)
    torch.manual_seed(8)
    cond, _ = get_condition(batch, "relation", tok)
    V = model.tokenizer.N_total

    class FakeCache:
        def __init__(self, ids):
            self.ids, self.self_kv, self.cross_kv, self.packed = ids, [], [], None
            self.toks = [[0] * 64 for _ in ids]       # the token whose key / value sits at each position, per row

    class FakeStep:
        def __init__(self, m, T, dev, nb):
            self.tok_h, self.pos_h, self.kpm_h = torch.zeros(nb, dtype=torch.long), torch.zeros(nb, dtype=torch.int32), torch.ones(nb, T, dtype=torch.uint8)
            self.cache = None

        def bind(self, cache):
            self.cache = cache

        def __call__(self):   # EVERY row steps, wanted or not (the captured graph has a fixed batch): a row the loop does not feed repeats its last step
            rows = []
            for b in range(self.tok_h.numel()):
                p = int(self.pos_h[b])
                self.cache.toks[b][p] = int(self.tok_h[b])
                rows.append(_fake_logits(b, self.cache.toks[b][:p + 1], V, scale))
            return torch.stack(rows)

    def fake_decoder_step(dec, tok_, pos, cache, rt, kpm):
        cache.toks[0][pos] = int(tok_[0])
        return _fake_logits(cache.ids[0], cache.toks[0][:pos + 1], V, scale)[None]
    monkeypatch.setattr(type(model), "_LockstepStep", FakeStep)
    monkeypatch.setattr(model, "_encode_into_memory", lambda enc: {"memory": torch.arange(B, dtype=torch.float32).view(B, 1, 1).expand(B, 3, 4).contiguous()})
    monkeypatch.setattr(RN, "decoder_init_cache", lambda dec, mem, rt, T: FakeCache(mem[:, 0, 0].long().tolist()))
    monkeypatch.setattr(RN, "decoder_step", fake_decoder_step)
    monkeypatch.setattr(type(model.rt), "to", lambda self, dev: self)
    monkeypatch.setattr(type(model.rt), "begin_step", lambda self: None)
    cfg = {"name": "deterministic", "temperature": 1.0}
    real = random.randint
    runs = {}
    for mode in ("sequential", "lockstep", "serial", "lockstep_forgetful"):
        calls = []
        monkeypatch.setattr(random, "randint", lambda a, b, _c=calls: (_c.append((a, b, real(a, b))), _c[-1][2])[1])
        random.seed(21)
        torch.manual_seed(21)
        model.__dict__.pop("_relation_lockstep_steps", None)
        monkeypatch.setenv("RALF_RELATION_SERIAL", "1" if mode == "serial" else "0")
        monkeypatch.setenv("RALF_RELATION_MEMO_DROP", "37" if mode == "lockstep_forgetful" else "0")
        out = model.sample_relation(cond, sampling_cfg=cfg, return_violation=False, RELATION_SIZE=30, use_graph=False, lockstep=(mode != "sequential"))
        runs[mode] = (out, list(calls), random.getstate())
        if mode != "sequential":
            print(mode, model.relation_stats)
        monkeypatch.setattr(random, "randint", real)
    (o1, c1, s1), (o2, c2, s2) = runs["sequential"], runs["lockstep"]
    wide = sum(1 for a, b, _ in c1 if b > a)
    print(f"scale {scale}: {len(c1)} draws over {B} samples, {wide} wide")
    assert len(c1) > 0
    first = next((i for i, (x, y) in enumerate(zip(c1, c2)) if x != y), None)
    assert first is None and len(c1) == len(c2), (first, c1[max(0, (first or 0) - 3):(first or 0) + 3], c2[max(0, (first or 0) - 3):(first or 0) + 3], len(c1), len(c2))
    assert s1 == s2
    for k in ("label", "mask", "center_x", "center_y", "width", "height"):
        assert torch.equal(o1[k], o2[k]), k
    o3, c3, s3 = runs["serial"]           # the sequential ORDER on the lock-step machinery (the GPU test's yardstick: same arithmetic per row)
    assert c3 == c1 and s3 == s1 and all(torch.equal(o1[k], o3[k]) for k in o1 if torch.is_tensor(o1[k]))
    o4, c4, s4 = runs["lockstep_forgetful"]   # the memo dropped every 37 iterations: prefixes are decoded again on caches other branches have written
    assert c4 == c1 and s4 == s1 and all(torch.equal(o1[k], o4[k]) for k in o1 if torch.is_tensor(o1[k]))
    assert model.relation_stats["refed_positions"] > 0
