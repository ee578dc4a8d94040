"""CPU: host integer path (tokenizer, task conditions, constraint serialisation) vs vectors recorded
from the reference (tests/golden/tokenizer.npz, host_path.npz) -- bit-exact."""
import numpy as np
import pytest
import torch

from ralf_amd.helpers.layout_tokenizer import LayoutSequenceTokenizer
from ralf_amd.helpers.task import get_condition
from ralf_amd.helpers.task_preprocessor import PREPROCESSOR

LABELS = {"pku": ["text", "logo", "underlay"], "cgl": ["logo", "text", "underlay", "embellishment"]}
CFGS = {
    "pku_n10_b128": dict(label_feature=LABELS["pku"], max_seq_length=10, num_bin=128),
    "cgl_n10_b128": dict(label_feature=LABELS["cgl"], max_seq_length=10, num_bin=128),
    "pku_n5_b32_xywh_shared": dict(label_feature=LABELS["pku"], max_seq_length=5, num_bin=32, var_order=["label", "center_x", "center_y", "width", "height"], is_loc_vocab_shared=True),
    "cgl_n11_b16": dict(label_feature=LABELS["cgl"], max_seq_length=11, num_bin=16),
}


@pytest.mark.parametrize("name", list(CFGS))
def test_tokenizer_golden(golden, name):
    g = golden("tokenizer.npz").sub(name)
    tok = LayoutSequenceTokenizer(**CFGS[name])
    m = g["meta"]
    assert (tok.N_total, tok.name_to_id("pad"), tok.name_to_id("bos"), tok.name_to_id("eos")) == tuple(int(m[k]) for k in ("N_total", "pad", "bos", "eos"))
    enc = tok.encode({k: v.clone() for k, v in g["in"].items()})
    assert torch.equal(enc["seq"], g["enc"]["seq"]) and torch.equal(enc["mask"], g["enc"]["mask"])
    for src, ref in ((enc["seq"][:, 1:], g["dec"]), (g["garbage_seq"], g["garbage_dec"])):
        dec = tok.decode(src.clone())
        for k in ("label", "mask", "center_x", "center_y", "width", "height"):
            assert torch.equal(dec[k], ref[k]), k
    if not CFGS[name].get("is_loc_vocab_shared"):
        assert torch.equal(tok.token_mask, g["token_mask"])


def test_tokenizer_roundtrip_properties():
    """the reference's own property tests (tests/train/helpers/test_layout_tokenizer.py:54-116, test_bucketizer.py:25-35)."""
    rng = np.random.default_rng(0)
    for _ in range(50):
        N, nb = int(rng.integers(2, 12)), int(rng.choice([16, 32, 64, 128, 256]))
        order = [["label", "width", "height", "center_x", "center_y"], ["label", "center_x", "center_y", "width", "height"]][int(rng.integers(2))]
        tok = LayoutSequenceTokenizer(LABELS["cgl"], N, num_bin=nb, var_order=order, is_loc_vocab_shared=bool(rng.integers(2)))
        B = 7
        n = torch.from_numpy(rng.integers(1, N + 1, B))
        mask = torch.arange(N).unsqueeze(0) < n.unsqueeze(1)
        lay = {"mask": mask, "label": torch.from_numpy(rng.integers(0, 4, (B, N))) * mask}
        for k in ("center_x", "center_y", "width", "height"):
            lay[k] = torch.from_numpy(rng.random((B, N)).astype(np.float32)) * mask
        enc = tok.encode(lay)
        dec = tok.decode(enc["seq"][:, 1:])
        assert torch.equal(dec["mask"], mask) and torch.equal(dec["label"], lay["label"])
        for k in ("center_x", "center_y", "width", "height"):
            assert ((dec[k] - lay[k]).abs()[mask] <= 0.5 / nb + 1e-6).all()
        enc2 = tok.encode(dec)
        assert torch.equal(enc2["seq"], enc["seq"])  # encode(decode(encode(x))) is idempotent


@pytest.mark.parametrize("task", ["uncond", "c", "cwh", "partial", "refinement"])
def test_condition_and_constraint_sequence(golden, task):
    g = golden("host_path.npz").sub(task)
    tok = LayoutSequenceTokenizer(LABELS["pku"], 10)
    pre = PREPROCESSOR[task](tokenizer=tok, global_task_embedding=False)
    assert pre.N_total == int(golden("host_path.npz").sub("meta")["preproc_N_total"])

    def batch():
        b = {k: v.clone() for k, v in g["batch"].items()}
        b["id"] = [str(1000 + i) for i in range(b["label"].shape[0])]
        r = {k: v.clone() for k, v in g["retrieved"].items()}
        r["image"] = torch.zeros(b["label"].shape[0], 16, 4, 1, 1)
        b["retrieved"] = [r]
        return b

    # training-time path: model.preprocess = get_condition -> preprocessor -> tokenizer.encode
    torch.manual_seed(1234)
    cond, b2 = get_condition(batch(), task, tok)
    seqc = pre(cond)
    data = tok.encode(b2)
    assert torch.equal(seqc["seq"], g["inputs"]["seq_layout_const"]) and torch.equal(seqc["pad_mask"], g["inputs"]["seq_layout_const_pad_mask"])
    assert torch.equal(data["seq"][:, :-1], g["inputs"]["seq"]) and torch.equal(~data["mask"][:, :-1], g["inputs"]["tgt_key_padding_mask"])
    assert torch.equal(data["seq"][:, 1:], g["targets"]["seq"])
    # inference-time path
    torch.manual_seed(4321)
    cond, _ = get_condition(batch(), task, tok)
    if task != "uncond":
        seq_before = cond.seq.clone()
    seqc = pre(cond)
    assert torch.equal(seqc["seq"], g["cond_const"]["seq"]) and torch.equal(seqc["pad_mask"], g["cond_const"]["pad_mask"])
    if task != "uncond":
        assert torch.equal(cond.mask, g["cond"]["mask"])
        assert torch.equal(cond.seq, g["cond"]["seq"]) or torch.equal(seq_before, g["cond"]["seq"])


@pytest.mark.parametrize("cond_type", ["c", "cwh", "refinement", "relation", "uncond"])
def test_forced_token_table_equals_per_step_restriction(cond_type):
    """the [T, B] forced-token table the decode loop indexes == the per-step restriction (restrict_* keep exactly one token
    where a token is forced and every token elsewhere)"""
    from ralf_amd.helpers.sampling import DECODE_SPACE_RESTRICTION, NEG_INF, forced_tokens, forced_tokens_all

    g = torch.Generator().manual_seed(3)
    B, T, V, pad, eos = 7, 50, 140, 139, 138
    cond = torch.randint(0, 137, (B, T + 1), generator=g)
    cond[torch.rand(B, T + 1, generator=g) < 0.3] = -1      # free slots
    for b, n in enumerate([0, 1, 6, 11, 26, 50, 51]):       # ragged lengths incl. empty and full
        cond[b, n:] = pad
    table = forced_tokens_all(cond, cond_type, pad, eos, T)
    if cond_type == "uncond":
        assert table is None
        return
    assert table.shape == (T, B) and table.is_contiguous()
    for step in range(1, T + 1):
        one = forced_tokens(step, cond, cond_type, pad, eos, T)
        row = table[step - 1]
        assert torch.equal(row, one if one is not None else torch.full_like(row, -1))
        kept = DECODE_SPACE_RESTRICTION[cond_type](step, cond, torch.zeros(B, V), pad, eos, T) != NEG_INF
        want = torch.ones(B, V, dtype=torch.bool)
        for b in range(B):
            if row[b] >= 0:
                want[b] = False
                want[b, row[b]] = True
        assert torch.equal(kept, want)
