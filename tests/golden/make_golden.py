"""Generate golden fixtures by RUNNING THE REFERENCE in this container (read-only /root/reference,
stub-imported per SURVEY.md Appendix A).  Commit the produced *.npz / *.json; this script cannot
run on the GPU box (the reference does not travel) and nothing in tests/, smoke() or bench.py
imports it.

    python tests/golden/make_golden.py

Weights are the deterministic name-keyed tensors of oracle/detweights.py loaded into the
reference modules with load_state_dict(strict=True); fixtures hold only inputs and the
reference's outputs.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import _ref_import  # noqa: E402

DictConfig = _ref_import.install()

import datasets as ds  # noqa: E402
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

import image2layout.train.models.autoreg as ar  # noqa: E402
import image2layout.train.models.retrieval_augmented_autoreg as raa  # noqa: E402
from image2layout.train.fid.model import FIDNetV3  # noqa: E402
from image2layout.train.helpers.layout_tokenizer import LayoutSequenceTokenizer  # noqa: E402
from image2layout.train.helpers.task import get_condition  # noqa: E402
from image2layout.train.models.common.attention import Attention, FeedForward  # noqa: E402
from image2layout.train.models.common.positional_encoding import (  # noqa: E402
    PositionalEncoding1d,
    PositionEmbeddingSine,
)
from oracle.detweights import det_state_dict  # noqa: E402

torch.set_num_threads(8)
META = {"torch": torch.__version__, "reference": "CyberAgentAILab/RALF @ 2024_08_07"}


THIN = 37  # big weight-gradient tensors are stored as flatten()[::THIN] to keep fixtures small


def thin(v):
    return v.flatten()[::THIN] if v.numel() > 20000 else v


def npify(d, prefix=""):
    out = {}
    for k, v in d.items():
        if isinstance(v, dict):
            out.update(npify(v, prefix + k + "/"))
        elif torch.is_tensor(v):
            if k.startswith("g_") or prefix.endswith("grads/"):
                v = thin(v)
            out[prefix + k] = v.detach().cpu().numpy()
        elif isinstance(v, np.ndarray):
            out[prefix + k] = v
    return out


def save(name, d):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **npify(d))
    print("wrote", name, os.path.getsize(path) // 1024, "KiB")


# ----------------------------------------------------------------------------------------
class StandInBackbone(nn.Module):
    """Parameter-free stand-in for ResnetFeatureExtractor (timm/torchvision absent): returns a
    tensor injected by the fixture so the fixtures are backbone-independent."""

    feat = None

    def __init__(self, **kw):
        super().__init__()

    def forward(self, img):
        return StandInBackbone.feat


def fidnet_no_ckpt(dataset_name="", num_classes=3, d_model=256, nhead=4, num_layers=4, max_seq_length=10, ckpt_dir=""):
    m = FIDNetV3(num_label=num_classes, d_model=d_model, nhead=nhead, num_layers=num_layers, max_bbox=max_seq_length)
    m.eval()
    for n in ("pos_token", "dec_transformer", "fc_out_disc", "fc_out_cls", "fc_out_bbox"):
        delattr(m, n)
    return m


raa.ResnetFeatureExtractor = StandInBackbone
ar.ResnetFeatureExtractor = StandInBackbone
raa.load_fidnet_feature_extractor = fidnet_no_ckpt

LABELS = {"pku": ["text", "logo", "underlay"], "cgl": ["logo", "text", "underlay", "embellishment"]}


def make_tokenizer(dataset="pku", N=10, num_bin=128, var_order=None, shared=False):
    return LayoutSequenceTokenizer(
        label_feature=ds.ClassLabel(names=LABELS[dataset]),
        max_seq_length=N,
        num_bin=num_bin,
        var_order=var_order or ["label", "width", "height", "center_x", "center_y"],
        pad_until_max=False,
        special_tokens=["pad", "bos", "eos"],
        is_loc_vocab_shared=shared,
        geo_quantization="linear",
    )


def synth_layouts(g, lead, N, C, full_first=False):
    """random layouts: n~U{1..N} (first sample full, second single-element)."""
    shape = tuple(lead) + (N,)
    n = torch.randint(1, N + 1, tuple(lead), generator=g)
    flat_n = n.view(-1)
    if full_first:
        flat_n[0] = N
        if flat_n.numel() > 1:
            flat_n[1] = 1
    mask = torch.arange(N).expand(shape) < n.unsqueeze(-1)
    out = {"mask": mask, "label": torch.randint(0, C, shape, generator=g) * mask}
    for k in ("center_x", "center_y", "width", "height"):
        out[k] = torch.rand(shape, generator=g) * mask
    return out


def synth_batch(seed, B, N, K, C, hw=8):
    g = torch.Generator().manual_seed(seed)
    b = synth_layouts(g, (B,), N, C, full_first=True)
    b["image"] = torch.rand(B, 3, hw, hw, generator=g)
    b["saliency"] = torch.rand(B, 1, hw, hw, generator=g)
    b["id"] = [str(1000 + i) for i in range(B)]
    r = synth_layouts(g, (B, K), N, C)
    r["image"] = torch.rand(B, K, 3, hw, hw, generator=g)
    r["saliency"] = torch.rand(B, K, 1, hw, hw, generator=g)
    b["retrieved"] = [r]
    return b


def clone_batch(b):
    out = {}
    for k, v in b.items():
        if torch.is_tensor(v):
            out[k] = v.clone()
        elif isinstance(v, list) and v and isinstance(v[0], dict):
            out[k] = [{kk: vv.clone() for kk, vv in v[0].items()}]
        else:
            out[k] = list(v)
    return out


# ----------------------------------------------------------------------------------------
def golden_tokenizer():
    out = {}
    cfgs = [
        ("pku_n10_b128", dict(dataset="pku", N=10, num_bin=128)),
        ("cgl_n10_b128", dict(dataset="cgl", N=10, num_bin=128)),
        ("pku_n5_b32_xywh_shared", dict(dataset="pku", N=5, num_bin=32, var_order=["label", "center_x", "center_y", "width", "height"], shared=True)),
        ("cgl_n11_b16", dict(dataset="cgl", N=11, num_bin=16)),
    ]
    for name, kw in cfgs:
        tok = make_tokenizer(**kw)
        N, nb = kw["N"], kw["num_bin"]
        C = len(LABELS[kw["dataset"]])
        g = torch.Generator().manual_seed(7)
        lay = synth_layouts(g, (12,), N, C, full_first=True)
        # edge cases: values exactly on bin boundaries, out of [0,1], an empty layout
        lay["center_x"][0, :N] = (torch.arange(N) % (nb + 1)).float() / nb
        lay["width"][0, :N] = torch.tensor([-0.5, 1.5, 0.0, 1.0, 0.999999] * 3)[:N]
        lay["height"][2] = lay["height"][2] * 0 + 1.0 / nb * torch.arange(N) * lay["mask"][2]
        lay["mask"][3] = False
        for k in ("label", "center_x", "center_y", "width", "height"):
            lay[k][3] = 0
        enc = tok.encode({k: v.clone() for k, v in lay.items()})
        dec = tok.decode(enc["seq"][:, 1:].clone())
        # decode of garbage (oov / eos in the middle)
        gseq = torch.randint(0, tok.N_total, (6, 5 * N), generator=g)
        gdec = tok.decode(gseq.clone())
        out[name] = {
            "in": lay, "enc": enc, "dec": dec, "garbage_seq": gseq, "garbage_dec": gdec,
            # (token_mask raises for is_loc_vocab_shared=True in the reference: mismatched stack sizes)
            "token_mask": tok.token_mask if not kw.get("shared") else torch.zeros(0),
            "meta": {"N_total": torch.tensor(tok.N_total), "pad": torch.tensor(tok.name_to_id("pad")),
                     "bos": torch.tensor(tok.name_to_id("bos")), "eos": torch.tensor(tok.name_to_id("eos"))},
        }
    save("tokenizer.npz", out)


def golden_host_path():
    """get_condition + task preprocessor + model.preprocess (a12) for every task without external tables."""
    tok = make_tokenizer("pku", 10)
    out = {}
    for task in ["uncond", "c", "cwh", "partial", "refinement"]:
        model = build_ralf(tok, task)
        batch = synth_batch(11, 6, 10, 16, 3)
        out[task] = {"batch": {k: v for k, v in batch.items() if torch.is_tensor(v)},
                     "retrieved": {k: v for k, v in batch["retrieved"][0].items() if k not in ("image", "saliency")}}
        torch.manual_seed(1234)
        inputs, targets = model.preprocess(clone_batch(batch))
        out[task]["inputs"] = {k: v for k, v in inputs.items() if torch.is_tensor(v) and k != "image"}
        out[task]["targets"] = targets
        # test-time condition (inference.py:389) + constraint sequence
        torch.manual_seed(4321)
        cond, _ = get_condition(clone_batch(batch), task, tok)
        seqc = model.preprocessor(cond)
        out[task]["cond"] = {"seq": cond.seq if cond.seq is not None else torch.zeros(0), "mask": cond.mask if cond.mask is not None else torch.zeros(0)}
        out[task]["cond_const"] = seqc
    out["meta"] = {"preproc_N_total": torch.tensor(model.preprocessor.N_total)}
    save("host_path.npz", out)


# ----------------------------------------------------------------------------------------
def load_det(module, prefix=""):
    sd = module.state_dict()
    det = det_state_dict({prefix + k: tuple(v.shape) for k, v in sd.items()})
    module.load_state_dict({k: det[prefix + k] for k in sd}, strict=True)
    return {prefix + k: tuple(v.shape) for k, v in sd.items()}


def golden_modules():
    out = {}
    g = torch.Generator().manual_seed(3)
    # a6 cross-attention fuse
    attn = Attention(256, 256, heads=8, dim_head=64, dropout=0.0).eval()
    load_det(attn, "attn.")
    x = torch.randn(2, 15, 256, generator=g, requires_grad=True)
    ctx = torch.randn(2, 16, 256, generator=g, requires_grad=True)
    y = attn(x, ctx)
    go = torch.randn(y.shape, generator=g)
    y.backward(go)
    out["xattn"] = {"x": x, "ctx": ctx, "y": y, "go": go, "gx": x.grad, "gctx": ctx.grad,
                    "g_to_q": attn.to_q.weight.grad, "g_to_kv": attn.to_kv.weight.grad, "g_norm_w": attn.norm.weight.grad}
    # FeedForward
    ff = FeedForward(256, 1024, dropout=0.0).eval()
    load_det(ff, "head.")
    x = torch.randn(2, 7, 256, generator=g, requires_grad=True)
    y = ff(x)
    go = torch.randn(y.shape, generator=g)
    y.backward(go)
    out["ff"] = {"x": x, "y": y, "go": go, "gx": x.grad, "g_w1": ff.net[1].weight.grad, "g_b2": ff.net[4].bias.grad}
    # pre-norm encoder layer with key padding
    enc = nn.TransformerEncoderLayer(256, 8, 1024, 0.1, batch_first=True, norm_first=True).eval()
    load_det(enc, "transformer_encoder.layers.0.")
    x = torch.randn(3, 9, 256, generator=g, requires_grad=True)
    kpm = torch.tensor([[False] * 9, [False] * 5 + [True] * 4, [False] * 8 + [True]])
    y = enc(x, src_key_padding_mask=kpm)
    go = torch.randn(y.shape, generator=g)
    y.backward(go)
    out["enc_layer"] = {"x": x, "kpm": kpm, "y": y, "go": go, "gx": x.grad,
                        "g_in_proj_w": enc.self_attn.in_proj_weight.grad, "g_in_proj_b": enc.self_attn.in_proj_bias.grad,
                        "g_lin2_w": enc.linear2.weight.grad, "g_norm1_b": enc.norm1.bias.grad}
    # pre-norm decoder layer, causal + padding
    dec = nn.TransformerDecoderLayer(256, 8, 1024, batch_first=True, norm_first=True).eval()
    load_det(dec, "decoder.transformer.layers.0.")
    x = torch.randn(3, 10, 256, generator=g, requires_grad=True)
    mem = torch.randn(3, 21, 256, generator=g, requires_grad=True)
    kpm = torch.tensor([[False] * 10, [False] * 6 + [True] * 4, [False] * 1 + [True] * 9])
    cm = nn.Transformer.generate_square_subsequent_mask(10)
    y = dec(x, mem, tgt_mask=cm, tgt_key_padding_mask=kpm)
    go = torch.randn(y.shape, generator=g)
    y.backward(go)
    out["dec_layer"] = {"x": x, "mem": mem, "kpm": kpm, "y": y, "go": go, "gx": x.grad, "gmem": mem.grad,
                        "g_cross_in_w": dec.multihead_attn.in_proj_weight.grad, "g_self_out_w": dec.self_attn.out_proj.weight.grad}
    # frozen layout encoder
    fid = fidnet_no_ckpt(num_classes=3, max_seq_length=10)
    load_det(fid, "layout_encoer.")
    lay = synth_layouts(g, (5,), 10, 3, full_first=True)
    with torch.no_grad():
        f = fid.extract_features({k: v for k, v in lay.items()})
    out["fidnet"] = {"in": lay, "feat": f}
    # positional encodings
    pe2 = PositionEmbeddingSine(d_model=256, normalize=True)
    z = torch.zeros(1, 256, 3, 5)
    out["pos2d_3x5"] = {"table": pe2(z)[0]}
    out["pos2d_16x16"] = {"table": pe2(torch.zeros(1, 256, 16, 16))[0]}
    pe1 = PositionalEncoding1d(d_model=256).eval()
    xx = torch.randn(2, 6, 256, generator=g)
    out["pe1d"] = {"x": xx, "y": pe1(xx), "pe_head": pe1.pe[0, :64]}
    save("modules.npz", out)


# ----------------------------------------------------------------------------------------
def build_ralf(tok, task, top_k=16):
    feats = ds.Features({"label": ds.Sequence(ds.ClassLabel(names=tok._label_feature.names))})
    m = raa.ConcateAuxilaryTaskConcateCrossAttnRetrievalAugmentedAutoreg(
        features=feats, tokenizer=tok, dataset_name="pku", max_seq_length=tok.max_seq_length, db_dataset=None,
        top_k=top_k, retrieval_backbone="dreamsim", random_retrieval=False, saliency_k="None", auxilary_task=task)
    return m


def build_autoreg(tok, task):
    feats = ds.Features({"label": ds.Sequence(ds.ClassLabel(names=tok._label_feature.names))})
    return ar.ConcateAuxilaryTaskAutoreg(features=feats, tokenizer=tok, auxilary_task=task)


GRAD_KEYS_RALF = [
    "decoder.head.1.weight", "decoder.emb.weight", "decoder.transformer.layers.5.multihead_attn.in_proj_weight",
    "decoder.transformer.layers.0.self_attn.in_proj_bias", "transformer_encoder.layers.0.linear1.weight",
    "transformer_encoder.layers.5.norm2.weight", "attn.to_kv.weight", "attn.to_out.0.bias", "head.net.1.weight",
    "layout_adapter.net.4.weight", "layout_adapter.net.0.bias", "user_const_encoder.emb.weight",
    "user_const_encoder.encoder.layers.3.self_attn.out_proj.weight", "task_emb.weight",
]


def golden_e2e():
    tok = make_tokenizer("pku", 10)
    out = {}
    shapes_written = False
    for task, hw in [("uncond", (3, 5)), ("refinement", (4, 4)), ("c", (2, 3))]:
        model = build_ralf(tok, task).eval()  # eval: dropout off (parity mode)
        shapes = load_det(model)
        if not shapes_written:
            with open(os.path.join(HERE, "ralf_state_shapes.json"), "w") as f:
                json.dump({"meta": META, "shapes": {k: list(v) for k, v in shapes.items()}}, f, indent=0)
            shapes_written = True
        B = 3
        batch = synth_batch(21, B, 10, 16, 3)
        torch.manual_seed(99)
        inputs, targets = model.preprocess(clone_batch(batch))
        g = torch.Generator().manual_seed(5)
        feat = torch.randn(B, 256, *hw, generator=g).requires_grad_(True)
        StandInBackbone.feat = feat
        model.zero_grad()
        outputs, losses = model.train_loss(inputs, targets)
        losses["nll_loss"].backward()
        named = dict(model.named_parameters())
        rec = {
            "feat": feat, "gfeat": feat.grad, "logits": outputs["logits"], "loss": losses["nll_loss"],
            "inputs": {k: v for k, v in inputs.items() if torch.is_tensor(v) and k != "image"},
            "retrieved": {k: v for k, v in inputs["retrieved"].items() if k not in ("image", "saliency")},
            "targets": targets,
            "grads": {k: named[k].grad for k in GRAD_KEYS_RALF},
            "gradnorm": torch.sqrt(sum((p.grad ** 2).sum() for p in model.parameters() if p.grad is not None)),
        }
        out["ralf_" + task] = rec
    # Autoreg baseline (BASELINE config 1)
    model = build_autoreg(tok, "uncond").eval()
    shapes = load_det(model)
    with open(os.path.join(HERE, "autoreg_state_shapes.json"), "w") as f:
        json.dump({"meta": META, "shapes": {k: list(v) for k, v in shapes.items()}}, f, indent=0)
    batch = synth_batch(22, 4, 10, 16, 3)
    batch.pop("retrieved")
    inputs, targets = model.preprocess(clone_batch(batch))
    g = torch.Generator().manual_seed(6)
    feat = torch.randn(4, 256, 3, 4, generator=g).requires_grad_(True)
    StandInBackbone.feat = feat
    outputs, losses = model.train_loss(inputs, targets)
    losses["nll_loss"].backward()
    named = dict(model.named_parameters())
    out["autoreg_uncond"] = {
        "feat": feat, "gfeat": feat.grad, "logits": outputs["logits"], "loss": losses["nll_loss"],
        "inputs": {k: v for k, v in inputs.items() if torch.is_tensor(v) and k != "image"}, "targets": targets,
        "grads": {k: named[k].grad for k in ["decoder.head.1.weight", "transformer_encoder.layers.2.self_attn.in_proj_weight", "task_emb.weight"]},
    }
    save("e2e.npz", out)


def golden_sample():
    """deterministic (argmax) sample() tokens -> decoded layouts, tasks without external tables."""
    tok = make_tokenizer("pku", 10)
    out = {}
    cfg = DictConfig(name="deterministic")
    for task in ["uncond", "c", "cwh", "refinement", "partial"]:
        model = build_ralf(tok, task).eval()
        load_det(model)
        B = 3
        batch = synth_batch(31, B, 10, 16, 3)
        torch.manual_seed(77)
        cond, _ = get_condition(clone_batch(batch), task, tok)
        g = torch.Generator().manual_seed(8)
        feat = torch.randn(B, 256, 2, 3, generator=g)
        StandInBackbone.feat = feat
        torch.manual_seed(78)
        with torch.no_grad():
            enc_in, seq_constraints = model._create_encoder_inputs(cond)
            # re-run via public API (it rebuilds encoder inputs itself; reseed for `partial` shuffling)
            torch.manual_seed(78)
            res, vio = model.sample(cond=cond, sampling_cfg=cfg, cond_type=task, return_violation=True, use_backtrack=False)
        out[task] = {
            "feat": feat,
            "cond_seq": cond.seq if cond.seq is not None else torch.zeros(0),
            "seq_layout_const": enc_in["seq_layout_const"], "seq_layout_const_pad_mask": enc_in["seq_layout_const_pad_mask"],
            "retrieved": {k: v for k, v in cond.retrieved.items() if k not in ("image", "saliency")},
            "result": res,
            "violation": {"total": torch.tensor(float(vio["total"])), "viorated": torch.tensor(float(vio["viorated"]))},
        }
    save("sample.npz", out)


def golden_reranker():
    """MMR / top-k re-ranking known answers from the reference implementation (pure NumPy)."""
    from image2layout.train.models.retrieval import reranker as RR

    rng = np.random.default_rng(5)
    out = {}
    for i, (n, k, lam, st) in enumerate([(64, 16, 0.5, "similarity"), (128, 16, 0.9, "similarity"), (40, 8, 0.3, "distance"), (20, 20, 1.0, "similarity"), (33, 5, 0.0, "distance")]):
        q = rng.standard_normal(n)
        f = rng.standard_normal((n, 12))
        f /= np.linalg.norm(f, axis=1, keepdims=True)
        pair = f @ f.T if st == "similarity" else np.linalg.norm(f[:, None] - f[None], axis=-1)
        out[f"case{i}"] = {"q": q, "pair": pair, "mmr": RR.maximal_marginal_relevance(q, pair, lam, k, st), "topk": RR.reranker_top_k(q, k, st),
                           "cfg": np.array([n, k, lam, 1.0 if st == "similarity" else 0.0])}
    save("reranker.npz", out)



def _enc_entry(e):
    """relationship-table entry -> JSON-able [label, elem value, relation value, label | "canvas", elem value | "pad"]"""
    return [e[0], int(e[1]), int(e[2]), e[3], (e[4] if isinstance(e[4], str) else int(e[4]))]


def _enc_constraints(rel):
    out = []
    for lst in rel:
        out.append([[("canvas", int(b)) if a == "canvas" else (int(a), int(b)) for a, b in lst]])
    return [[list(x) for x in l[0]] for l in out]


def golden_relation():
    """`relation` task (needs a relationship table: synthesised with the reference's own rules and written where the
    reference looks for it, in a scratch cwd): compute_relation, RelationshipPreprocessor, the per-step constraint
    masks of TransformerSortByDictRelationConstraint, and end-to-end sample_relation tokens (deterministic sampling)."""
    import random
    import tempfile

    from image2layout.train.helpers import relationships as RS
    from image2layout.train.models.layoutformerpp.relation_restriction import TransformerSortByDictRelationConstraint

    tok = make_tokenizer("pku", 10)
    names = LABELS["pku"]
    out, js = {}, {}
    B = 4
    batch = synth_batch(41, B, 10, 16, 3)
    batch["mask"][2, 1:] = False   # a single-element layout: only a canvas relation
    for k in ("label", "center_x", "center_y", "width", "height"):
        batch[k][2, 1:] = 0
    # --- compute_relation (helpers/relationships.py:112-166) ---
    random.seed(5)
    cr = RS.compute_relation({k: v for k, v in batch.items() if torch.is_tensor(v)}, edge_ratio=0.5)
    out["compute_relation"] = {"batch": {k: batch[k] for k in ("label", "mask", "center_x", "center_y", "width", "height")}, **cr}
    # --- table with the rules of preprocess/precompute_relationship.py:57-126 (restated here with the reference's detectors/enums) ---
    def build_table(batch):
        table = {}
        for b in range(batch["label"].size(0)):
            seen, uniq = {}, []
            for lab, m in zip(batch["label"][b].tolist(), batch["mask"][b].tolist()):
                if not m:
                    uniq.append(None)
                    continue
                seen[lab] = seen.get(lab, 0) + 1
                uniq.append([names[lab], list(RS.RelElement)[seen[lab] - 1]])
            valid = [i for i, m in enumerate(batch["mask"][b].tolist()) if m][::-1]
            box = lambda i: [batch[k][b, i].item() for k in ("center_x", "center_y", "width", "height")]  # noqa: E731
            loc, size, canvas = [], [], []
            for pos, i in enumerate(valid):
                for j in valid[pos + 1:]:
                    loc.append([*uniq[i], RS.detect_loc_relation_between_elements(box(i), box(j)), *uniq[j]])
                    size.append([*uniq[i], RS.detect_size_relation(box(i), box(j)), *uniq[j]])
                canvas.append([*uniq[i], RS.detect_loc_relation_between_element_and_canvas(box(i)), "canvas", "pad"])
            table[batch["id"][b]] = loc + size + canvas
        return table

    def drive(fn, seqs, n_iter, seed):
        """walk the constraint function with random admissible tokens, cuts on infeasible steps and spontaneous cuts"""
        rng = np.random.default_rng(seed)
        tm = tok.token_mask
        steps, cons = {}, {}
        for b in range(seqs.size(0)):
            rel = fn.prepare(seqs[b])
            cons[str(b)] = _enc_constraints(rel)
            seq = torch.full((1, 1), tok.name_to_id("bos"))
            rec_mask, rec_back, rec_tok, rec_len = [], [], [], []
            for it in range(n_iter):
                mask, back = fn(seq, rel)
                n_dec = seq.size(1) - 1
                rec_len.append(n_dec)
                rec_mask.append(mask.clone())
                rec_back.append(-1 if back is None else int(back))
                ok = (~mask) & tm[n_dec]
                if int(ok.sum()) == 0:             # infeasible: back-track like sample_relation (or cut two tokens)
                    cut = back if (back is not None and rng.random() < 0.7) else max(2, seq.size(1) - 2)
                    seq = seq[:, :cut]
                    rec_tok.append(-1)
                    continue
                cand = ok.nonzero().flatten()
                t = int(cand[int(rng.integers(0, len(cand)))])
                rec_tok.append(t)
                seq = torch.cat([seq, torch.tensor([[t]])], dim=1)
                if t == tok.name_to_id("eos") or seq.size(1) == tok.max_token_length + 1:
                    break
                if rng.random() < 0.04 and seq.size(1) > 4:   # spontaneous cut: exercises the history slicing
                    seq = seq[:, : int(rng.integers(2, seq.size(1)))]
            steps[f"s{b}"] = {"mask": torch.stack(rec_mask), "back": torch.tensor(rec_back), "token": torch.tensor(rec_tok), "n_decoded": torch.tensor(rec_len)}
        return steps, cons

    table = build_table(batch)
    js["table"] = {k: [_enc_entry(e) for e in v] for k, v in table.items()}
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "cache"))
    torch.save(table, os.path.join(tmp, "cache", "pku_cgl_relationships_dic_using_canvas_sort_label_lexico.pt"))
    os.chdir(tmp)
    torch.serialization.add_safe_globals([RS.RelElement, RS.RelLoc, RS.RelSize])   # torch >= 2.6 loads with weights_only=True
    try:
        # --- preprocessor + constraint masks ---
        random.seed(6)
        torch.manual_seed(6)
        model = build_ralf(tok, "relation").eval()      # its ctor builds the RelationshipPreprocessor (shuffles the table: random)
        load_det(model)
        random.seed(7)
        torch.manual_seed(7)
        cond, _ = get_condition(clone_batch(batch), "relation", tok)
        model.preprocessor.set_relation_size(30)
        seqc = model.preprocessor(cond)
        out["preprocessor"] = {"cond_seq": cond.seq, "cond_mask": cond.mask, "edge_indexes": cond.edge_indexes, "edge_attributes": cond.edge_attributes,
                               "seq": seqc["seq"], "pad_mask": seqc["pad_mask"]}
        fn = TransformerSortByDictRelationConstraint(model.preprocessor)
        out["steps"], js["constraints"] = drive(fn, seqc["seq"], 140, 9)
        # --- scenario 2: dense layouts, 60 % of the relations (every relation kind on every slot) ---
        from image2layout.train.models.layoutformerpp.task_preprocessor import RelationshipPreprocessor
        batch2 = synth_batch(43, 6, 10, 16, 3)
        for b2 in range(6):                         # at least 6 elements everywhere
            n = max(int(batch2["mask"][b2].sum()), 6 + b2 % 4)
            g2 = torch.Generator().manual_seed(100 + b2)
            batch2["mask"][b2] = torch.arange(10) < n
            batch2["label"][b2] = torch.randint(0, 3, (10,), generator=g2) * batch2["mask"][b2]
            for k in ("center_x", "center_y"):
                batch2[k][b2] = (0.1 + 0.8 * torch.rand(10, generator=g2)) * batch2["mask"][b2]
            for k in ("width", "height"):
                batch2[k][b2] = (0.05 + 0.3 * torch.rand(10, generator=g2)) * batch2["mask"][b2]
        table2 = build_table(batch2)
        js["table2"] = {k: [_enc_entry(e) for e in v] for k, v in table2.items()}
        torch.save(table2, os.path.join(tmp, "cache", "pku_cgl_relationships_dic_using_canvas_sort_label_lexico.pt"))
        random.seed(16)
        pre2 = RelationshipPreprocessor(tokenizer=tok, global_task_embedding=False)
        random.seed(17)
        torch.manual_seed(17)
        condb, _ = get_condition(clone_batch(batch2), "relation", tok)
        pre2.set_relation_size(60)
        seqc2 = pre2(condb)
        out["preprocessor2"] = {"batch": {k: batch2[k] for k in ("label", "mask", "center_x", "center_y", "width", "height")},
                                "cond_seq": condb.seq, "seq": seqc2["seq"], "pad_mask": seqc2["pad_mask"]}
        out["steps2"], js["constraints2"] = drive(TransformerSortByDictRelationConstraint(pre2), seqc2["seq"], 220, 19)
        # --- end-to-end sample_relation (deterministic = argmax; back-tracking uses Python's random) ---
        g = torch.Generator().manual_seed(12)
        feat = torch.randn(B, 256, 2, 3, generator=g)
        StandInBackbone.feat = feat
        random.seed(8)
        torch.manual_seed(8)
        cond2, _ = get_condition(clone_batch(batch), "relation", tok)
        random.seed(10)
        torch.manual_seed(10)
        res, vio = model.sample(cond=cond2, sampling_cfg=DictConfig(name="deterministic", temperature=1.0), cond_type="relation",
                                return_violation=True, use_backtrack=True, RELATION_SIZE=30)
        out["sample"] = {"feat": feat, "cond_seq": cond2.seq, "cond_mask": cond2.mask,
                         "retrieved": {k: v for k, v in cond2.retrieved.items() if k not in ("image", "saliency")},
                         "result": res, "violation": {"total": torch.tensor(float(vio["total"])), "viorated": torch.tensor(float(vio["viorated"]))}}
    finally:
        os.chdir(cwd)
    with open(os.path.join(HERE, "relation.json"), "w") as f:
        json.dump(js, f)
    save("relation.npz", out)


def golden_retrieval_augment():
    """the RetrievalAugmentation block shared by the *_ra baselines (models/common/retrieval_augment.py:18-101)"""
    import image2layout.train.models.common.retrieval_augment as rag

    rag.load_fidnet_feature_extractor = fidnet_no_ckpt
    m = rag.RetrievalAugmentation(d_model=256, dataset_name="pku", top_k=16, num_classes=3, max_seq_length=10, use_reference_image=False).eval()
    shapes = load_det(m)
    with open(os.path.join(HERE, "retrieval_augment_state_shapes.json"), "w") as f:
        json.dump({"meta": META, "shapes": {k: list(v) for k, v in shapes.items()}}, f, indent=0)
    B = 3
    batch = synth_batch(31, B, 10, 16, 3)
    retrieved = m.preprocess_retrieved_samples(batch["retrieved"])
    g = torch.Generator().manual_seed(8)
    feat = torch.randn(B, 12, 256, generator=g).requires_grad_(True)
    w = torch.randn(B, 12 + 12 + 16, 256, generator=g)
    memory = m(None, feat, retrieved)
    (memory * w).sum().backward()
    named = dict(m.named_parameters())
    save("retrieval_augment.npz", {
        "feat": feat, "w": w, "memory": memory, "gfeat": feat.grad,
        "retrieved": {k: v for k, v in retrieved.items() if k not in ("image", "saliency")},
        "grads": {k: named[k].grad for k in ["attn.to_kv.weight", "attn.to_out.0.bias", "head.net.1.weight", "layout_adapter.net.4.weight", "layout_adapter.net.0.bias"]},
    })


def golden_backbone_wrapper():
    """a1 wrapper: the reference's OWN ResnetBackbone constructor and forward (common/image.py:27-120) around a stand-in
    body -- timm / torchvision are absent here, so `timm.create_model` hands the constructor a module with a 3-channel 7x7
    `conv1` (the only attribute the wrapper touches) whose forward returns injected {"layer3", "layer4"} maps.  Recorded:
    the 4-channel stem weight the constructor builds from the 3-channel one (image.py:70-77), its output on an image, and the
    FPN fuse + projection (image.py:99-111) with gradients, at 16x16 / 8x8 (256x256 canvas) and 22x15 / 11x8 (350x240)."""
    import io

    import image2layout.train.models.common.image as im

    class Body(nn.Module):
        feats = None

        def __init__(self):
            super().__init__()
            self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)

        def forward(self, img):
            return Body.feats

    class FakeFS:
        def exists(self, p):
            return True

        def open(self, p, mode="rb"):
            return io.BytesIO(b"")

    body = Body()
    w3 = det_state_dict({"encoder.extractor.body.conv1.weight3": (64, 3, 7, 7)})["encoder.extractor.body.conv1.weight3"]
    body.conv1.weight.data = w3.clone()
    saved = (im.timm.create_model, im.fsspec.core.url_to_fs, im.torch.load, im.create_feature_extractor)
    im.timm.create_model = lambda name: body
    im.fsspec.core.url_to_fs = lambda path: (FakeFS(), path)
    im.torch.load = lambda f: body.state_dict()
    im.create_feature_extractor = lambda model, return_nodes: model
    try:
        bb = im.ResnetBackbone(backbone="resnet50", d_model=256, head="transformer")
    finally:
        im.timm.create_model, im.fsspec.core.url_to_fs, im.torch.load, im.create_feature_extractor = saved
    out = {"stem": {"w3": w3, "w4": bb.body.conv1.weight.data.clone()}}
    g = torch.Generator().manual_seed(17)
    img = torch.rand(2, 4, 32, 40, generator=g)
    out["stem"]["img"] = img
    out["stem"]["y"] = bb.body.conv1(img)
    fpn = {k: v for k, v in bb.state_dict().items() if k.startswith(("fpn_", "proj."))}
    det = det_state_dict({"encoder.extractor." + k: tuple(v.shape) for k, v in fpn.items()})
    bb.load_state_dict({**bb.state_dict(), **{k: det["encoder.extractor." + k] for k in fpn}}, strict=True)
    # the injected maps are regenerated from seeds by the tests (torch's CPU generator is deterministic for a fixed build,
    # the same convention as oracle/detweights.py), so the fixture holds only the reference's outputs
    for name, (h3, w3_, h4, w4_, seed) in {"c256": (16, 16, 8, 8, 171), "c350x240": (22, 15, 11, 8, 172)}.items():
        gg = torch.Generator().manual_seed(seed)
        f3 = torch.randn(1, 1024, h3, w3_, generator=gg).requires_grad_(True)
        f4 = torch.randn(1, 2048, h4, w4_, generator=gg).requires_grad_(True)
        go = torch.randn(1, 256, h3, w3_, generator=gg)
        Body.feats = {"layer3": f3, "layer4": f4}
        bb.zero_grad()
        y = bb(torch.zeros(1, 4, 8, 8))
        y.backward(go)
        named = dict(bb.named_parameters())
        out[name] = {"seed": torch.tensor(seed), "y": y, "g_layer3": thin(f3.grad), "g_layer4": thin(f4.grad),
                     "grads": {k: named[k].grad for k in ["fpn_conv11_4.weight", "fpn_conv11_5.bias", "fpn_conv33.weight", "proj.weight", "proj.bias"]}}
    save("backbone_wrapper.npz", out)


def golden_resnet_body_hf():
    """independent check of the ResNet-50 BODY (timm ^0.9.5 / torchvision are absent, so the body cannot be run from the reference):
    transformers' ResNetModel with layer_type="bottleneck", depths [3, 4, 6, 3], downsample_in_bottleneck=False is the same published
    architecture (ResNet-v1.5: stride on the 3x3) written by other people.  The deterministic body weights of oracle/detweights.py are
    mapped into it key by key; recorded: the layer3 / layer4 maps (the two taps of common/image.py:66-67) and gradients, with running
    statistics (eval) and with batch statistics (train), on a 96 x 128 canvas."""
    from oracle.detweights import resnet50_fpn_shapes

    # (the stub modules that stand in for the reference's absent imports have no __spec__, which transformers' availability probe trips over)
    stubs = {k: sys.modules.pop(k) for k in list(sys.modules) if k.split(".")[0] in ("torchvision", "timm")}
    try:
        from transformers import ResNetConfig, ResNetModel
        ResNetModel(ResNetConfig(depths=[1, 1, 1, 1]))   # resolve the lazy imports now
    finally:
        sys.modules.update(stubs)
    cfg = ResNetConfig(num_channels=4, embedding_size=64, hidden_sizes=[256, 512, 1024, 2048], depths=[3, 4, 6, 3], layer_type="bottleneck",
                       hidden_act="relu", downsample_in_bottleneck=False)
    hf = ResNetModel(cfg)
    pre = "encoder.extractor.body."
    det = det_state_dict({k: v for k, v in resnet50_fpn_shapes().items() if k.startswith(pre)})

    def hf_key(k):   # torchvision / timm naming -> transformers naming
        parts = k[len(pre):].split(".")
        if parts[0] in ("conv1", "bn1"):
            return "embedder.embedder." + ("convolution." if parts[0] == "conv1" else "normalization.") + parts[-1]
        stage, blk, mod = int(parts[0][5:]) - 1, int(parts[1]), parts[2]
        base = f"encoder.stages.{stage}.layers.{blk}."
        if mod == "downsample":
            return base + "shortcut." + ("convolution." if parts[3] == "0" else "normalization.") + parts[-1]
        return base + f"layer.{int(mod[-1]) - 1}." + ("convolution." if mod.startswith("conv") else "normalization.") + parts[-1]

    mapped = {hf_key(k): v.clone() for k, v in det.items()}
    assert set(mapped) == set(hf.state_dict()), sorted(set(mapped) ^ set(hf.state_dict()))[:6]
    hf.load_state_dict(mapped, strict=True)
    assert sum(p.numel() for p in hf.parameters()) - 64 * 7 * 7 == 23508032, "ResNet-50 body: 23 508 032 parameters with the 3-channel stem"
    g = torch.Generator().manual_seed(41)
    img = torch.rand(2, 4, 96, 128, generator=g)
    go3, go4 = torch.randn(2, 1024, 6, 8, generator=g) * 0.1, torch.randn(2, 2048, 3, 4, generator=g) * 0.1
    keys = ["conv1.weight", "bn1.weight", "layer1.0.conv1.weight", "layer1.0.downsample.0.weight", "layer1.2.bn3.bias", "layer2.0.conv2.weight",
            "layer2.3.conv3.weight", "layer3.0.downsample.1.weight", "layer3.5.conv2.weight", "layer4.0.conv1.weight", "layer4.2.conv3.weight", "layer4.2.bn3.weight"]
    named = dict(hf.named_parameters())
    # (img / go3 / go4 are regenerated from the seed by the tests, like the injected maps of golden_backbone_wrapper; the maps are stored thinned)
    out = {"seed": torch.tensor(41)}
    for mode in ("eval", "train"):
        hf.train(mode == "train")
        hf.zero_grad()
        x = img.clone().requires_grad_(True)
        hs = hf(x, output_hidden_states=True).hidden_states
        l3, l4 = hs[3], hs[4]
        ((l3 * go3).sum() + (l4 * go4).sum()).backward()
        out[mode] = {"layer3": l3.flatten()[::5], "layer4": l4.flatten()[::3], "g_img": x.grad.clone(), "grads": {k: named[hf_key(pre + k)].grad.clone() for k in keys}}
        if mode == "train":   # the running statistics after ONE training forward (momentum 0.1, unbiased variance)
            sd1 = hf.state_dict()
            out[mode]["running"] = {k: sd1[hf_key(pre + k)].clone() for k in ("bn1.running_mean", "layer2.0.bn2.running_var", "layer4.2.bn3.running_mean")}
    save("resnet_body_hf.npz", out)


def golden_e2e_cgl():
    """BASELINE config 3's model: the CGL label set (4 labels -> V = 519, Vc = 549, 4-row layout-encoder label table;
    helpers/layout_tokenizer.py:253-274).  train_loss + gradients (task c) and deterministic sample() tokens (task cwh)."""
    tok = make_tokenizer("cgl", 10)
    feats = ds.Features({"label": ds.Sequence(ds.ClassLabel(names=LABELS["cgl"]))})

    def build(task):
        return raa.ConcateAuxilaryTaskConcateCrossAttnRetrievalAugmentedAutoreg(
            features=feats, tokenizer=tok, dataset_name="cgl", max_seq_length=10, db_dataset=None, top_k=16,
            retrieval_backbone="dreamsim", random_retrieval=False, saliency_k="None", auxilary_task=task)

    out = {}
    model = build("c").eval()
    shapes = load_det(model)
    with open(os.path.join(HERE, "ralf_cgl_state_shapes.json"), "w") as f:
        json.dump({"meta": META, "shapes": {k: list(v) for k, v in shapes.items()}}, f, indent=0)
    B = 3
    batch = synth_batch(51, B, 10, 16, 4)
    torch.manual_seed(199)
    inputs, targets = model.preprocess(clone_batch(batch))
    g = torch.Generator().manual_seed(15)
    feat = torch.randn(B, 256, 3, 4, generator=g).requires_grad_(True)
    StandInBackbone.feat = feat
    model.zero_grad()
    outputs, losses = model.train_loss(inputs, targets)
    losses["nll_loss"].backward()
    named = dict(model.named_parameters())
    out["ralf_c"] = {
        "feat": feat, "gfeat": feat.grad, "logits": outputs["logits"], "loss": losses["nll_loss"],
        "inputs": {k: v for k, v in inputs.items() if torch.is_tensor(v) and k != "image"},
        "retrieved": {k: v for k, v in inputs["retrieved"].items() if k not in ("image", "saliency")},
        "targets": targets,
        "grads": {k: named[k].grad for k in GRAD_KEYS_RALF},
        "gradnorm": torch.sqrt(sum((p.grad ** 2).sum() for p in model.parameters() if p.grad is not None)),
        "meta": {"N_total": torch.tensor(tok.N_total), "preproc_N_total": torch.tensor(model.preprocessor.N_total)},
    }
    model = build("cwh").eval()
    load_det(model)
    batch = synth_batch(52, B, 10, 16, 4)
    torch.manual_seed(177)
    cond, _ = get_condition(clone_batch(batch), "cwh", tok)
    feat = torch.randn(B, 256, 2, 3, generator=torch.Generator().manual_seed(18))
    StandInBackbone.feat = feat
    torch.manual_seed(178)
    with torch.no_grad():
        enc_in, _ = model._create_encoder_inputs(cond)
        torch.manual_seed(178)
        res, vio = model.sample(cond=cond, sampling_cfg=DictConfig(name="deterministic"), cond_type="cwh", return_violation=True, use_backtrack=False)
    out["sample_cwh"] = {
        "feat": feat, "cond_seq": cond.seq,
        "seq_layout_const": enc_in["seq_layout_const"], "seq_layout_const_pad_mask": enc_in["seq_layout_const_pad_mask"],
        "retrieved": {k: v for k, v in cond.retrieved.items() if k not in ("image", "saliency")},
        "result": res,
        "violation": {"total": torch.tensor(float(vio["total"])), "viorated": torch.tensor(float(vio["viorated"]))},
    }
    save("e2e_cgl.npz", out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["tokenizer", "host", "modules", "e2e", "sample", "reranker", "relation", "retrieval_augment", "backbone_wrapper", "e2e_cgl", "resnet_body_hf"]
    fns = {"tokenizer": golden_tokenizer, "host": golden_host_path, "modules": golden_modules, "e2e": golden_e2e, "sample": golden_sample,
           "reranker": golden_reranker, "relation": golden_relation, "retrieval_augment": golden_retrieval_augment,
           "backbone_wrapper": golden_backbone_wrapper, "e2e_cgl": golden_e2e_cgl, "resnet_body_hf": golden_resnet_body_hf}
    for w in which:
        fns[w]()
