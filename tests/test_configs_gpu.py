"""GPU: the BASELINE.json configurations the earlier suites did not reach at their sizes -- CGL (4 labels) in bf16 at B = 64 / 256,
relation decoding at B = 256 -- and the bf16 throughput mode against the fp32 parity mode BLOCK BY BLOCK on the unmodified
initialisation, with error budgets derived from the bf16 unit round-off and the number of rounding points of a block."""
import os
import random

import pytest
import torch

import bench
from test_model_cpu import CGL_LABELS, build
from test_model_gpu import FeatStandIn, load_det
from ralf_amd.synthetic import make_batch, to_device

pytestmark = pytest.mark.gpu
DEV = "cuda"
U = 2.0 ** -9          # bf16 unit round-off (round to nearest, 8 significand bits)


def rel(a, b):
    """|a - b|_F / |b|_F"""
    a, b = a.double().flatten(), b.double().flatten()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def cos(a, b):
    return torch.nn.functional.cosine_similarity(a.flatten().double(), b.flatten().double(), dim=0).item()


def cgl_model(dtype, task="uncond"):
    from ralf_amd.helpers.layout_tokenizer import LabelFeature, LayoutSequenceTokenizer
    from ralf_amd.models.generator import ConcateAuxilaryTaskConcateCrossAttnRetrievalAugmentedAutoreg as RALF

    tok = LayoutSequenceTokenizer(CGL_LABELS, 10)
    torch.manual_seed(0)
    m = RALF(features={"label": LabelFeature(CGL_LABELS)}, tokenizer=tok, dataset_name="cgl", max_seq_length=10, db_dataset=None, top_k=16,
             retrieval_backbone="dreamsim", random_retrieval=False, saliency_k="None", auxilary_task=task, compute_dtype=dtype, pretrained=False)
    return m.to(DEV).train()


def test_cgl_b64_bf16_train_step():
    """BASELINE config 3's model (CGL: 4 labels, V = 519) at the per-GPU batch of the data-parallel run, bf16: loss / logits against
    the fp32 parity mode on identical weights and inputs, then three graph-replayed optimisation steps (dropout on)."""
    from ralf_amd.engine import TrainStep

    m32, m16 = cgl_model("float32"), cgl_model("bfloat16")
    assert m16.tokenizer.N_total == 519
    m16.load_state_dict(m32.state_dict())
    keep = (m32.rt.drop_p, m16.rt.drop_p)
    for m in (m32, m16):
        m.rt.drop_p = lambda p: 0.0
    inputs, targets = m32.preprocess(make_batch(64, 10, num_labels=4, seed=2))
    inputs, targets = to_device(inputs, DEV), to_device(targets, DEV)
    inputs["retrieved"] = {k: v for k, v in inputs["retrieved"].items() if k != "image"}
    out = {}
    for name, m in (("f32", m32), ("bf16", m16)):
        o, l = m.train_loss(inputs, targets)
        out[name] = (o["logits"].float(), l["nll_loss"].item())
    assert out["f32"][0].shape == (64, 50, 519)
    assert abs(out["f32"][1] - out["bf16"][1]) < 3e-2 * abs(out["f32"][1]), out
    assert cos(out["f32"][0], out["bf16"][0]) > 0.999
    del m32
    m16.rt.drop_p = keep[1]
    step = TrainStep(m16, use_graph=True)
    losses = [step(inputs, targets).item() for _ in range(3)]
    assert all(torch.isfinite(torch.tensor(losses))) and losses[-1] < losses[0], losses


def test_cgl_b256_decode_rows_equal_reference_rows_and_bf16(golden):
    """CGL greedy `cwh` decode at B = 256: fp32 rows equal the rows recorded from the reference (3 samples tiled); the bf16
    throughput mode decodes the same labels and >= 90 % identical geometry tokens (an argmax flips where two logits lie within the
    bf16 rounding of each other)."""
    from ralf_amd.engine import GraphedDecode
    from ralf_amd.helpers.task import RetrievalAugmentedConditionalInputsForDiscreteLayout as Cond

    r = golden("e2e_cgl.npz").sub("sample_cwh")
    B = 256
    n0 = r["feat"].shape[0]
    tile = lambda t: t.repeat((B // n0 + 1,) + (1,) * (t.dim() - 1))[:B]   # noqa: E731
    outs = {}
    for dtype in ("float32", "bfloat16"):
        model = load_det(build(task="cwh", dataset="cgl", compute_dtype=dtype), "ralf_cgl_state_shapes.json").eval()
        model.encoder = FeatStandIn(tile(r["feat"]).cuda())
        cond = Cond(image=torch.zeros(B, 4, 8, 8), task="cwh", seq=tile(r["cond_seq"]), mask=None, retrieved={k: tile(v) for k, v in r["retrieved"].items()})
        model._create_encoder_inputs = lambda c: ({"image": c.image, "retrieved": c.retrieved, "seq_layout_const": tile(r["seq_layout_const"]),
                                                   "seq_layout_const_pad_mask": tile(r["seq_layout_const_pad_mask"])}, None)
        dec = GraphedDecode(model, "cwh", {"name": "deterministic"})
        for _ in range(2):
            outs[dtype] = model.sample(cond=cond, sampling_cfg={"name": "deterministic"}, cond_type="cwh", return_violation=False, decoder=dec)
    for k in ("label", "mask", "center_x", "center_y", "width", "height"):
        assert torch.equal(outs["float32"][k], tile(r["result"][k])), k
    a, b = outs["float32"], outs["bfloat16"]
    assert torch.equal(a["label"], b["label"]) and torch.equal(a["mask"], b["mask"])
    m = a["mask"]
    same = torch.stack([(a[k][m] == b[k][m]).float().mean() for k in ("center_x", "center_y", "width", "height")]).mean().item()
    close = max((a[k][m] - b[k][m]).abs().max().item() for k in ("center_x", "center_y"))
    print(f"CGL B=256 decode, bf16 vs fp32: {same:.4f} of the geometry tokens identical, largest centre shift {close:.4f}")
    assert same >= 0.90


def test_relation_b256_rows_equal_reference_rows(golden):
    """BASELINE config 5, relationship task at batch 256: the 4 samples recorded from the reference (relation.npz "sample": tokens
    of sample_relation with back-tracking), tiled to 256 rows.  The reference's loop is sequential per sample and draws from
    Python's `random` only after three failures at one step; every tile that starts from the reference's random state must
    reproduce the reference's rows (all tiles when no draw happens)."""
    from ralf_amd.helpers.layout_tokenizer import LayoutSequenceTokenizer
    from ralf_amd.helpers.task import get_condition
    from test_relation_cpu import LABELS, load_table
    from test_relation_cpu import make_batch as rel_batch

    g = golden("relation.npz")
    r = g.sub("sample")
    table, _ = load_table()
    random.seed(6)
    model = load_det(build(task="relation", relation_table=table), "ralf_state_shapes.json").eval()
    B, n0 = 256, r["feat"].shape[0]
    tile = lambda t: t.repeat((B // n0,) + (1,) * (t.dim() - 1))   # noqa: E731
    tok = LayoutSequenceTokenizer(LABELS, 10)
    batch = rel_batch(g.sub("compute_relation"))
    batch["retrieved"] = [dict(r["retrieved"], image=torch.zeros(n0, 16, 4, 1, 1))]
    random.seed(8)
    torch.manual_seed(8)
    cond, _ = get_condition(batch, "relation", tok)
    random.seed(10)
    torch.manual_seed(10)
    model.preprocessor.set_relation_size(30)
    enc4, seqc4 = model._create_encoder_inputs(cond)          # the constraint serialisation draws its relations here, like the reference
    state = random.getstate()

    def tiled(x):
        if torch.is_tensor(x):
            return tile(x)
        if isinstance(x, dict):
            return {k: tiled(v) for k, v in x.items()}
        return x
    enc, seqc = tiled(enc4), tiled(seqc4)
    model.encoder = FeatStandIn(tile(r["feat"]).cuda())
    cond.image, cond.seq = tile(cond.image), tile(cond.seq)
    model._create_encoder_inputs = lambda c: (enc, seqc)
    draws = [0]
    real_randint = random.randint

    def counting_randint(a, b):
        draws[0] += 1
        return real_randint(a, b)
    random.randint = counting_randint
    try:
        random.setstate(state)
        import time
        t0 = time.perf_counter()
        out, vio = model.sample(cond=cond, sampling_cfg={"name": "deterministic", "temperature": 1.0}, cond_type="relation", return_violation=True,
                                use_backtrack=True, RELATION_SIZE=30, lockstep=False)      # sample after sample: the reference's loop
        dt = time.perf_counter() - t0
    finally:
        random.randint = real_randint
    after_seq = random.getstate()
    print(f"relation decode with back-tracking, B = {B}: {dt * 1e3 / B:.2f} ms per sample, {draws[0]} random back-track draws")
    # the lock-step form, the DEFAULT since round 6 (one batched decoder step for all samples, every element at its own position; range-one draws
    # take their value at once and are consumed from the stream in sample order, other draws wait for the samples before them): the same stream
    # consumption, so ALL 256 rows equal the sequential loop's and `random` is left in the same state
    random.setstate(state)
    out_l = model.sample(cond=cond, sampling_cfg={"name": "deterministic", "temperature": 1.0}, cond_type="relation", return_violation=False,
                         use_backtrack=True, RELATION_SIZE=30)
    assert random.getstate() == after_seq
    for k in ("label", "mask", "center_x", "center_y", "width", "height"):
        assert torch.equal(out_l[k], out[k]), ("lockstep", k)
    rows = B if draws[0] == 0 else n0
    for k in ("label", "mask", "center_x", "center_y", "width", "height"):
        assert out[k].shape[0] == B
        assert torch.equal(out[k][:rows], tile(r["result"][k])[:rows]), k
    if draws[0] == 0:
        assert (vio["total"], vio["viorated"]) == (int(r["violation"]["total"]) * (B // n0), int(r["violation"]["viorated"]) * (B // n0))
    # rng="per_sample" (opt-in throughput mode, NOT the reference's draw order): every sample has a generator of its own, the whole batch decodes
    # in lock-step.  The relation violations are those of the exact mode up to the noise of the few random back-track positions ...
    random.setstate(state)
    t0 = time.perf_counter()
    out_p, vio_p = model.sample(cond=cond, sampling_cfg={"name": "deterministic", "temperature": 1.0}, cond_type="relation", return_violation=True,
                                use_backtrack=True, RELATION_SIZE=30, rng="per_sample")
    dt_p = time.perf_counter() - t0
    print(f"relation decode, rng=per_sample, B = {B}: {dt_p * 1e3 / B:.2f} ms per sample ({dt_p:.2f} s per batch; exact order {dt:.2f} s)")
    assert vio_p["total"] == vio["total"] and out_p["label"].shape == out["label"].shape
    assert abs(vio_p["viorated"] - vio["viorated"]) <= max(8, 0.05 * vio["total"]), (vio_p, vio)
    assert dt_p < 0.5 * dt
    # ... decoded with the masks, gates and arg-maxes of all samples taken on ONE [B, V] array (the default for argmax decoding) or sample by
    # sample with _relation_advance's tensor arithmetic: the same tokens
    random.setstate(state)
    os.environ["RALF_RELATION_BATCHED"] = "0"
    try:
        t0 = time.perf_counter()
        out_q, vio_q = model.sample(cond=cond, sampling_cfg={"name": "deterministic", "temperature": 1.0}, cond_type="relation", return_violation=True,
                                    use_backtrack=True, RELATION_SIZE=30, rng="per_sample")
        print(f"relation decode, rng=per_sample, per-sample host arithmetic: {time.perf_counter() - t0:.2f} s per batch")
    finally:
        del os.environ["RALF_RELATION_BATCHED"]
    for k in ("label", "mask", "center_x", "center_y", "width", "height"):
        assert torch.equal(out_p[k], out_q[k]), k
    assert vio_p == vio_q
    # ... and a batch of ONE is the sequential loop itself: sample 0 continues the global stream and leaves it where the loop would have
    def first(x):
        if torch.is_tensor(x):
            return x[:1].contiguous()
        if isinstance(x, dict):
            return {k: first(v) for k, v in x.items()}
        return x
    enc1, seqc1 = first(enc), first(seqc)
    model._create_encoder_inputs = lambda c: (enc1, seqc1)
    model.encoder = FeatStandIn(tile(r["feat"])[:1].cuda())
    cond.image, cond.seq = cond.image[:1].contiguous(), cond.seq[:1].contiguous()
    res = {}
    for mode in ("shared", "per_sample"):
        random.setstate(state)
        res[mode] = model.sample(cond=cond, sampling_cfg={"name": "deterministic", "temperature": 1.0}, cond_type="relation", return_violation=False,
                                 use_backtrack=True, RELATION_SIZE=30, rng=mode)
        res[mode + "_state"] = random.getstate()
    for k in ("label", "mask", "center_x", "center_y", "width", "height"):
        assert torch.equal(res["shared"][k], res["per_sample"][k]), k
    assert res["shared_state"] == res["per_sample_state"]


# ---- bf16 against fp32, block by block ---------------------------------------------------------------------------------------------
P_FLIP = 2 * 0.3989 * 4 * U     # see _block_case


def _block_case(name, f32, f16, x32, n_round, report, extra32=(), extra16=(), bwd=True, n_relu=0):
    """one block in both precisions on the SAME input (the fp32 model's activation, rounded once for the bf16 block) and the same
    upstream gradient.  Budgets (relative Frobenius error against the fp32 block):
      forward:  2 u (n + 1)      -- n rounding points of u = 2^-9 each + the input's, adding at most linearly; the factor 2 covers the
                                    amplification by a normalisation that follows a small-variance channel;
      backward: 2 u (2 n + 2) + 1.5 sqrt(n_relu * p_flip)  -- the rounding points of the forward AND backward chains, plus the ReLU
                decisions that rounding flips: a pre-activation of unit scale known to ~4 u changes sign with probability
                p_flip = 2 phi(0) 4 u = 0.62 %, and a flipped element carries a whole gradient entry, i.e. sqrt(p_flip) = 8 % of the
                gradient's norm per ReLU.  (The forward does not see this: a flipped pre-activation is ~0 on both sides.)
    All rows are collected; the caller asserts at the end so that the whole table is printed."""
    x32 = x32.detach().requires_grad_(bwd)
    x16 = x32.detach().to(torch.bfloat16).requires_grad_(bwd)
    o32, o16 = f32(x32, *extra32), f16(x16, *extra16)
    e_fwd = rel(o16.float(), o32)
    budget_f, budget_b = 2 * U * (n_round + 1), 2 * U * (2 * n_round + 2) + 1.5 * (n_relu * P_FLIP) ** 0.5
    row = {"fwd": e_fwd, "budget_fwd": budget_f}
    if bwd:
        g = torch.Generator(device=DEV).manual_seed(17)
        dy = torch.randn(o32.shape, device=DEV, generator=g) * o32.detach().abs().mean()
        p32 = [p for p in f32.__self__.parameters() if p.requires_grad] if hasattr(f32, "__self__") else []
        p16 = [p for p in f16.__self__.parameters() if p.requires_grad] if hasattr(f16, "__self__") else []
        for p in p32 + p16:
            p.grad = None
        o32.backward(dy)
        o16.backward(dy.to(torch.bfloat16))
        row["dx"] = rel(x16.grad.float(), x32.grad)
        row["dW_worst"] = max([rel(b.grad, a.grad) for a, b in zip(p32, p16) if a.grad is not None and a.grad.norm() > 0] or [0.0])
        row["budget_bwd"] = budget_b
    report[name] = {k: round(v, 5) for k, v in row.items()}
    return o32.detach()


def test_bf16_block_by_block_error_budgets_plain_init():
    """B = 64, 256x256, the UNMODIFIED initialisation (no damped residual branches): every block of the network -- stem, the 16
    bottlenecks, the FPN, the 6 image-encoder layers, the retrieval fusion, the head FFN, the 6 decoder layers -- is run in
    bf16 and in fp32 on the activation the fp32 network produced, forward and backward.  Errors are measured relative to the fp32
    block (Frobenius) against budgets of u = 2^-9 per rounding point.  (Whole-network gradient cosines of a deep, randomly initialised
    ReLU network are not a parity statement -- mask flips decorrelate them under any perturbation; block budgets are.)"""
    m32 = bench.build_model(torch.device(DEV), 10, "float32")
    m16 = bench.build_model(torch.device(DEV), 10, "bfloat16")
    m16.load_state_dict(m32.state_dict())
    for m in (m32, m16):
        m.rt.drop_p = lambda p: 0.0
        m.rt.to(torch.device(DEV)).begin_step()
    rt32, rt16 = m32.rt, m16.rt
    inputs, _ = m32.preprocess(make_batch(64, 10, seed=1))
    inputs = to_device(inputs, DEV)
    report = {}
    b32, b16 = m32.encoder.extractor, m16.encoder.extractor

    class Stem:
        def __init__(self, bb, rt):
            self.bb, self.rt = bb, rt

        def parameters(self):
            return list(self.bb.body.conv1.parameters()) + list(self.bb.body.bn1.parameters())

        def run(self, img):
            from ralf_amd import functional as RF
            from ralf_amd import ops
            B, C, H, W = img.shape
            x = ops.permute4(img.contiguous().float(), (B, H, W, 8), (4 * H * W, W, 1, H * W), 4, self.rt.dtype)
            y, st = self.bb.body.conv1(x, self.rt, stats=True)
            return RF.MaxPoolFn.apply(self.bb.body.bn1(y, self.rt, True, stats=st))

    img = inputs["image"]
    s32, s16 = Stem(b32, rt32), Stem(b16, rt16)
    with torch.enable_grad():
        x = _block_case("stem", s32.run, s16.run, img, 2, report, bwd=False)   # (the image itself has no gradient)
        feats = {}
        for li in (1, 2, 3, 4):
            for bi, (k32, k16) in enumerate(zip(getattr(b32.body, f"layer{li}"), getattr(b16.body, f"layer{li}"))):
                rt32.begin_step(); rt16.begin_step()
                n_round = 8 if k32.downsample is not None else 6
                x = _block_case(f"layer{li}.{bi}", k32.forward, k16.forward, x, n_round, report, extra32=(rt32,), extra16=(rt16,), n_relu=3)
            feats[li] = x

        class Fpn:
            def __init__(self, bb, rt, l4):
                self.bb, self.rt, self.l4 = bb, rt, l4

            def parameters(self):
                return [p for m in (self.bb.fpn_conv11_4, self.bb.fpn_conv11_5, self.bb.fpn_conv33, self.bb.proj) for p in m.parameters()]

            def run(self, l3):
                return self.bb.fpn(l3, self.l4, self.rt)
        f32_, f16_ = Fpn(b32, rt32, feats[4]), Fpn(b16, rt16, feats[4].to(torch.bfloat16))
        x = _block_case("fpn", f32_.run, f16_.run, feats[3], 5, report)
        for i, (k32, k16) in enumerate(zip(m32.transformer_encoder.layers, m16.transformer_encoder.layers)):
            rt32.begin_step(); rt16.begin_step()
            x = _block_case(f"encoder.{i}", k32.forward, k16.forward, x, 8, report, extra32=(rt32,), extra16=(rt16,), n_relu=1)
        mem = x
        rt32.begin_step(); rt16.begin_step()
        with torch.no_grad():
            ref = m32._retrieved_features(inputs["retrieved"], torch.device(DEV))
        _block_case("fuse_attention", m32.attn.forward, m16.attn.forward, mem, 5, report, extra32=(ref, rt32), extra16=(ref.to(torch.bfloat16), rt16))
        _block_case("head_ffn", m32.head.forward, m16.head.forward, mem, 4, report, extra32=(rt32,), extra16=(rt16,))
        with torch.no_grad():
            from ralf_amd import functional as RF
            h = RF.EmbedFn.apply(inputs["seq"], m32.decoder.emb.weight, m32.decoder.pos_emb.pe[0], rt32)
        kpm = inputs["tgt_key_padding_mask"].to(torch.uint8).contiguous()
        memory = torch.cat([mem, mem[:, :20]], dim=1).contiguous()      # 276 memory rows (content is what the encoder produced)
        for i, (k32, k16) in enumerate(zip(m32.decoder.transformer.layers, m16.decoder.transformer.layers)):
            rt32.begin_step(); rt16.begin_step()
            h = _block_case(f"decoder.{i}", k32.forward, k16.forward, h, 12, report, extra32=(memory, rt32, kpm), extra16=(memory.to(torch.bfloat16), rt16, kpm), n_relu=1)
    worst = {k: max(report[n].get(k, 0.0) / report[n].get("budget_fwd" if k == "fwd" else "budget_bwd", 1.0) for n in report) for k in ("fwd", "dx", "dW_worst")}
    print("bf16 vs fp32 per block (relative Frobenius error):")
    for n, row in report.items():
        print(f"  {n:16s} {row}")
    print("largest used fraction of the budget:", {k: round(v, 3) for k, v in worst.items()})
    for n, row in report.items():
        assert row["fwd"] < row["budget_fwd"], (n, row)
        if "dx" in row:
            assert row["dx"] < row["budget_bwd"] and row["dW_worst"] < row["budget_bwd"], (n, row)
