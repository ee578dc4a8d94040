"""GPU: the fused train step (flat buffers, HIP clip+AdamW, hipGraph replay) matches an eager torch
optimizer on the same model, and graph replay matches eager execution."""
import copy

import pytest
import torch

from test_model_cpu import build
from test_model_gpu import FeatStandIn, load_det, to_dev

pytestmark = pytest.mark.gpu


def make(golden, dtype="float32"):
    r = golden("e2e.npz").sub("ralf_c")
    model = load_det(build(task="c", compute_dtype=dtype), "ralf_state_shapes.json").eval()  # eval: no dropout -> comparable
    model.encoder = FeatStandIn(r["feat"].cuda())
    inputs = to_dev(dict(r["inputs"]))
    inputs["retrieved"] = to_dev(r["retrieved"])
    inputs["image"] = torch.zeros(3, 4, 8, 8, device="cuda")
    return model, inputs, {"seq": r["targets"]["seq"].cuda()}


def test_fused_step_matches_torch_adamw(golden):
    from ralf_amd.engine import TrainStep

    ref, inputs, tgt = make(golden)
    opt = torch.optim.AdamW(ref.optim_groups(1e-4, 1e-4, custom_lr={"encoder.extractor.body": 1e-5}), betas=(0.9, 0.999), eps=1e-8)
    fused, _, _ = make(golden)
    step = TrainStep(fused, lr=1e-4, weight_decay=1e-4, max_norm=0.1, use_graph=False)
    for _ in range(3):
        opt.zero_grad(set_to_none=True)
        loss = ref.train_loss(inputs, tgt)[1]["nll_loss"]
        loss.backward()
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.1)
        opt.step()
        loss_f = step(inputs, tgt)
        torch.testing.assert_close(loss_f, loss.detach(), atol=2e-5, rtol=2e-5)
    a, b = dict(ref.named_parameters()), dict(fused.named_parameters())
    # Adam's first steps move every weight by ~lr*sign(g): where |g| is at rounding level (fp32 atomics reorder the
    # gradient sums) the sign can flip, so a handful of elements may differ by up to steps*lr; all others agree tightly.
    for k in ("decoder.head.1.weight", "transformer_encoder.layers.0.self_attn.in_proj_weight", "task_emb.weight", "attn.to_kv.weight", "head.net.0.bias"):
        diff = (b[k] - a[k]).abs()
        assert diff.max().item() <= 3.2e-4, (k, diff.max().item())
        assert (diff <= 2e-6 + 1e-4 * a[k].abs()).float().mean().item() >= 0.995, k


@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_graph_replay_matches_eager(golden, dtype):
    from ralf_amd.engine import TrainStep

    m1, inputs, tgt = make(golden, dtype)
    m2, _, _ = make(golden, dtype)
    eager, graphed = TrainStep(m1, use_graph=False), TrainStep(m2, use_graph=True)
    le = [eager(inputs, tgt).item() for _ in range(5)]
    lg = [graphed(inputs, tgt).item() for _ in range(5)]   # the first call warms up, captures and replays ONCE: exactly one step
    assert graphed.steps_done == 5 and int(graphed.opt.step_dev) == 5 == int(eager.opt.step_dev)   # (the device counter drives Adam's bias correction)
    assert le[0] > le[-1]                        # it trains
    tol = 2e-2 if dtype == "bfloat16" else 2e-4
    for a, b in zip(le, lg):
        assert abs(a - b) < tol, (le, lg)
    torch.cuda.synchronize()
    mx = (eager.opt.P - graphed.opt.P).abs().max().item()
    assert mx <= (8e-4 if dtype == "bfloat16" else 5.5e-4), mx   # Adam turns sign flips of ~0 gradients into +-lr per step


def test_master_rewrite_refreshes_bf16_shadow(golden):
    """a bf16 TrainStep keeps a bf16 shadow of the fp32 masters that only the AdamW kernel rewrites; load_state_dict (resume,
    best-checkpoint eval) or any other in-place write to the masters must reach the forward too"""
    from ralf_amd.engine import TrainStep

    m1, inputs, tgt = make(golden, "bfloat16")
    step = TrainStep(m1, use_graph=False)
    for _ in range(2):
        step(inputs, tgt)
    fresh, _, _ = make(golden, "bfloat16")                 # the initial weights again
    m1.load_state_dict(fresh.state_dict(), strict=True)    # masters rewritten behind the optimizer's back
    m1.eval(), fresh.eval()
    with torch.no_grad():
        a = m1.train_loss(inputs, tgt)[0]["logits"]
        b = fresh.train_loss(inputs, tgt)[0]["logits"]
    torch.testing.assert_close(a, b, atol=1e-6, rtol=0)
    # explicit form (after torch.distributed.broadcast / manual re-init through .data)
    step.opt.P.mul_(0.5)
    step.opt.sync_shadow()
    torch.testing.assert_close(step.opt.P16.float(), step.opt.P.bfloat16().float(), atol=0, rtol=0)


def test_multistep_lr_follows_torch_scheduler_through_graph_replay(golden):
    """the scheduler factor lives on the device: a captured graph keeps replaying while the learning rate steps down
    (train/schedulers/multi_step_lr.py semantics: milestones as fractions of the epochs, gamma 0.1)."""
    from ralf_amd.engine import MultiStepLR, TrainStep

    ref, inputs, tgt = make(golden)
    opt = torch.optim.AdamW(ref.optim_groups(1e-3, 1e-4, custom_lr={"encoder.extractor.body": 1e-4}), betas=(0.9, 0.999), eps=1e-8)
    sched_ref = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[int(0.5 * 4)], gamma=0.1)
    fused, _, _ = make(golden)
    step = TrainStep(fused, lr=1e-3, weight_decay=1e-4, max_norm=0.1, use_graph=True)
    sched = MultiStepLR(step.opt, epochs=4, milestones=[0.5], gamma=0.1)
    losses_ref, losses = [], []
    for epoch in range(4):
        for _ in range(2):
            opt.zero_grad(set_to_none=True)
            loss = ref.train_loss(inputs, tgt)[1]["nll_loss"]
            loss.backward()
            torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.1)
            opt.step()
            losses_ref.append(loss.item())
        sched_ref.step()
        for _ in range(2):
            step(inputs, tgt)
        sched.step()
        assert sched.get_last_lr()[-1] == pytest.approx(sched_ref.get_last_lr()[-1])
    assert sched.get_last_lr()[-1] == pytest.approx(1e-4)
    assert float(step.opt.lr_scale) == pytest.approx(0.1)


def test_lr_scale_changes_the_update_like_torch(golden):
    from ralf_amd.engine import MultiStepLR, TrainStep

    ref, inputs, tgt = make(golden)
    opt = torch.optim.AdamW(ref.optim_groups(1e-3, 1e-2, custom_lr={"encoder.extractor.body": 1e-4}), betas=(0.9, 0.999), eps=1e-8)
    sched_ref = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[1], gamma=0.1)
    fused, _, _ = make(golden)
    step = TrainStep(fused, lr=1e-3, weight_decay=1e-2, max_norm=0.1, use_graph=False)
    sched = MultiStepLR(step.opt, epochs=2, milestones=[1], gamma=0.1)
    for epoch in range(2):
        for _ in range(2):
            opt.zero_grad(set_to_none=True)
            ref.train_loss(inputs, tgt)[1]["nll_loss"].backward()
            torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.1)
            opt.step()
            step(inputs, tgt)
        sched_ref.step()
        sched.step()
    a, b = dict(ref.named_parameters()), dict(fused.named_parameters())
    for k in ("decoder.head.1.weight", "attn.to_kv.weight", "head.net.0.bias"):
        diff = (b[k] - a[k]).abs()
        # 2 steps at 1e-3 then 2 at 1e-4: an unscaled second epoch would be off by ~1.8e-3 on most elements
        assert (diff <= 3e-5 + 1e-4 * a[k].abs()).float().mean().item() >= 0.99, (k, diff.max().item())


@pytest.mark.parametrize("use_graph,wire", [(False, "fp32"), (True, "fp32"), (False, "bf16")])
def test_staged_backward_with_overlapped_exchange_matches_plain_step(golden, use_graph, wire):
    """data-parallel mode: backward cut after layer2 of the ResNet, stage-1 gradients all-reduced (RCCL, a 1-rank group here)
    while stage 2 runs, three graphs instead of two -- same losses and the same weights as the plain step."""
    import os

    import torch.distributed as dist
    from ralf_amd.engine import TrainStep

    if not dist.is_initialized():   # one 1-rank RCCL group for the whole test process (left to process exit)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29000 + os.getpid() % 2000))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    if True:
        import bench
        from ralf_amd.synthetic import make_batch, to_device

        dev = torch.device("cuda", 0)
        m1, m2 = bench.build_model(dev, 10, "bfloat16"), bench.build_model(dev, 10, "bfloat16")   # the full model: ResNet-50 + FPN in front
        m2.load_state_dict(m1.state_dict())
        inputs, tgt = m1.preprocess(make_batch(4, 10, seed=3))
        inputs, tgt = to_device(inputs, dev), to_device(tgt, dev)
        inputs["retrieved"] = {k: v for k, v in inputs["retrieved"].items() if k != "image"}
        plain = TrainStep(m1, use_graph=use_graph, overlap_allreduce=False)
        staged = TrainStep(m2, use_graph=use_graph, process_group=dist.group.WORLD, overlap_allreduce=True, grad_wire=wire)
        assert staged.exchange.wire == wire and (staged.exchange.stage is not None) == (wire == "bf16")
        assert staged.staged and staged._late and staged._early
        n_late = sum(b - a for a, b in staged._late)
        assert 0 < n_late < 0.1 * staged.opt.G.numel()          # stem + layer1-2: a few percent of the gradient bytes
        # reference for run-to-run noise (fp32 atomics reorder gradient sums; Adam turns sign flips of ~0 gradients into +-lr)
        m3 = bench.build_model(dev, 10, "bfloat16")
        m3.load_state_dict(m1.state_dict())
        again = TrainStep(m3, use_graph=use_graph, overlap_allreduce=False)
        # first step: identical weights everywhere, so the gradient buffers differ only by summation-order noise
        lp, ls, la = [plain(inputs, tgt).item()], [staged(inputs, tgt).item()], [again(inputs, tgt).item()]
        torch.cuda.synchronize()
        if True:   # (graphed or not, the first call is exactly one step)
            for a, b in staged._late + staged._early:
                ga, gs, gn = plain.opt.G[a:b], staged.opt.G[a:b], again.opt.G[a:b]
                noise = (ga - gn).abs().max().item()
                # (bf16 wire: every element makes a round trip through bf16, 2^-9 relative)
                assert (ga - gs).abs().max().item() <= max(4 * noise, (1e-3 if wire == "fp32" else 4e-3) * ga.abs().max().item()), (a, b, noise)
        lp += [plain(inputs, tgt).item() for _ in range(3)]
        ls += [staged(inputs, tgt).item() for _ in range(3)]
        la += [again(inputs, tgt).item() for _ in range(3)]
        for a, b in zip(lp, ls):
            assert abs(a - b) < 2e-2, (lp, ls, la)
        torch.cuda.synchronize()
        mx = (plain.opt.P - staged.opt.P).abs().max().item()
        mx0 = (plain.opt.P - again.opt.P).abs().max().item()
        assert mx <= max(8e-4, 2 * mx0), (mx, mx0)       # Adam turns sign flips of ~0 gradients into +-lr per step
        # stage 2 really produced the early-layer gradients (not zeros)
        a, b = staged._late[0]
        assert staged.opt.G[a:b].abs().sum().item() > 0


def test_static_batch_replay_equals_copied_batch(golden):
    """a loader may fill TrainStep.static_batch() in place and hand the same tensors back: no per-tensor copies, same step"""
    from ralf_amd.engine import TrainStep

    m1, inputs, tgt = make(golden, "float32")
    m2, _, _ = make(golden, "float32")
    a, b = TrainStep(m1, use_graph=True), TrainStep(m2, use_graph=True)
    a(inputs, tgt), b(inputs, tgt)
    si, st = b.static_batch()
    assert si["seq"].data_ptr() != inputs["seq"].data_ptr()
    for _ in range(3):
        la = a(inputs, tgt).item()
        for k, v in tgt.items():          # "the loader": write the batch into the step's buffers
            st[k].copy_(v)
        lb = b(si, st).item()
        assert abs(la - lb) < 1e-3, (la, lb)   # two runs differ by the summation order of fp32 atomics


@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_stacked_cross_kv_projection_equals_per_layer_path(golden, dtype):
    """the decoder's cross-attention K/V projections of all layers in one batched product (forward, weight gradient) and one
    chained-K data-gradient product (functional.CrossKVFn) against the per-layer path: same loss, same flat gradient buffer"""
    from ralf_amd import functional as RF
    from ralf_amd.engine import TrainStep

    m1, inputs, tgt = make(golden, dtype)
    m2, _, _ = make(golden, dtype)
    a, b = TrainStep(m1, use_graph=False), TrainStep(m2, use_graph=False)
    was = RF.CrossKVPlan.enabled
    RF.CrossKVPlan.enabled = True
    try:
        la = a(inputs, tgt).item()
        assert RF.CrossKVPlan.make([l.multihead_attn for l in m1.decoder.transformer.layers], m1.rt) is not None   # the flat layout allows it
        RF.CrossKVPlan.enabled = False
        lb = b(inputs, tgt).item()
    finally:
        RF.CrossKVPlan.enabled = was
    torch.cuda.synchronize()
    tol = 2e-2 if dtype == "bfloat16" else 1e-5
    assert abs(la - lb) < tol, (la, lb)
    ga, gb = a.opt.G, b.opt.G
    rel = ((ga - gb).norm() / gb.norm()).item()
    assert rel < (2e-2 if dtype == "bfloat16" else 1e-4), rel
    for k in ("decoder.transformer.layers.3.multihead_attn.in_proj_weight", "decoder.transformer.layers.5.multihead_attn.in_proj_bias", "head.net.4.weight"):
        p1, p2 = dict(m1.named_parameters())[k], dict(m2.named_parameters())[k]
        r = ((p1.grad - p2.grad).norm() / p2.grad.norm().clamp_min(1e-12)).item()
        assert r < (3e-2 if dtype == "bfloat16" else 1e-4), (k, r)


def test_grouped_wgrads_and_branches_equal_plain_step(golden):
    """engine modes: weight / bias gradients collected into grouped launches + independent sub-networks on their own graph
    branches, against the plain one-launch-per-product, single-stream step: same loss, same flat gradient buffer (bf16)"""
    from ralf_amd.engine import TrainStep

    m1, inputs, tgt = make(golden, "bfloat16")
    m2, _, _ = make(golden, "bfloat16")
    a, b = TrainStep(m1, use_graph=False), TrainStep(m2, use_graph=False)
    assert m1.rt.group_wgrads and m1.rt.branches
    m2.rt.group_wgrads = m2.rt.branches = False
    la, lb = a(inputs, tgt).item(), b(inputs, tgt).item()
    torch.cuda.synchronize()
    assert abs(la - lb) < 1e-4, (la, lb)
    rel = ((a.opt.G - b.opt.G).norm() / b.opt.G.norm()).item()
    assert rel < 2e-3, rel          # identical bf16 products, fp32 sums in a different order
    for k in ("transformer_encoder.layers.2.linear1.weight", "transformer_encoder.layers.2.linear1.bias", "decoder.transformer.layers.4.multihead_attn.in_proj_weight",
              "user_const_encoder.encoder.layers.0.self_attn.out_proj.weight", "head.net.4.weight"):
        p1, p2 = dict(m1.named_parameters())[k], dict(m2.named_parameters())[k]
        r = ((p1.grad - p2.grad).norm() / p2.grad.norm().clamp_min(1e-12)).item()
        assert r < 2e-3, (k, r)


def test_side_graph_mode_equals_the_branch_mode(golden, monkeypatch):
    """RALF_SIDE_GRAPH=1 (opt-in, read when the step is captured): the weight / bias gradient launches are a SECOND graph replayed on the side
    stream, each item behind event nodes of the main graph (functional.ExternalEvent), instead of parallel branches inside the main graph.
    Same kernels, same order per gradient region: losses and parameters agree as two runs of one mode do (the step has fp32 atomics: loss
    and LayerNorm-parameter sums are not bit-reproducible run to run)."""
    from ralf_amd.engine import TrainStep

    m1, inputs, tgt = make(golden, "bfloat16")
    m2, _, _ = make(golden, "bfloat16")
    m3, _, _ = make(golden, "bfloat16")
    monkeypatch.setenv("RALF_SIDE_GRAPH", "0")
    a = TrainStep(m1, use_graph=True)
    la = [a(inputs, tgt).item() for _ in range(4)]
    monkeypatch.setenv("RALF_SIDE_GRAPH", "1")
    b = TrainStep(m2, use_graph=True)
    lb = [b(inputs, tgt).item() for _ in range(4)]
    torch.cuda.synchronize()
    assert a._graphs[3] is None and a.side_graph is False and b._graphs[3] is not None and len(b._graphs[4]) > 0 and b.side_graph is True
    # the default: both captured, three replays of each timed on this model and batch, the faster kept -- and exactly ONE optimisation step done
    monkeypatch.setenv("RALF_SIDE_GRAPH", "auto")
    c = TrainStep(m3, use_graph=True)
    lc = [c(inputs, tgt).item() for _ in range(4)]
    assert set(c.side_graph_ms) == {False, True} and c.side_graph == min(c.side_graph_ms, key=c.side_graph_ms.get) and (c._graphs[3] is not None) == c.side_graph
    assert c.steps_done == 4 and int(c.opt.step_dev) == 4 and max(abs(x - y) for x, y in zip(la, lc)) < 2e-3, (la, lc)
    assert (a.opt.P - c.opt.P).abs().max().item() <= 4 * 2 * 1e-4 + 1e-6
    assert max(abs(x - y) for x, y in zip(la, lb)) < 2e-3 and la[0] > la[-1], (la, lb)
    assert ((a.opt.G - b.opt.G).norm() / a.opt.G.norm()).item() < 2e-2
    # Adam's first steps move every weight by ~lr * sign(g): a sign flip at rounding level moves an element by 2 lr per step
    assert (a.opt.P - b.opt.P).abs().max().item() <= 4 * 2 * 1e-4 + 1e-6 and ((a.opt.P - b.opt.P).abs() > 1e-5).float().mean().item() < 0.02


def test_unchanged_reference_loop_reaches_the_graph_through_graphed_adamw(golden):
    """image2layout/train/train.py:432-454 verbatim (zero_grad -> train_loss -> backward -> clip_grad_norm_ -> optimizer.step) with
    optimizer._target_=ralf_amd.engine.GraphedAdamW trains exactly like TrainStep: the adapter replays the captured step inside train_loss"""
    from ralf_amd.engine import GraphedAdamW, TrainStep

    m1, inputs, tgt = make(golden, "bfloat16")
    m2, _, _ = make(golden, "bfloat16")
    for m in (m1, m2):
        m.train()
        m.rt.drop_p = lambda p: 0.0   # training mode (the adapter's hook), dropout off (comparable runs)
    groups = m1.optim_groups(base_lr=1e-4, weight_decay=1e-4, custom_lr={"encoder.extractor.body": 1e-5})   # train/train.py:217-223
    opt = GraphedAdamW(params=groups, weight_decay=0.01, max_norm=0.1)   # instantiate(cfg.optimizer)(params=...) with the override
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[1], gamma=0.1)
    step = TrainStep(m2, lr=1e-4, weight_decay=1e-4, max_norm=0.1, use_graph=True)
    # train/train.py:434-439 moves only top-level tensors to the device: the nested `retrieved` dict arrives on the HOST
    loop_inputs = dict(inputs, retrieved={k: v.cpu() for k, v in inputs["retrieved"].items()})
    la, lb = [], []
    for it in range(4):
        if it == 0:
            torch.nn.Module.zero_grad(m1)   # what a wrapper's own zero_grad() does (the reference's DDPWrapper): .grad = None BEFORE the capture
        m1.zero_grad()
        out, losses = m1.train_loss(loop_inputs, tgt)
        loss = sum(losses.values())
        loss.backward()
        total = torch.nn.utils.clip_grad_norm_(m1.parameters(), 0.1)
        opt.step()
        la.append(loss.cpu().item())
        lb.append(step(inputs, tgt).item())
        assert out["logits"].shape[0] == inputs["seq"].shape[0] and float(total) == 0.0   # nothing left for the host-side clip to scale
        assert all(p.grad is None for p in m1.parameters())
    assert la[0] > la[-1]
    for a, b in zip(la, lb):
        assert abs(a - b) < 2e-3, (la, lb)
    torch.cuda.synchronize()
    assert (opt.engine.opt.P - step.opt.P).abs().max().item() <= 8e-4
    # evaluate() (train/train.py:492-520: eval mode, no_grad) takes the ordinary forward and does not step
    before = opt.engine.steps_done
    m1.eval()
    with torch.no_grad():
        _, l = m1.train_loss(inputs, tgt, test=True)
    assert opt.engine.steps_done == before and torch.isfinite(l["nll_loss"])
    m1.train()
    # the epoch-wise scheduler rewrites param_groups[i]["lr"]: the captured AdamW follows through its device-resident factor
    sched.step()
    opt.step()
    assert abs(float(opt.engine.opt.lr_scale) - 0.1) < 1e-7


def test_graphed_adamw_loss_lag_reports_the_previous_step(golden):
    """optimizer.loss_lag=1: train_loss returns the previous replay's loss as a host scalar (no wait for the replay just enqueued); the training
    itself is the lag-0 training"""
    from ralf_amd.engine import GraphedAdamW

    curves = []
    for lag in (0, 1):
        m, inputs, tgt = make(golden, "bfloat16")
        m.train()
        m.rt.drop_p = lambda p: 0.0
        opt = GraphedAdamW(params=m.optim_groups(base_lr=1e-4, weight_decay=1e-4), max_norm=0.1, loss_lag=lag)
        ls = []
        for _ in range(5):
            m.zero_grad()
            _, losses = m.train_loss(inputs, tgt)
            loss = sum(losses.values())
            loss.backward()
            opt.step()
            ls.append(loss.cpu().item())
        torch.cuda.synchronize()
        curves.append((ls, opt.engine.opt.P.clone()))
    (l0, p0), (l1, p1) = curves
    assert abs(l1[0] - l0[0]) < 2e-3 and all(abs(a - b) < 2e-3 for a, b in zip(l1[1:], l0[:-1])), (l0, l1)
    assert (p0 - p1).abs().max().item() <= 8e-4


def _slice_batch(tree, n, lc=None):
    """the first n samples of a batch tree (and, lc: constraint sequences cut to lc tokens, as a batch whose longest sample is shorter)"""
    out = {}
    for k, v in tree.items():
        if isinstance(v, dict):
            out[k] = _slice_batch(v, n)
        elif torch.is_tensor(v) and v.dim() > 0:
            v = v[:n]
            if lc is not None and k in ("seq_layout_const", "seq_layout_const_pad_mask"):
                v = v[:, :lc]
            out[k] = v.contiguous()
        else:
            out[k] = v
    return out


def test_graphed_step_follows_the_batch_shape(golden):
    """the reference's loader keeps the smaller last batch of an epoch (drop_last=False, train/train.py:166) and the conditional tasks'
    constraint sequences have batch-dependent lengths (helpers/task_preprocessor.py): a captured graph must never be fed another shape.  Every
    shape signature gets graphs of its own (up to max_graph_shapes), further shapes run the eager step; the sequence of losses and the weights
    equal those of a step that never uses a graph."""
    from ralf_amd.engine import TrainStep

    m1, inputs, tgt = make(golden, "bfloat16")
    m2, _, _ = make(golden, "bfloat16")
    B, Lc = inputs["seq"].shape[0], inputs["seq_layout_const"].shape[1]
    assert B >= 3 and Lc >= 4
    # A, B, A, C, D (cache full: eager), C AGAIN (the shape that was current when the fallback happened: ADVICE r5, its replay must report its
    # own static loss, not the eager step's tensor), B, A
    batches = [(inputs, tgt), (_slice_batch(inputs, B - 1), _slice_batch(tgt, B - 1)), (inputs, tgt), (_slice_batch(inputs, B, Lc - 1), tgt),
               (_slice_batch(inputs, 1), _slice_batch(tgt, 1)), (_slice_batch(inputs, B, Lc - 1), tgt),
               (_slice_batch(inputs, B - 1), _slice_batch(tgt, B - 1)), (inputs, tgt)]
    eager, graphed = TrainStep(m1, use_graph=False), TrainStep(m2, use_graph=True)
    graphed.max_graph_shapes = 3
    le = [eager(i, t).item() for i, t in batches]
    lg, handles = [], []
    for i, t in batches:
        loss = graphed(i, t)
        lg.append(loss.item())
        handles.append((loss.data_ptr(), graphed.outputs["logits"].data_ptr()))
    assert graphed.eager_fallbacks == 1 and len(graphed._by_shape) == 3          # the fourth shape (one sample) ran eagerly
    assert handles[5] == handles[3] and handles[5] != handles[4], "the step after a fallback reports the eager step's tensors"
    assert handles[7] == handles[2] == handles[0] and handles[6] == handles[1]
    assert graphed.steps_done == len(batches) == int(graphed.opt.step_dev)
    for a, b in zip(le, lg):
        assert abs(a - b) <= 2e-2 * max(1.0, abs(a)), (le, lg)
    for (k, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        if p1.requires_grad:
            assert (p1 - p2).abs().max().item() <= 2.5e-3, k      # eight Adam steps of lr 1e-4: sign flips at rounding level bound the distance


def test_graphed_adamw_refuses_a_silent_clip_default(golden):
    from ralf_amd.engine import GraphedAdamW

    m, _, _ = make(golden, "bfloat16")
    with pytest.raises(TypeError, match="max_norm"):
        GraphedAdamW(params=m.optim_groups(base_lr=1e-4, weight_decay=1e-4))
