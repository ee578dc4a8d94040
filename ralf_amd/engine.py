"""Training-step engine: flat parameter / gradient buffers, fused clip + AdamW (HIP), whole-step
hipGraph capture, data-parallel gradient all-reduce over RCCL.

Replaces the inner loop of image2layout/train/train.py:409-489 (`train`): forward, loss.backward(),
clip_grad_norm_(0.1), optimizer.step() -- with the same arithmetic (AdamW groups from
`optim_groups`, global-norm clip) but
  * gradients of all trainable tensors live in ONE fp32 buffer (one collective, one norm pass),
  * the update is one kernel per (lr, weight-decay) group and also refreshes the bf16 weight shadow,
  * the whole step is captured once into a hipGraph and replayed (launch-bound otherwise: ~1.5k
    kernels per step), with the dropout seed advanced on the device.
The reference's DDP wiring never all-reduces (SURVEY.md section 5 "DDP quirk"); this engine does.
"""
from __future__ import annotations

import os
import weakref
from typing import Optional

import torch

from . import ops
from .functional import ExternalEvent


# other threads (RCCL's watchdog polls events) may touch the HIP runtime while this thread captures
_CAPTURE_MODE = "thread_local"


def _align(n, a=64):
    return (n + a - 1) // a * a


class FlatAdamW:
    """AdamW + global-norm clipping over flat buffers; `groups` as returned by model.optim_groups()."""

    def __init__(self, groups, betas=(0.9, 0.999), eps=1e-8, max_norm: float = 0.0, shadow_dtype=None, runtime=None):
        self.betas, self.eps, self.max_norm = betas, eps, max_norm
        params = [p for g in groups for p in g["params"]]
        dev = params[0].device
        sizes = [_align(p.numel()) for p in params]
        total = _align(sum(sizes), 1024)   # (tail padding, never updated: every range of the staged exchange can end on a multiple of world * 64)
        self.P = torch.zeros(total, dtype=torch.float32, device=dev)
        self.G = torch.zeros(total, dtype=torch.float32, device=dev)
        self.M = torch.zeros(total, dtype=torch.float32, device=dev)
        self.V = torch.zeros(total, dtype=torch.float32, device=dev)
        self.P16 = torch.zeros(total, dtype=torch.bfloat16, device=dev) if shadow_dtype == torch.bfloat16 else None
        self.groups = []
        off = 0
        for g in groups:
            start = off
            for p in g["params"]:
                n = p.numel()
                self.P[off:off + n].copy_(p.data.reshape(-1))
                p.data = self.P[off:off + n].view(p.shape)
                p.grad = self.G[off:off + n].view(p.shape)
                if runtime is not None:
                    runtime.register_grad_view(p, p.grad)
                if self.P16 is not None and runtime is not None:
                    runtime.register_shadow(p, self.P16[off:off + n].view(p.shape))
                off += _align(n)
            self.groups.append({"range": (start, off), "lr": g["lr"], "weight_decay": g["weight_decay"]})
        if self.P16 is not None:
            self.P16.copy_(self.P)
        self.step_count = 0
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)  # device-resident step counter (graph replay)
        self.runtime = runtime
        self._ss = torch.zeros(ops.SUMSQ_PARTS, dtype=torch.float32, device=dev)   # per-workgroup partial sums of |G|^2 (deterministic clip)
        self.coef = torch.ones(1, dtype=torch.float32, device=dev)
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=dev)
        self.lr_scale = torch.ones(1, dtype=torch.float32, device=dev)   # scheduler factor, read by the kernel (graph-replay safe)

    def zero_grad(self):
        ops.zero_(self.G)

    def sync_shadow(self):
        """the fp32 masters were rewritten behind the optimizer's back (load_state_dict, broadcast, re-init, EMA copy): refresh
        the bf16 shadow the forward reads and invalidate the runtime's derived weight layouts"""
        if self.P16 is not None:
            ops.cast_into(self.P, self.P16)
        if self.runtime is not None:
            for e in self.runtime._shadow.values():   # the views are current again: re-stamp them
                w = e[3]()
                if w is not None:
                    e[1], e[2] = w._version, w.data_ptr()
            self.runtime.weights_changed()

    def clip(self):
        # fixed-order reduction: the same G gives the same coefficient bit for bit on every rank (replicas stay identical)
        ops.sumsq_partials(self.G, self._ss)
        ops.clip_coef_partials(self._ss, self.max_norm, self.coef, self.grad_norm)

    def step(self):
        """the step counter for the bias corrections is advanced ON THE DEVICE, so a captured graph
        replays with the right corrections."""
        self.step_count += 1
        ops.counter_add_(self.step_dev, 1)
        if self.max_norm > 0:
            self.clip()
        for g in self.groups:
            a, b = g["range"]
            ops.adamw(self.P[a:b], self.G[a:b], self.M[a:b], self.V[a:b], g["lr"], self.betas[0], self.betas[1], self.eps,
                      g["weight_decay"], 1, self.coef if self.max_norm > 0 else None,
                      self.P16[a:b] if self.P16 is not None else None, self.step_dev, self.lr_scale)
        if self.runtime is not None:
            self.runtime.weights_changed()


class MultiStepLR:
    """train/schedulers/multi_step_lr.py:10-46 for the fused optimizer: at the given fractions of `epochs` (or absolute
    epochs) the learning rate of every group is multiplied by gamma.  step() once per epoch, like train/train.py:283-287."""

    def __init__(self, optimizer: FlatAdamW, epochs: int, milestones=(0.7,), gamma: float = 0.1, **_ignored):
        ms = list(milestones)
        if isinstance(ms[0], float):
            assert all(0.0 <= m <= 1.0 for m in ms)
            ms = [int(m * epochs) for m in ms]
        self.opt, self.milestones, self.gamma, self.epoch = optimizer, sorted(int(m) for m in ms), gamma, 0
        self.base_lrs = [g["lr"] for g in optimizer.groups]

    def _factor(self) -> float:
        return self.gamma ** sum(1 for m in self.milestones if m <= self.epoch)

    def step(self) -> None:
        self.epoch += 1
        self.opt.lr_scale.fill_(self._factor())

    def get_last_lr(self):
        return [lr * self._factor() for lr in self.base_lrs]


def average_gradients(flat: torch.Tensor, world: int, group=None) -> None:
    """data-parallel gradient exchange: ONE sum all-reduce over the flat gradient buffer.  The backward
    pass is seeded with 1/world, so the sum IS the average (no extra scaling pass).  backend "nccl" is RCCL
    over xGMI on ROCm; "gloo" is used by the CPU tests."""
    if world > 1:
        torch.distributed.all_reduce(flat, op=torch.distributed.ReduceOp.SUM, group=group)


class GradExchange:
    """Sum of the flat fp32 gradient buffer over the ranks, by element ranges (the staged backward exchanges the buffer in
    two parts, TrainStep._exchange_around).  Replaces what DDP's reducer would do behind train/train.py:207-210.

    wire = "fp32" (default): the fp32 ranges themselves are reduced in place (172 MB per step for RALF), like DDP.
    wire = "bf16" (opt-in): each range is packed to a bf16 staging buffer (ralf_copy2d), reduced there and unpacked back into the
        fp32 buffer: HALF the bytes per xGMI link (86 MB); the backward was seeded with 1/world, so the wire carries
        averages-in-the-making of bf16-computed gradients (their own rounding is 2^-9 relative, no error feedback) and clip /
        AdamW still run on fp32.
    mode = "allreduce": one all_reduce per range.
    mode = "rs_ag": reduce_scatter_tensor into this rank's 1/world shard of the range, then all_gather_into_tensor of the shards,
        both in place -- the two halves of a ring all-reduce as separate collectives, each of which RCCL can spread over the 7
        point-to-point xGMI links of a rank instead of one ring.  Ranges whose length is not a multiple of the
        world size fall back to all_reduce.
    `pack(src_fp32, dst_bf16)` / `unpack(src_bf16, dst_fp32)` default to the HIP cast kernel; the gloo CPU tests pass
    torch copies."""

    def __init__(self, flat: torch.Tensor, world: int, group=None, wire: str = "fp32", pack=None, unpack=None, mode: str = "allreduce"):
        assert wire in ("fp32", "bf16") and mode in ("allreduce", "rs_ag")
        self.flat, self.world, self.group, self.wire, self.mode = flat, world, group, wire, mode
        self.stage = torch.empty(flat.numel(), dtype=torch.bfloat16, device=flat.device) if (wire == "bf16" and self.active) else None
        self._pack = pack or ops.cast_into
        self._unpack = unpack or ops.cast_into
        self.rank = torch.distributed.get_rank(group) if self.active and torch.distributed.is_initialized() else 0
        self.ran: dict = {}
        self.rounds = 0          # collective launches so far (diagnostics: ranks of one job must agree, tests/test_two_ranks_gpu.py)

    @property
    def active(self) -> bool:
        return self.world > 1 or self.group is not None

    def bytes_on_wire(self, ranges) -> int:
        return sum(b - a for a, b in ranges) * (2 if self.wire == "bf16" else 4)

    def _shard(self, buf, a, b):
        n = (b - a) // self.world
        return buf[a + self.rank * n:a + (self.rank + 1) * n]

    def start(self, ranges, async_op: bool):
        """launch the exchange of `ranges`; returns a token for finish()"""
        if not self.active:
            return None
        dist = torch.distributed
        buf = self.flat if self.wire == "fp32" else self.stage
        works = []
        for a, b in ranges:
            self.rounds += 1
            if self.wire == "bf16":
                self._pack(self.flat[a:b], self.stage[a:b])
            rs = self.mode == "rs_ag" and (b - a) % self.world == 0 and b > a
            self.ran[(a, b)] = "rs_ag" if rs else "allreduce"   # (diagnostics: bench.py reports which collective carried each range)
            if rs:
                works.append((dist.reduce_scatter_tensor(self._shard(buf, a, b), buf[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=async_op), a, b))
            else:
                works.append((dist.all_reduce(buf[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=async_op), None, None))
        return works, ranges

    def finish(self, token) -> None:
        if token is None:
            return
        works, ranges = token
        buf = self.flat if self.wire == "fp32" else self.stage
        gathers = []
        for w, a, b in works:
            if w is not None:
                w.wait()
            if a is not None:   # second half: every rank's reduced shard to everyone (issued once the shard is complete)
                gathers.append(torch.distributed.all_gather_into_tensor(buf[a:b], self._shard(buf, a, b), group=self.group, async_op=True))
        for g in gathers:
            g.wait()
        if self.wire == "bf16":
            for a, b in ranges:
                self._unpack(self.stage[a:b], self.flat[a:b])

    def run(self, ranges) -> None:
        self.finish(self.start(ranges, False))


def flat_ranges(params, flat: torch.Tensor, align: int = 1):
    """[(begin, end)] element ranges of the flat buffer `flat` covered by the .grad views of `params` (views INTO flat),
    adjacent views merged (gaps up to `align` - 1 padding elements are bridged)."""
    es = flat.element_size()
    iv = sorted(((p.grad.data_ptr() - flat.data_ptr()) // es, p.numel()) for p in params)
    out = []
    for a, n in iv:
        assert 0 <= a and a + n <= flat.numel(), "parameter gradient is not a view into the flat buffer"
        if out and a - out[-1][1] < align:
            out[-1][1] = a + n
        else:
            out.append([a, a + n])
    return [(a, b) for a, b in out]


def _round_ranges(ranges, q: int, total: int):
    """grow every [a, b) to multiples of q (clipped to the buffer, overlaps merged)"""
    out = []
    for a, b in sorted(ranges):
        a, b = a // q * q, min(total, (b + q - 1) // q * q)
        if out and a <= out[-1][1]:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return [(a, b) for a, b in out]


def complement_ranges(ranges, total: int):
    out, pos = [], 0
    for a, b in sorted(ranges):
        if a > pos:
            out.append((pos, a))
        pos = max(pos, b)
    if pos < total:
        out.append((pos, total))
    return out


def _clone_tree(x, device=None):
    """private copies of all tensors of a tree; with `device`: ON that device (the unchanged reference loop leaves the nested
    `retrieved` dict on the host, train/train.py:434-439 -- a captured graph cannot hold a host-to-device copy)"""
    if torch.is_tensor(x):
        return x.clone() if (device is None or x.device == device) else x.to(device)
    if isinstance(x, dict):
        return {k: _clone_tree(v, device) for k, v in x.items()}
    return x


def _record_tree(x, stream):
    if torch.is_tensor(x):
        if x.is_cuda:
            x.record_stream(stream)
    elif isinstance(x, dict):
        for v in x.values():
            _record_tree(v, stream)


class _HostStager:
    """host-resident leaves of a batch tree (the nested `retrieved` dict the reference's loop leaves on the CPU) reach the static device buffers
    through page-locked mirrors: a pageable source makes every copy_ a synchronous staging copy that waits for the stream to drain, i.e. for
    the previous step's replay (11 ms of host time per iteration measured).  Three mirror sets rotate; a set is reused only after the
    event that follows its copies."""

    def __init__(self, depth: int = 3):
        self.sets = [dict() for _ in range(depth)]
        self.events = [None] * depth
        self.i = 0
        self.cur = None

    def begin(self):
        j = self.i % len(self.sets)
        self.i += 1
        if self.events[j] is not None:
            self.events[j].synchronize()
        self.cur = j

    def mirror(self, dst, src):
        m = self.sets[self.cur].get(id(dst))
        if m is None or m.shape != src.shape or m.dtype != src.dtype:
            m = self.sets[self.cur][id(dst)] = torch.empty(src.shape, dtype=src.dtype, pin_memory=True)
        m.copy_(src)
        return m

    def end(self):
        ev = torch.cuda.Event()
        ev.record()
        self.events[self.cur] = ev


def _tree_signature(x):
    """shapes and dtypes of a batch tree's tensors (what a captured graph is specific to)"""
    if torch.is_tensor(x):
        return (tuple(x.shape), str(x.dtype))
    if isinstance(x, dict):
        return tuple((k, _tree_signature(v)) for k, v in sorted(x.items()) if v is not None)
    return None


def _copy_tree(dst, src, stager=None):
    if torch.is_tensor(dst):
        if dst is not src:   # a caller that filled the step's own buffers in place (static_batch()) pays no copy
            # (copy_ would BROADCAST a size-1 tail batch into 64 rows and train on duplicates: the shapes were matched by the caller, checked again here)
            assert dst.shape == src.shape, f"batch tensor of shape {tuple(src.shape)} for a graph captured on {tuple(dst.shape)}"
            if stager is not None and dst.is_cuda and not src.is_cuda and not src.is_pinned():
                src = stager.mirror(dst, src)
            dst.copy_(src, non_blocking=True)
    elif isinstance(dst, dict):
        for k in dst:
            if dst[k] is not None:
                _copy_tree(dst[k], src[k], stager)


class TrainStep:
    """One optimisation step of `model` (a ralf_amd generator) = forward + backward + (all-reduce) + clip + AdamW."""

    def __init__(self, model, lr=1e-4, weight_decay=1e-4, max_norm=0.1, backbone_lr_scale=0.1, betas=(0.9, 0.999), eps=1e-8,
                 use_graph=True, process_group=None, overlap_wgrad=True, overlap_allreduce=None, grad_wire=None, grad_exchange=None, groups=None,
                 own_stream=False):
        """groups: ready-made AdamW groups (what `model.optim_groups(...)` returned to the caller, train/train.py:217-223) instead of
        lr / weight_decay / backbone_lr_scale"""
        self.model = model
        rt = model.rt.to(model.device)
        # own_stream: the step (batch copy into the static buffers, replays, collectives) runs on a library-owned stream, ordered after the
        # caller's stream at the call but NOT the other way round: the caller's next host-to-device copy does not queue up behind the replay
        # (measured in the reference's loop: `.to(rank)` of the next batch waited 22 ms for the previous step).  Results are ordered for the
        # caller by sync_caller().
        self._run = ops.own_stream("run", model.device) if own_stream else None
        if groups is None:
            groups = model.optim_groups(base_lr=lr, weight_decay=weight_decay, custom_lr={"encoder.extractor.body": lr * backbone_lr_scale})
        self.opt = FlatAdamW(groups, betas, eps, max_norm, shadow_dtype=rt.dtype if rt.dtype == torch.bfloat16 else None, runtime=rt)
        rt.direct_grads = True   # kernels accumulate parameter gradients straight into the flat buffer
        rt.overlap = overlap_wgrad  # ... on a side stream: a parallel branch of the captured graph
        rt.branches = os.environ.get("RALF_BRANCHES", "1") != "0"   # independent sub-networks on their own graph branches
        rt.group_wgrads = os.environ.get("RALF_GROUP_WGRADS", "1") != "0"   # weight / bias gradients in grouped launches
        self.use_graph = use_graph
        self.pg = process_group
        self.world = torch.distributed.get_world_size(process_group) if (process_group is not None or torch.distributed.is_initialized()) else 1
        # gradient exchange: fp32 on the wire, like the reference's DDP averages (train/train.py:207-210).  "bf16" (half the bytes per
        # xGMI link, no error feedback) changes the training numerics and is OPT-IN: grad_wire="bf16" or RALF_GRAD_WIRE=bf16
        wire = grad_wire or os.environ.get("RALF_GRAD_WIRE", "fp32")
        assert wire == "fp32" or rt.dtype == torch.bfloat16, "the bf16 wire belongs to the bf16 throughput mode"
        self.exchange = GradExchange(self.opt.G, self.world, process_group, wire, mode=grad_exchange or os.environ.get("RALF_GRAD_EXCHANGE", "allreduce"))
        if self.world > 1:
            self._sync_replicas(rt)
        self._static = None
        self._stager = None
        self._graphs = None
        self._seed_grad = None
        self.loss = None
        self.outputs = None
        self.steps_done = 0
        # A captured graph is specific to the SHAPES of its batch.  The reference's loader keeps the last, smaller batch of an epoch
        # (drop_last=False, train/train.py:166) and the conditional tasks' constraint sequences have batch-dependent lengths
        # (helpers/task_preprocessor.py: kmax = n_valid.max()): each new shape signature gets graphs of its own, up to `max_graph_shapes`
        # of them (activations of every set stay allocated); further shapes run the same step eagerly -- same arithmetic, launch-bound.
        self._by_shape: dict = {}
        self._sig = None
        self.max_graph_shapes = int(os.environ.get("RALF_MAX_GRAPH_SHAPES", "3"))
        self.eager_fallbacks = 0
        self.captures = 0
        # Data parallel: the backward runs in two stages around rt.grad_cut() (after layer2 of the ResNet).  Stage 1
        # (decoder, encoders, FPN, layer4, layer3) completes 94 % of the gradient bytes; their all-reduce runs on RCCL's
        # stream WHILE stage 2 (layer2, layer1, stem: most of the backbone's backward time, 6 % of the bytes) computes.
        # Only the small stage-2 exchange is exposed.  (xGMI is point-to-point: 172 MB cost ~1-2 ms per step un-overlapped.)
        self.staged = bool(overlap_allreduce) if overlap_allreduce is not None else self.world > 1
        self._late = self._early = None
        if self.staged:
            before = self._params_before_cut()
            if before:
                # boundaries on multiples of world * 64 elements (parameters start on multiples of 64; the few elements a range grows by
                # are whole neighbours or zero padding, exchanged twice at worst): every range keeps the reduce-scatter + all-gather form
                self._late = _round_ranges(flat_ranges(before, self.opt.G, align=64), 64 * max(self.world, 1), self.opt.G.numel())   # exchanged after stage 2
                self._early = complement_ranges(self._late, self.opt.G.numel())     # exchanged during stage 2
            else:
                self.staged = False

    def _params_before_cut(self):
        out = []
        for m in self.model.modules():
            if hasattr(m, "parameters_before_cut"):
                out += m.parameters_before_cut()
        return out

    # ---- the step body, split at the collectives so multi-GPU runs keep RCCL outside the graphs ----
    def _fwd_bwd(self, inputs, targets):
        """forward + backward (stage 1 only when staged: down to the grad_cut points)"""
        rt = self.model.rt
        self.opt.zero_grad()
        rt.cut_enabled, rt._cuts = self.staged, []
        rt._fans = []
        try:
            out, losses = self.model._train_loss(inputs, targets)
        finally:
            rt.cut_enabled = False
        loss = losses["nll_loss"]
        self.outputs = {k: v.detach() for k, v in out.items() if torch.is_tensor(v)}   # (inside a capture: the graph's static output tensors)
        if self._seed_grad is None or self._seed_grad.device != loss.device:   # d(loss) = 1 / world, a constant made once (no fill kernel per step)
            self._seed_grad = torch.full_like(loss, 1.0 / self.world)
        loss.backward(self._seed_grad)
        rt.join_side()
        rt.join_all_branches()
        if not rt._cuts:
            rt.check_fans()
        return loss.detach()

    def _bwd_rest(self):
        """stage 2: the backward of everything in front of the grad_cut points"""
        rt = self.model.rt
        cuts, rt._cuts = rt._cuts, []
        if cuts:
            torch.autograd.backward([o for o, _ in cuts], [leaf.grad for _, leaf in cuts])
            rt.join_side()
            rt.join_all_branches()
            rt.check_fans()

    def _sync_replicas(self, rt):
        """what the reference's DDP constructor does (train/train.py:208): every rank starts from rank 0's parameters and
        buffers; the dropout generator is decorrelated across ranks (seed offset by rank)."""
        rank = torch.distributed.get_rank(self.pg)
        src = torch.distributed.get_global_rank(self.pg, 0) if self.pg is not None else 0
        torch.distributed.broadcast(self.opt.P, src, group=self.pg)
        for b in self.model.buffers():
            if b.is_floating_point() or b.dtype in (torch.int64, torch.int32, torch.bool, torch.uint8):
                torch.distributed.broadcast(b.data, src, group=self.pg)
        self.opt.sync_shadow()
        rt._seed_host = rt._seed_host + 0x51ED27 * rank
        rt.seed = None
        rt.to(self.model.device)

    def _allreduce(self):
        self.exchange.run([(0, self.opt.G.numel())])

    def _update(self):
        self.opt.step()
        self.model.rt.advance_seed()

    def _exchange_around(self, stage2):
        """stage-1 gradients on the wire while `stage2` (the rest of the backward) is issued; then the stage-2 gradients"""
        token = self.exchange.start(self._early, True)
        stage2()
        self.exchange.finish(token)
        self.exchange.run(self._late)

    def _eager(self, inputs, targets):
        loss = self._fwd_bwd(inputs, targets)
        if self.staged:
            self._exchange_around(self._bwd_rest)
        else:
            self._allreduce()
        self._update()
        return loss

    def __call__(self, inputs, targets):
        if self._run is None:
            return self._step(inputs, targets)
        self._run.wait_stream(torch.cuda.current_stream())
        _record_tree({"inputs": inputs, "targets": targets}, self._run)   # (allocated on the caller's stream, read on this one)
        with torch.cuda.stream(self._run):
            return self._step(inputs, targets)

    def sync_caller(self):
        """order the caller's current stream after everything issued by the step so far (own_stream mode)"""
        if self._run is not None:
            torch.cuda.current_stream().wait_stream(self._run)

    def _step(self, inputs, targets):
        sig = _tree_signature({"inputs": inputs, "targets": targets}) if self.use_graph else None
        if self.use_graph and sig != self._sig:   # another batch shape than the last step's: its own graphs, or (cache full) the eager step
            if self._sig is not None:
                self._by_shape[self._sig] = (self._static, self._graphs, self.loss, self.outputs)   # (every captured set, the current one included)
            if sig in self._by_shape:
                self._static, self._graphs, self.loss, self.outputs = self._by_shape[sig]
                self._sig = sig
            elif len(self._by_shape) < self.max_graph_shapes:
                self._static, self._graphs, self._sig = None, None, sig
            else:
                sig = None
        if not self.use_graph or sig is None:
            if self.use_graph:
                # shape cache full: the same arithmetic, eagerly.  The graphs of the previous shape keep THEIR static loss / output tensors
                # (saved in _by_shape above); forgetting the current signature makes the next graphed step restore its own set from the
                # cache instead of replaying into handles this eager step has just replaced (a stale loss after a fallback).
                self.eager_fallbacks += 1
                self._static = self._graphs = self._sig = None
            self.loss = self._eager(inputs, targets)
        else:
            if self._graphs is None:
                self._capture(inputs, targets)
            else:
                if self._stager is None:
                    self._stager = _HostStager()
                self._stager.begin()
                _copy_tree(self._static, {"inputs": inputs, "targets": targets}, self._stager)
                self._stager.end()
            ga, gm, gb, gs, _work, gs2, _work2 = self._graphs
            if gm is None:
                self._replay_with_side(ga, gs)   # (the parameter-gradient work: branches of ga, or a graph of its own on the side stream)
                self._allreduce()
            elif gs is None and gs2 is None:
                ga.replay()
                self._exchange_around(gm.replay)
            else:
                self._staged_with_side(ga, gs, gm, gs2, True)
            gb.replay()
        self.steps_done += 1
        return self.loss

    def static_batch(self):
        """(inputs, targets) trees the captured graphs read: a loader may write the next batch INTO these tensors and pass
        them back to __call__, which then replays without the ~120 small device copies of a fresh batch."""
        assert self._static is not None, "static_batch() exists after the first graphed step"
        return self._static["inputs"], self._static["targets"]

    def _snapshot(self):
        """everything a step mutates besides the gradients: optimizer state, step counters, dropout seed, model buffers
        (BatchNorm running statistics / num_batches_tracked)"""
        o, rt = self.opt, self.model.rt
        return {"P": o.P.clone(), "M": o.M.clone(), "V": o.V.clone(), "P16": None if o.P16 is None else o.P16.clone(),
                "step_dev": o.step_dev.clone(), "step_count": o.step_count, "seed": rt.seed.clone(),
                "buffers": [b.clone() for b in self.model.buffers()], "steps_done": self.steps_done}

    def _restore(self, snap):
        o, rt = self.opt, self.model.rt
        o.P.copy_(snap["P"]); o.M.copy_(snap["M"]); o.V.copy_(snap["V"])
        if o.P16 is not None:
            o.P16.copy_(snap["P16"])
        o.step_dev.copy_(snap["step_dev"])
        o.step_count = snap["step_count"]
        rt.seed.copy_(snap["seed"])
        for b, v in zip(self.model.buffers(), snap["buffers"]):
            b.copy_(v)
        self.steps_done = snap["steps_done"]
        rt.weights_changed()

    def _replay_with_side(self, g, gs):
        """a backward graph and, on the side stream, its side graph (whose wait nodes refer to the records `g` just enqueued); joined"""
        g.replay()
        if gs is not None:
            side = self.model.rt._side[0]
            with torch.cuda.stream(side):
                gs.replay()
            torch.cuda.current_stream().wait_stream(side)

    def _staged_with_side(self, ga, gs, gm, gs2, exchange: bool):
        """the staged backward with side graphs: the main stream goes straight from stage 1 to stage 2; the side stream runs stage 1's weight
        gradients, then -- behind all of stage 1 -- ISSUES the exchange of the stage-1 ranges (the collective's stream waits for the stream it is
        issued from), then stage 2's weight gradients; one join at the end.  Nothing of the side work sits between the two stages of the chain."""
        rt, main = self.model.rt, torch.cuda.current_stream()
        side = rt._side[0]
        ga.replay()
        token = None
        with torch.cuda.stream(side):
            if gs is not None:
                gs.replay()
            side.wait_stream(main)      # (all of stage 1: the directly written gradients of the chain as well)
            if exchange:
                token = self.exchange.start(self._early, True)
        gm.replay()
        if gs2 is not None:
            with torch.cuda.stream(side):
                gs2.replay()
        main.wait_stream(side)
        if exchange:
            self.exchange.finish(token)
            self.exchange.run(self._late)

    def _time_fwd_bwd(self, pairs, stream, reps: int = 3) -> float:
        """milliseconds per replay of the forward + backward graphs (each with its side graph, joined): one untimed pass, then `reps` timed ones"""
        def once():
            (ga, gs), (gm, gs2) = pairs
            if gm is None:
                self._replay_with_side(ga, gs)
            elif gs is None and gs2 is None:
                ga.replay()
                gm.replay()
            else:
                self._staged_with_side(ga, gs, gm, gs2, False)
        with torch.cuda.stream(stream):
            once()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                once()
            e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / reps

    def _capture(self, inputs, targets):
        # No collector pass while graphs are being captured: a pass that runs inside a capture frees whatever cyclic garbage it finds there and
        # then (graph executables, events of an earlier TrainStep), and a destructor's HIP call on the capturing thread aborts the process
        # (seen as "Fatal Python error: Aborted ... Garbage-collecting" in the middle of a captured forward once a capture had become several
        # captures and timed replays).  Everything collectable goes before the first capture begins.
        import gc
        gc.collect()
        was = gc.isenabled()
        gc.disable()
        try:
            self._capture_impl(inputs, targets)
        finally:
            if was:
                gc.enable()

    def _capture_impl(self, inputs, targets):
        self._static = _clone_tree({"inputs": inputs, "targets": targets}, self.model.device)
        si, st = self._static["inputs"], self._static["targets"]
        # the warm-up steps (allocator, lazy inits, LDS attributes) must not train: the state they touch is put back, so the
        # first graphed call is exactly ONE optimisation step, like use_graph=False and like the reference's loop
        snap = self._snapshot()
        # warm-up: one full step (collectives included) on a stream of its own, then the CAPTURED pieces alone on the capture
        # stream, so its per-stream workspaces reach their size before the capture and outside its pool.  No collective may ever
        # run on the capture stream: RCCL issues a synchronous collective on the CURRENT stream, and its completion event --
        # polled by the watchdog thread -- must not sit on a stream that starts capturing (hipErrorCapturedEvent -> abort).
        # Data parallel: only the FIRST capture's warm-up runs the gradient exchange (every rank captures at its first step, so the rounds
        # match and RCCL / the wire's staging buffers are warm afterwards).  A later capture is triggered by THIS rank's batch shape (ragged
        # last batch, kmax = n_valid.max() of its own data) while other ranks replay a cached shape: a warm-up exchange here would pair this
        # rank's throw-away gradients with their real ones and shift every later round by one.
        side, cap = ops.own_stream("warmup"), ops.own_stream("capture")
        side.wait_stream(torch.cuda.current_stream())
        if self.world == 1 or self.captures == 0:
            with torch.cuda.stream(side):
                self._eager(si, st)
        self.captures += 1
        cap.wait_stream(side)
        with torch.cuda.stream(cap):
            self._fwd_bwd(si, st)
            if self.staged:
                self._bwd_rest()
            self._update()
        side.wait_stream(cap)
        torch.cuda.current_stream().wait_stream(side)
        self._restore(snap)
        torch.cuda.synchronize()
        dot = os.environ.get("RALF_GRAPH_DOT")   # diagnostics: the captured forward + backward graph as a DOT file (tools/graph_dot.py)
        gb = torch.cuda.CUDAGraph()
        rt = self.model.rt
        # Where the side-stream work (weight / bias gradients) runs: as parallel BRANCHES of the forward + backward graph (the hipGraph executor
        # places them: in practice behind the data-gradient chain), or as a SECOND graph replayed on the side stream, every item behind an event
        # node of the main graph (functional.ExternalEvent) -- concurrency by stream order.  Which one is faster depends on what the chain is
        # made of (whole step, B = 64: 14.6 -> 13.9 ms with the side graph, the HBM-bound backbone kernels leave the matrix pipes to the
        # weight gradients; encoder-decoder alone: 4.66 -> 4.81 ms, its chain of sub-256-workgroup kernels loses more to the sharing than the
        # tail gains).  RALF_SIDE_GRAPH=auto (default): capture both, time three replays of each on THIS model, batch shape and box, keep the
        # faster, free the other.  0 / 1 force a mode.  The staged (data-parallel) backward has a side graph per stage.
        want = os.environ.get("RALF_SIDE_GRAPH", "auto")
        can_defer = rt.overlap and rt.direct_grads and rt.n_side == 1 and want != "0" and not dot and ExternalEvent.supported()
        modes = [False] if (want == "0" or not can_defer or dot) else [True] if want == "1" else [False, True]
        cands, self.side_graph_ms = {}, {}

        def capture(graph, pool, body, mode):
            """one piece of the backward under capture (+ its side graph in side-graph mode) -> (side graph, its work list, body's result)"""
            rt.defer_side, rt._deferred = mode, []
            try:
                with torch.cuda.graph(graph, pool=pool, stream=cap, capture_error_mode=_CAPTURE_MODE):
                    res = body()
            finally:
                rt.defer_side = False
            gs, work = None, None
            if mode and rt._deferred:
                work, rt._deferred = rt._deferred, []   # (kept with the graphs: the closures hold operands in graph memory that must not be handed out again)
                gs = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gs, pool=graph.pool(), stream=rt._side[0], capture_error_mode=_CAPTURE_MODE):
                    for evs, fn, _keep in work:
                        for ev in evs:
                            ev.wait(torch.cuda.current_stream())
                        fn()
            return gs, work, res
        for mode in modes:
            ga = torch.cuda.CUDAGraph(keep_graph=True) if dot else torch.cuda.CUDAGraph()
            rt.weights_changed()   # every capture derives its own weight images (casts, packed layouts): a hit in a cache the OTHER capture filled would leave them out of this graph
            gs, side_work, loss = capture(ga, None, lambda: self._fwd_bwd(si, st), mode)
            gm = gs2 = work2 = None
            if self.staged:
                gm = torch.cuda.CUDAGraph()
                gs2, work2, _ = capture(gm, ga.pool(), self._bwd_rest, mode)
            cands[mode] = (ga, gm, gs, side_work, gs2, work2, loss, self.outputs)
            if len(modes) > 1:
                # (timed on the stream the replays will be launched from: which hardware queue it shares with the side stream is part of the answer)
                self.side_graph_ms[mode] = self._time_fwd_bwd([(ga, gs), (gm, gs2)], torch.cuda.current_stream())
        self.side_graph = min(self.side_graph_ms, key=self.side_graph_ms.get) if len(modes) > 1 else modes[0]
        ga, gm, gs, side_work, gs2, work2, self.loss, self.outputs = cands.pop(self.side_graph)
        if cands:   # the slower variant: its graph memory goes back to the allocator
            del cands, loss
            self._restore(snap)   # (the timed replays moved BatchNorm's running statistics)
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
        if dot:
            import ctypes
            hip = ctypes.CDLL("libamdhip64.so")
            hip.hipGraphDebugDotPrint.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint]
            rc = hip.hipGraphDebugDotPrint(ctypes.c_void_p(ga.raw_cuda_graph()), dot.encode(), int(os.environ.get("RALF_GRAPH_DOT_FLAGS", "1")))
            assert rc == 0, f"hipGraphDebugDotPrint: {rc}"
            ga.instantiate()
        with torch.cuda.graph(gb, pool=ga.pool(), stream=cap, capture_error_mode=_CAPTURE_MODE):
            self._update()
        self._graphs = (ga, gm, gb, gs, side_work, gs2, work2)
        # (capturing ran the host side of the step once more without executing kernels: put the host counters back too)
        self.opt.step_count = snap["step_count"]
        self.model.rt._wtoken += 1


# models that handed out optimizer groups, by the token their groups carry ("ralf_model"): lets an optimizer built from nothing but
# `model.optim_groups(...)` (train/train.py:217-223) find the model it trains
_MODELS: "weakref.WeakValueDictionary[int, torch.nn.Module]" = weakref.WeakValueDictionary()


def register_model(model) -> int:
    _MODELS[id(model)] = model
    return id(model)


class GraphedAdamW(torch.optim.Optimizer):
    """`torch.optim.AdamW`-shaped front of `TrainStep`, so that the reference's UNCHANGED loop body (train/train.py:432-454) runs the
    graph-replayed step.  Selected by one more Hydra override next to the generator's:

        optimizer._target_=ralf_amd.engine.GraphedAdamW  +optimizer.max_norm=${training.clip_max_norm}

    The loop calls, in order: `model.zero_grad()` -> `model.train_loss(inputs, targets)` -> `loss.backward()` -> `clip_grad_norm_(
    model.parameters(), max_norm)` -> `optimizer.step()`.  With this optimizer attached, `train_loss` (training mode, grad enabled) copies the
    batch into the step's static buffers and replays the captured forward + backward + (RCCL exchange) + global-norm clip + AdamW; the
    loss it returns is a leaf, so `loss.backward()` is a no-op on the model; parameters carry no `.grad` (the flat gradient buffer is
    private), so `clip_grad_norm_` has nothing to scale -- the clip already ran inside the graph with `max_norm`, which is why it is a
    constructor keyword here; `step()` only follows the learning-rate schedule.  `evaluate()` (eval mode, no_grad) takes the ordinary
    forward.  lr / weight_decay per group come from `model.optim_groups`, exactly as with torch's AdamW.

    The step runs on a library-owned stream and `model.preprocess` uploads the batch on a library-owned copy stream (the loop's `.to(rank)`
    then finds device tensors); with loss_lag = 0 the loop's stream is ordered after the step before train_loss returns, so the loss and the
    returned `outputs` can be read as usual.  With loss_lag = 1 the returned loss is a HOST scalar of the previous step and the loop's stream
    is not ordered after the step just enqueued: a caller that reads `outputs["logits"]` must `torch.cuda.synchronize()` first (the
    reference's loop does not read them).  Wrapping the model in the reference's DDPWrapper is fine: its own `zero_grad()` only drops `.grad`
    views the engine keeps a registry of, and its reducer never fires (the engine exchanges the flat gradient buffer itself)."""

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2, max_norm: Optional[float] = None,
                 loss_lag: int = 0, host_threads: Optional[int] = 16, **step_kw):
        if max_norm is None:
            # a silent default would train WITHOUT the reference's gradient clipping (training.clip_max_norm = 0.1): the loop's own
            # clip_grad_norm_ finds no .grad to scale
            raise TypeError("GraphedAdamW needs max_norm: the global-norm clip runs inside the captured step, not in the loop's clip_grad_norm_ "
                            "(+optimizer.max_norm=${training.clip_max_norm}; max_norm=0 switches clipping off explicitly)")
        groups = [dict(g) for g in params]
        tokens = {g.get("ralf_model") for g in groups}
        assert len(tokens) == 1 and None not in tokens, "GraphedAdamW takes the groups returned by a ralf_amd generator's optim_groups()"
        model = _MODELS.get(tokens.pop())
        assert model is not None, "the model these optimizer groups came from no longer exists"
        super().__init__(groups, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        for g in self.param_groups:
            assert g["lr"] is not None, "optim_groups(base_lr=...) sets every group's learning rate"
        self._base_lrs = [g["lr"] for g in self.param_groups]
        self._factor = 1.0
        # intra-op CPU threads of THIS process (the data loader's workers are processes of their own): torch defaults to one per core, and a
        # 128-thread OpenMP team on a busy 256-core host stalled the loop body for 50-80 ms every few iterations -- one descheduled thread
        # holds the team's barrier in the batch assembly copies (tools/loop_phases.py: 34-60 ms per iteration, 17.4 with 8 threads)
        if host_threads and torch.get_num_threads() > host_threads:
            torch.set_num_threads(int(host_threads))
        step_kw.setdefault("own_stream", True)
        self.engine = TrainStep(model, max_norm=max_norm, betas=betas, eps=eps, use_graph=True,
                                groups=[{"params": g["params"], "lr": g["lr"], "weight_decay": g["weight_decay"]} for g in self.param_groups], **step_kw)
        self._detached = False
        # loss_lag = 1 (opt-in): train_loss hands back the PREVIOUS step's loss as a host scalar, so the loop's `loss.cpu().item()`
        # (train/train.py:456) does not wait for the replay it has just enqueued and the host prepares batch i + 1 while the GPU runs step i.
        # The logged curve is shifted by one iteration; the training itself is unchanged.  0 (default): the loop's exact behaviour.
        self.loss_lag = int(loss_lag)
        assert self.loss_lag in (0, 1)
        self._prev = None
        model._engine = self

    def train_step(self, inputs, targets):
        self._follow_lr()   # a scheduler that stepped since the last update (per-epoch schedules step AFTER optimizer.step()) takes effect in THIS step
        loss = self.engine(inputs, targets)
        done = torch.cuda.Event()
        done.record(self.engine._run or torch.cuda.current_stream())
        self.engine.model._uploaded_batch_consumed(done)   # (the device buffer preprocess filled may be rewritten after this point)
        if not self._detached:   # the replays do not involve autograd: from here on the flat gradient buffer is the engine's own
            for g in self.param_groups:
                for p in g["params"]:
                    p.grad = None
            self._detached = True
        if self.loss_lag:
            run = self.engine._run or torch.cuda.current_stream()
            with torch.cuda.stream(run):
                host = torch.empty(1, dtype=torch.float32, pin_memory=True)
                host.copy_(loss.reshape(1), non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(run)
            prev, self._prev = self._prev, (host, ev)
            if prev is None:
                prev = (host, ev)          # first call: nothing older to report
            prev[1].synchronize()          # (the replay BEFORE the one just enqueued: long finished)
            return self.engine.outputs, {"nll_loss": prev[0].reshape(()).clone().requires_grad_()}
        self.engine.sync_caller()   # the loop reads the loss (and may read the logits) on its own stream
        return self.engine.outputs, {"nll_loss": loss.clone().requires_grad_()}

    def zero_grad(self, set_to_none: bool = True):   # the captured step zeroes its gradient buffer itself
        pass

    @torch.no_grad()
    def step(self, closure=None):
        """the update itself ran inside train_loss(); a scheduler that rewrote param_groups[i]["lr"] (MultiStepLR and the like scale every
        group by one factor) is followed through the device-resident factor the captured AdamW reads -- here and at the top of train_step"""
        self._follow_lr()
        return None

    @torch.no_grad()
    def _follow_lr(self):
        fs = [g["lr"] / b for g, b in zip(self.param_groups, self._base_lrs) if b]
        if fs and abs(fs[0] - self._factor) > 1e-12 * max(1.0, abs(self._factor)):
            assert all(abs(f - fs[0]) <= 1e-9 * max(1.0, abs(fs[0])) for f in fs), "per-group learning-rate schedules need a re-capture (not supported)"
            self._factor = fs[0]
            self.engine.opt.lr_scale.fill_(self._factor)
        return None


class GraphedDecode:
    """hipGraph replay of model.decode_tokens (encoder + the whole KV-cached autoregressive loop + fused
    mask/sample kernels): the eager loop is host-launch bound (~5k launches per batch).

    The image batch is the call's one big input (268 MB of fp32 at B = 256: 4.9 ms on the host link, a fifth of the replay).  With `gates` slices
    (RALF_DECODE_GATES, default 2; 0 = off) the captured backbone works through the batch slice by slice, every slice behind an event-wait NODE
    (functional.ExternalEvent), and `upload_image` -- called by model.sample() before its host-side preprocessing -- copies the page-locked host batch
    slice by slice straight into the graph's own image buffer on the library's copy stream, recording slice i's event behind its copy: the graph is
    launched without waiting for the copies, slice 0's kernels start when its images are there, and the link carries slice 1 meanwhile."""

    def __init__(self, model, cond_type: str, sampling_cfg, use_kv_cache: bool = True):
        self.model, self.cond_type, self.cfg, self.kv = model, cond_type, sampling_cfg, use_kv_cache
        self._graph = None
        self._static = None
        self._out = None
        n = int(os.environ.get("RALF_DECODE_GATES", "2"))
        self._gates = [ExternalEvent() for _ in range(n)] if (n > 1 and ExternalEvent.supported()) else None
        self._image_in_flight = False
        self.piped_calls = 0      # calls whose image went through upload_image (tests)
        # Which copy stream: the replay's image chain runs on a stream of the graph executable, and when the hardware queue behind it is also the copy stream's,
        # the chain queues up behind BOTH slices' copies (one graph instance in four: 31 ms instead of 28 at B = 256).  Two candidate copy streams, created
        # one after the other (neighbouring hardware queues); the first piped calls time one each (copy issue -> end of the replay), the faster one stays.
        self._cs_names, self._cs_pick, self._cs_ms, self._cs_probe = ("h2d", "h2d-b"), None, {}, None

    def upload_image(self, img) -> bool:
        """start the copy of a page-locked host image batch into the captured graph's image buffer (see the class comment); False: not applicable
        (nothing captured yet, no gates, another shape / dtype, pageable or device memory) -- the caller then hands the image to __call__ as usual"""
        if self._graph is None or self._gates is None or not torch.is_tensor(img) or img.is_cuda or not img.is_pinned():
            return False
        dst = self._static["enc"].get("image")
        if dst is None or dst.shape != img.shape or dst.dtype != img.dtype or dst.shape[0] < 2 * len(self._gates):
            return False
        from .helpers.task import pinned_copy_issued
        if self._cs_probe is not None:                  # the previous piped call was a timed one (it has long finished: sample() reads the tokens back)
            i, e0, e1 = self._cs_probe
            self._cs_probe = None
            if e1 is not None and e1.query():
                self._cs_ms.setdefault(i, []).append(e0.elapsed_time(e1))
        which = self._cs_pick
        if which is None:
            n = [len(self._cs_ms.get(i, [])) for i in range(len(self._cs_names))]
            if min(n) >= 2:                             # two timed calls each: keep the stream whose better call was faster
                best = [min(self._cs_ms[i]) for i in range(len(self._cs_names))]
                which = self._cs_pick = best.index(min(best))
            else:
                which = n.index(min(n))
        cs = ops.own_stream(self._cs_names[which], dst.device)
        cs.wait_stream(torch.cuda.current_stream())     # (the previous replay has read the buffer)
        if self._cs_pick is None:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record(cs)
            self._cs_probe = [which, e0, None]
        bounds = self.model.rt.input_bounds(dst.shape[0], len(self._gates))   # (nn.ResnetBackbone.body_features slices the batch the same way)
        with torch.cuda.stream(cs):
            for gi in range(len(bounds) - 1):
                dst[bounds[gi]:bounds[gi + 1]].copy_(img[bounds[gi]:bounds[gi + 1]], non_blocking=True)
                self._gates[gi].record_now(cs)
        ev = torch.cuda.Event()
        ev.record(cs)
        pinned_copy_issued(img, ev)                     # (possibly a cat_image staging buffer: not rewritten before these copies have read it)
        self._image_in_flight = True
        self.piped_calls += 1
        return True

    def static_image(self):
        return self._static["enc"]["image"]

    def __call__(self, enc_in: dict, cond_seq):
        rt = self.model.rt
        piped = False
        if self._graph is None:
            self._static = _clone_tree({"enc": enc_in, "seq": cond_seq})
            s = self._static
            side = ops.own_stream("capture")
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self.model.decode_tokens(s["enc"], s["seq"], self.cond_type, self.cfg, self.kv)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self._graph = torch.cuda.CUDAGraph()
            import gc
            gc.collect()
            was = gc.isenabled()
            gc.disable()   # (no collector pass inside a capture: TrainStep._capture)
            rt.input_gates = self._gates
            try:
                with torch.cuda.graph(self._graph, stream=ops.own_stream("capture"), capture_error_mode=_CAPTURE_MODE):
                    self._out = self.model.decode_tokens(s["enc"], s["seq"], self.cond_type, self.cfg, self.kv)
            finally:
                rt.input_gates = None
                if was:
                    gc.enable()
        else:
            # (upload_image: the image is on its way into the graph's buffer and the caller hands that buffer back -- _copy_tree skips a tensor that IS its
            #  destination; everything else travels on this stream)
            piped = self._image_in_flight and enc_in.get("image") is self._static["enc"].get("image")
            if self._image_in_flight and not piped:   # an upload nobody followed up on (the caller failed in between): this call's image overwrites it, in order
                for name in self._cs_names:
                    torch.cuda.current_stream().wait_stream(ops.own_stream(name, self._static["enc"]["image"].device))
            _copy_tree(self._static, {"enc": enc_in, "seq": cond_seq})
        if self._gates is not None and not piped:
            for g in self._gates:   # the image reached its buffer on THIS stream: the graph's gates open behind it
                g.record_now(torch.cuda.current_stream())
        self._image_in_flight = False
        self._graph.replay()
        if self._cs_probe is not None:
            if piped and self._cs_probe[2] is None:
                self._cs_probe[2] = torch.cuda.Event(enable_timing=True)
                self._cs_probe[2].record(torch.cuda.current_stream())
            elif self._cs_probe[2] is None:
                self._cs_probe = None
        return self._out
