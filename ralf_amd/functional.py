"""Autograd wiring of the HIP ops: every forward/backward below is a sequence of C-ABI calls
(ralf_amd/ops.py); torch only owns the tensors, the tape and gradient accumulation.

`Runtime` carries what the reference keeps implicitly in torch global state: compute dtype
(fp32 parity mode / bf16 throughput mode), train/eval flag for dropout + BatchNorm, the
device-resident dropout seed and a per-forward stream id counter for the counter-based RNG.
"""
from __future__ import annotations

import math
import os
import weakref
from typing import Optional

import torch
from torch.autograd import Function

from . import ops


class ExternalEvent:
    """a HIP event whose record and wait are EVENT NODES of the graphs under capture, so that a wait in one graph orders behind a record in
    ANOTHER graph (the side graph of engine.TrainStep).  torch.cuda.Event(external=True) is refused on ROCm builds of torch and the runtime
    rejects hipEventRecordWithFlags(.., hipEventRecordExternal) under capture, so the nodes are added by hand: the capture's graph and the
    stream's current dependency set (hipStreamGetCaptureInfo_v2), hipGraphAddEventRecordNode / hipGraphAddEventWaitNode behind them, and the
    new node as the stream's dependency set (hipStreamUpdateCaptureDependencies).  A wait binds to the record most recently ENQUEUED:
    replay the recording graph first (tools/lab/ext_event_probe.py: the waiting graph then runs beside the rest of the recording one)."""
    _hip = None

    def __init__(self):
        import ctypes
        vp = ctypes.c_void_p
        if ExternalEvent._hip is None:
            hip = ctypes.CDLL("libamdhip64.so")
            hip.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(vp), ctypes.c_uint]
            hip.hipEventDestroy.argtypes = [vp]
            hip.hipStreamGetCaptureInfo_v2.argtypes = [vp, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_ulonglong), ctypes.POINTER(vp),
                                                       ctypes.POINTER(ctypes.POINTER(vp)), ctypes.POINTER(ctypes.c_size_t)]
            hip.hipGraphAddEventRecordNode.argtypes = [ctypes.POINTER(vp), vp, ctypes.POINTER(vp), ctypes.c_size_t, vp]
            hip.hipGraphAddEventWaitNode.argtypes = [ctypes.POINTER(vp), vp, ctypes.POINTER(vp), ctypes.c_size_t, vp]
            hip.hipStreamUpdateCaptureDependencies.argtypes = [vp, ctypes.POINTER(vp), ctypes.c_size_t, ctypes.c_uint]
            ExternalEvent._hip = hip
        self._ev = vp()
        rc = ExternalEvent._hip.hipEventCreateWithFlags(ctypes.byref(self._ev), 2)   # hipEventDisableTiming
        assert rc == 0, f"hipEventCreateWithFlags: {rc}"

    def _node(self, stream, record: bool):
        import ctypes
        vp, hip = ctypes.c_void_p, ExternalEvent._hip
        status, cid, graph, deps, nd = ctypes.c_int(), ctypes.c_ulonglong(), vp(), ctypes.POINTER(vp)(), ctypes.c_size_t()
        rc = hip.hipStreamGetCaptureInfo_v2(stream.cuda_stream, ctypes.byref(status), ctypes.byref(cid), ctypes.byref(graph), ctypes.byref(deps), ctypes.byref(nd))
        assert rc == 0 and status.value == 1, f"ExternalEvent: the stream is not capturing (rc {rc}, status {status.value})"
        node = vp()
        rc = (hip.hipGraphAddEventRecordNode if record else hip.hipGraphAddEventWaitNode)(ctypes.byref(node), graph, deps, nd.value, self._ev)
        assert rc == 0, f"hipGraphAddEvent{'Record' if record else 'Wait'}Node: {rc}"
        rc = hip.hipStreamUpdateCaptureDependencies(stream.cuda_stream, (vp * 1)(node), 1, 1)   # hipStreamSetCaptureDependencies
        assert rc == 0, f"hipStreamUpdateCaptureDependencies: {rc}"

    _ok = None

    @classmethod
    def supported(cls) -> bool:
        """one probe per process: a record node and a wait node in two tiny captures (a runtime without the calls, or one that rejects the nodes,
        keeps the engine on graph branches)"""
        if cls._ok is None:
            try:
                s1, s2 = ops.own_stream("event-probe-a"), ops.own_stream("event-probe-b")
                x = torch.zeros(64, device="cuda")
                torch.cuda.synchronize()
                ev, g1, g2 = cls(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                with torch.cuda.graph(g1, stream=s1, capture_error_mode="thread_local"):
                    x.add_(1.0)
                    ev.record(torch.cuda.current_stream())
                with torch.cuda.graph(g2, pool=g1.pool(), stream=s2, capture_error_mode="thread_local"):
                    ev.wait(torch.cuda.current_stream())
                    x.mul_(2.0)
                with torch.cuda.stream(s1):
                    g1.replay()
                with torch.cuda.stream(s2):
                    g2.replay()
                torch.cuda.synchronize()
                cls._ok = float(x[0]) == 2.0
            except Exception:   # noqa: BLE001 -- whatever the runtime objects to: no side graph
                cls._ok = False
        return cls._ok

    def record(self, stream) -> "ExternalEvent":
        self._node(stream, True)
        return self

    def wait(self, stream):
        self._node(stream, False)

    def record_now(self, stream) -> "ExternalEvent":
        """an ordinary record on a stream that is NOT capturing (a copy stream): a wait NODE of a graph launched afterwards orders behind it"""
        import ctypes
        hip = ExternalEvent._hip
        hip.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        rc = hip.hipEventRecord(self._ev, ctypes.c_void_p(stream.cuda_stream))
        assert rc == 0, f"hipEventRecord: {rc}"
        return self

    # (no __del__: the event is referenced by nodes of captured graphs whose executables may be destroyed AFTER this object -- the garbage
    #  collector frees a TrainStep's members in no particular order, and hipGraphExecDestroy on a graph whose event is gone aborts the
    #  process.  A capture creates a few dozen of these; they live as long as the process.)


class Runtime:
    def __init__(self, dtype=torch.float32, seed: int = 0):
        self.dtype = dtype
        self.training = False
        self._seed_host = seed
        self.seed: Optional[torch.Tensor] = None  # int64[1] on device, advanced once per step
        self._call = 0
        self._lp: dict = {}
        self._shadow: dict = {}   # id(param) -> persistent low-precision view kept current by the fused optimizer
        self.direct_grads = False  # engine mode: parameter gradients are accumulated by the kernels straight into p.grad
        self._packs: dict = {}    # fragment-order weight copies by weight identity (packed())
        self._gviews: dict = {}   # id(param) -> (weakref, flat-buffer gradient view): survives `p.grad = None` (engine.GraphedAdamW)
        self._wtoken = 0          # bumped when weights are rewritten behind torch's version counters
        self._conv_table = None   # batched re-layout job table of the convolution weights (refresh_conv_shadows)
        self.overlap = False      # engine mode: weight / bias gradient kernels run on a side stream, off the data-gradient chain
        self._side: list = []
        self.n_side = int(os.environ.get("RALF_SIDE_STREAMS", "1"))
        self.side_policy = os.environ.get("RALF_SIDE_POLICY", "rr")
        self._side_rr = 0
        self._keep: list = []     # operands of side-stream work in flight (kept alive until join_side)
        # defer_side (set by the engine while it captures the forward + backward graph): the side-stream work is not issued but QUEUED, each
        # item behind EXTERNAL events recorded at its producers; the engine captures the queue as a graph of its own and replays it on the
        # side stream next to the main graph -- concurrency by stream order instead of by the hipGraph executor's branch placement
        self.defer_side = False
        self.cut_branches = os.environ.get("RALF_BRANCH_CUT", "1") != "0"   # side-graph capture: the constraint encoder's backward is side work too (branch_cut)
        self._deferred: list = []   # (events to wait for, closure, operands kept alive)
        self._wdep_streams: dict = {}
        self.cut_enabled = False  # engine mode (data parallel): split the backward at grad_cut() points
        self._cuts: list = []     # (tensor of the early graph, detached leaf the late graph continued from)
        # engine mode: independent sub-networks (constraint encoder, retrieved-layout branch) are issued on their own HIP
        # streams = parallel branches of the captured graph; autograd replays every backward node on its forward stream, so
        # the branches' backward chains of tiny launch-latency-bound kernels overlap the big kernels of the image branch too
        self.branches = False
        self.kv_ahead = os.environ.get("RALF_KV_AHEAD", "1") != "0"   # decoder: the memory's K/V projections ahead of the layers (Runtime.ahead)
        self._branch_streams: dict = {}
        self._branch_keep: list = []
        self._fans: list = []            # fan-out aliases of the current forward (check_fans)
        # engine mode (bf16): weight / bias gradients of the linear layers are COLLECTED during the backward and issued as grouped
        # launches (ops.wgrad_grouped: every output tile walks its whole reduction, no split-K slabs, no reduce kernels; one
        # column-sum launch for all bias gradients) whenever `group_tiles` output tiles are pending
        self.group_wgrads = False
        self.group_tiles = int(os.environ.get("RALF_WGRAD_GROUP_TILES", "100"))
        self.group_target_wgs = int(os.environ.get("RALF_WGRAD_GROUP_WGS", "2048"))
        self.fold_bgrads = os.environ.get("RALF_FOLD_BGRADS", "1") != "0"   # bias gradients from the grouped weight-gradient product's dy tiles
        self._wjobs: list = []
        self._bjobs: list = []
        self._wtiles = 0
        self._wdeps: dict = {}
        self._main_stream = None
        self._step_start = None
        # y = res + dropout(f(x)) outputs are tagged with their mask (p, call); the LayerNorm that consumes y then also emits the
        # MASKED gradient of y from its backward kernel, which the backward of f picks up instead of launching ralf_dropout
        self._drop_tags: dict = {}
        self._masked: dict = {}
        self.ln_dropout = True
        # z = relu(BN(x) (+ res)) outputs are tagged with what the BatchNorm backward reduces over (x, ReLU mask bits, mean); the convolution
        # that consumes z then masks dz and emits the reductions from its data-gradient epilogue (RalfGemmDesc.bnb_*), and the BatchNorm
        # backward that receives exactly that dz skips its own pass over dz and x
        self.bn_bwd_fused = os.environ.get("RALF_BN_BWD_FUSED", "1") != "0"
        self._bn_tags: dict = {}
        self._bn_stats: dict = {}
        self.bn_fused_hits = 0    # BatchNorm backwards that took their reductions from a data-gradient epilogue (tests / diagnostics)
        # training / full-sequence forward of the SHORT-sequence transformer layers (decoder: 5N <= 64 tokens; constraint encoder): one launch per
        # layer instead of 12 (7) (ops.tlayer_fwd; bf16, d = 256, 8 heads, ff = 1024).  The backward pass is the unfused one.
        self.fused_layers = os.environ.get("RALF_FUSED_LAYERS", "1") != "0"
        # ... and the feed-forward half of the LONG-sequence encoder layers (image encoder, 16 384 rows): LayerNorm + both products on 64-row strips
        self.fused_ffn = os.environ.get("RALF_FUSED_FFN", "1") != "0"
        self.fused_ffn_out = os.environ.get("RALF_FUSED_FFN_OUT", "1") != "0"   # ... with the attention's out-projection + residual in front
        self.fused_lnqkv = os.environ.get("RALF_FUSED_LNQKV", "1") != "0"       # ... and LayerNorm 1 + the q | k | v projection as one launch
        self.fused_ffn_bwd = os.environ.get("RALF_FUSED_FFN_BWD", "1") != "0"   # ... and the tail's four backward data-gradient launches as one
        # KV-cached decode step: out-projection + LayerNorm + feed-forward per layer as ONE launch on 32-row strips (ops.tlayer_tail).  OFF: measured
        # 36.2 vs 34.8 ms per B = 256 decode loop -- a workgroup streams all 1.15 MB of the layer's weights through one CU's L2 port (18.3 us per
        # launch, 26 k of its 39 k cycles in the feed-forward weight stream), the four launches it replaces spread them over the chip
        self.fused_decode_tail = os.environ.get("RALF_DECODE_TAIL", "0") == "1"
        self.fused_decode = True  # KV-cached decode step: LayerNorm + projections + attention per block in one launch (bf16, d = 256, 8 heads)
        # ... the WHOLE step (embedding, every layer, head) as one launch with a workgroup per sample (ops.decode_token, csrc/decode_token.hip)
        self.fused_decode_token = os.environ.get("RALF_DECODE_TOKEN", "1") != "0"
        # ... and the decode-space mask + token choice inside that launch (ralf_decode_token's s_* arguments; 0: a launch of its own, the same tokens)
        self.fused_decode_sample = os.environ.get("RALF_DECODE_SAMPLE", "1") != "0"
        # inference batches above this many images go through the backbone in slices (nn.ResnetBackbone.body_features); 0 = whole batch
        self.infer_chunk = int(os.environ.get("RALF_INFER_CHUNK", "0"))
        # sample(): opt-in (RALF_UPLOAD_LP=1) -- the image batch leaves the host in the compute dtype when that is bf16 (half the bytes over the host link; the
        # rounding the backbone's first kernel would apply, applied by the host copy into the staging buffer: the same tokens).  Off: at B = 256 the host's
        # conversion pass (0.5-3 ms, beside the task preprocessing it slows down) costs what the shorter copy saves (profiles/r06_sample_phases.txt)
        self.upload_lp = os.environ.get("RALF_UPLOAD_LP", "0") == "1"
        # engine.GraphedDecode: events the image batch's slices are gated on INSIDE the captured graph (nn.ResnetBackbone.body_features waits for
        # gate i in front of slice i's first kernel), so that the copy of slice i + 1 runs beside the backbone of slice i.  None: no gates
        self.input_gates = None
        self.fold_bn = True       # inference: eval-mode BatchNorm folded into the convolution epilogues (conv_bn_infer)
        # decode step: LayerNorm inside the few-row product that follows it (RalfGemmDesc.ln_*) and the four-wave split of the 256 x 256 x 1024
        # product (few_row_split) -- same arithmetic, other summation orders than the separate launches (off: their bits)
        # KV-cached loop of large batches as independent row slices on streams of their own (models/ralf.py: decode_tokens).  OFF: measured at B = 256
        # (tools/decode_once.py, same tokens): 31.6 ms for one chain, 35.7 for two slices, 40.5 for three, 46.6 for four -- the captured branches do
        # not overlap their launch-latency-bound kernels, they add cross-queue hand-offs
        # (round 5: every slice as a graph of ITS OWN replayed on its stream -- real concurrency, as the train step's side graph has -- is no better:
        #  34.6 ms for two slices, 53 for three or four: two chains of chip-wide 5-25 us kernels take turns, they do not overlap)
        self.decode_slices = int(os.environ.get("RALF_DECODE_SLICES", "1"))
        self.decode_ln_gemm = True
        self.decode_few_row_split = True
        self.conv_wgrad_direct = os.environ.get("RALF_CONV_WGRAD_DIRECT", "1") != "0"   # 3x3 / stride-1 weight gradients in the direct form (ops.conv3x3_wgrad)
        self.stem_direct = os.environ.get("RALF_STEM_DIRECT", "1") != "0"               # the 7x7 stem convolution in direct form (ops.stem7x7_fwd)
        self.fused_stem = os.environ.get("RALF_FUSED_STEM", "1") != "0"   # training: the stem's BatchNorm + ReLU + max-pool as one pass (StemBNReluPoolFn)

    @staticmethod
    def input_bounds(B: int, n: int):
        """slice boundaries of a gated image batch (engine.GraphedDecode.upload_image and nn.ResnetBackbone.body_features must agree).  Equal slices; with two
        slices RALF_DECODE_FIRST_SLICE sets the first one's share -- a shorter first slice (its copy is all that stands between the call and the backbone's
        first kernel) measured no better at B = 256: 28.1-28.4 ms at 0.5, 28.3-28.5 at 0.375, 28.6-28.7 at 0.25 (its kernels end before the rest has arrived)"""
        frac = float(os.environ.get("RALF_DECODE_FIRST_SLICE", "0.5"))
        first = min(B - (n - 1), max(1, int(round(B * frac / 8.0)) * 8 if B >= 16 else int(round(B * frac)))) if n == 2 else -(-B // n)
        rest = B - first
        out = [0, first]
        for i in range(1, n):
            out.append(first + (rest * i) // (n - 1))
        return out
    def to(self, device):
        if self.seed is None or self.seed.device != device:
            self.seed = torch.tensor([self._seed_host], dtype=torch.int64, device=device)
        return self

    def begin_step(self):
        """start of a forward: restart the stream-id counter (the device seed distinguishes steps)."""
        self._call = 0
        self._side_rr = 0
        self._main_stream = torch.cuda.current_stream() if torch.cuda.is_available() else None
        if self.branches and self._main_stream is not None:
            # branches depend on the step's inputs only: they wait for THIS point of the main stream, not for whatever the main
            # stream has been given by the time the branch is issued
            self._step_start = torch.cuda.Event()
            self._step_start.record(self._main_stream)
        self._branch_keep.clear()
        self._drop_tags.clear()
        self._masked.clear()
        self._bn_tags.clear()
        self._bn_stats.clear()

    def advance_seed(self):
        ops.counter_add_(self.seed, 0x9E3779B1)  # on-device: safe inside a captured graph

    def next_call(self) -> int:
        self._call += 1
        return self._call

    def drop_p(self, p: float) -> float:
        return p if (self.training and p > 0.0) else 0.0

    def grad_cut(self, x: torch.Tensor) -> torch.Tensor:
        """identity, unless the engine asked for a staged backward: then the autograd graph is cut here, so the
        gradients of everything computed AFTER this point are complete (and can be exchanged between ranks) before the
        backward of what came BEFORE it runs -- engine.TrainStep overlaps the two."""
        if not (self.cut_enabled and torch.is_grad_enabled() and x.requires_grad):
            return x
        leaf = x.detach().requires_grad_()
        self._cuts.append((x, leaf))
        return leaf

    def branch_cut(self, x: torch.Tensor) -> torch.Tensor:
        """identity, unless the engine is capturing with the side work deferred (defer_side): then the autograd graph is cut at this output of a
        sub-network branch whose backward feeds nothing but parameter gradients (the constraint encoder), and that backward is QUEUED like a
        weight gradient -- behind an event recorded where the gradient of x is complete -- and captured into the side graph: it runs beside the
        data-gradient chain instead of where the hipGraph executor places the branch (after the chain: a tail of ~70 tiny kernels)."""
        if not (self.defer_side and self.cut_branches and torch.is_grad_enabled() and x.requires_grad):
            return x
        leaf = x.detach().requires_grad_()

        def hook(g):
            ev = ExternalEvent().record(torch.cuda.current_stream())

            def run():   # (under capture of the side graph, current stream = the side stream: autograd forks to the branch's stream and joins back)
                ov, self.overlap = self.overlap, False   # the branch's own weight gradients run in line, on its stream
                try:
                    torch.autograd.backward([x], [g])
                    # autograd joins the streams it accumulated LEAF gradients on; the parameter gradients here are written straight into the flat
                    # buffer, so a stream it forked to may be left open: close every one that is part of this capture
                    here = torch.cuda.current_stream()
                    for st in list(self._branch_streams.values()) + [self._main_stream]:
                        if st is not None and st.cuda_stream != here.cuda_stream:
                            with torch.cuda.stream(st):
                                forked = torch.cuda.is_current_stream_capturing()
                            if forked:
                                here.wait_stream(st)
                    self.flush_wgrads()
                finally:
                    self.overlap = ov
            self._deferred.append(([ev], run, (x, g)))
            return g
        leaf.register_hook(hook)
        return leaf

    def register_shadow(self, w: torch.Tensor, view: torch.Tensor):
        """low-precision view kept current by the fused optimizer (which rewrites the master through its raw pointer, i.e.
        WITHOUT bumping w._version): any other in-place write to the master (load_state_dict, broadcast, re-init, EMA copy)
        does bump it, and lp() then re-casts the view."""
        self._shadow[id(w)] = [view, w._version, w.data_ptr(), weakref.ref(w)]

    def weights_changed(self):
        self._wtoken += 1

    # parameter-gradient kernels (dW, db) feed nothing but the optimizer: with `overlap` they are issued on a second
    # HIP stream (a parallel branch of the captured graph) so the many sub-256-workgroup launches of the
    # data-gradient chain share the CUs with them.  All side work is serialised on ONE stream, so accumulations
    # into the same gradient view keep their order.
    def side(self, fn, target, *operands):
        if not (self.overlap and self.direct_grads):
            return fn()
        if not self._side:
            self._side = self._make_side()
        if self.defer_side:
            self._deferred.append(([ExternalEvent().record(torch.cuda.current_stream())], fn, operands))
            return
        if self.side_policy == "rr" and len(self._side) == 1:   # call order: the same assignment in every run
            st = self._side[0]
        else:
            # several side streams: one gradient region -> always the same stream, so two non-atomic `accumulate` writes into the
            # same region (a weight used twice in a step, a deferred and a direct gradient of one W) stay ordered
            st = self._side[(target.data_ptr() >> 8) % len(self._side)]
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            fn()
        self._keep.append(operands)

    def _make_side(self):
        # (measured and not kept: the side stream on a priority queue -- whole step 22-25 ms -- or restricted to 32 / 64 / 128 CUs by
        #  hipExtStreamCreateWithCUMask -- 27-28 ms: either one is a FIFTH hardware queue, and the graph-replayed step leaves its 13.9 ms
        #  regime as soon as the process owns more than four)
        return [ops.own_stream(("side", i)) for i in range(self.n_side)]

    def fanout_alias(self, x: torch.Tensor) -> torch.Tensor:
        """an alias of x to hand to SEVERAL linear layers: their backward products are summed in the GEMM epilogue
        (LinearFn.backward: the first one's output is the buffer, the others accumulate into it) instead of by autograd's add
        kernels.  The alias is a node of its own in the autograd graph, so that sum is complete when it reaches x however many
        OTHER consumers x itself has (their gradients meet the sum in autograd's ordinary accumulation).  Only the linear layers
        that receive the returned tensor object take part (the mark travels as an attribute of that object)."""
        if not (torch.is_grad_enabled() and x.requires_grad):
            return x
        a = _AliasFn.apply(x)
        a._ralf_fan = [None, 0]   # [the shared d(x) buffer, backwards still to come]
        self._fans.append(a._ralf_fan)
        if len(self._fans) > 64:   # (forward-only callers never reach check_fans)
            del self._fans[:-64]
        return a

    def check_fans(self):
        """end of a COMPLETE backward (engine.TrainStep): every fan-out alias handed its summed d(x) to autograd.  The buffer is returned by the
        LAST of the fan's backwards (LinearFn.backward); a consumer whose backward never ran -- an output that does not reach the loss -- would
        leave the counter above zero and the gradient of the shared tensor silently dropped."""
        fans, self._fans = self._fans, []
        for f in fans:
            if f[1] != 0:
                raise RuntimeError(f"fanout_alias: {f[1]} of the linear layers that consumed the alias never ran their backward; d(x) of the shared "
                                   "tensor was not handed to autograd (a consumer's output does not reach the loss)")

    def tag_dropout(self, y: torch.Tensor, p: float, call: int):
        if p > 0.0 and self.ln_dropout:   # (called inside Function.forward, where grad mode is off: no grad-mode test here)
            self._drop_tags[y.data_ptr()] = (p, call)

    def dropout_tag(self, x: torch.Tensor):
        return self._drop_tags.get(x.data_ptr())

    def tag_bn_output(self, z: torch.Tensor, x2: torch.Tensor, mask, mean: torch.Tensor):
        if self.bn_bwd_fused and x2.shape[1] % 64 == 0:   # (called inside Function.forward: no grad-mode test here)
            self._bn_tags[z.data_ptr()] = (x2, mask, mean)

    def take_bn_tag(self, z: torch.Tensor):
        """the tag goes to the FIRST convolution that consumes z (a bottleneck's conv1 with fork=True, which sums the gradients of all
        other consumers in its own epilogue; conv2 / conv3 are sole consumers)"""
        return self._bn_tags.pop(z.data_ptr(), None)

    def offer_bn_stats(self, dz: torch.Tensor, partials: torch.Tensor):
        self._bn_stats[dz.data_ptr()] = (partials, dz)   # (dz itself is kept so that its address cannot be reused meanwhile)

    def bn_stats(self, dy: torch.Tensor, M: int, C: int):
        """the partial reductions a data-gradient GEMM wrote next to dy (already ReLU-masked), when dy IS that GEMM's output"""
        hit = self._bn_stats.pop(dy.data_ptr(), None)
        if hit is not None and hit[1].numel() == M * C and hit[0].shape[2] == C and dy.is_contiguous():
            self.bn_fused_hits += 1
            return hit[0]
        return None

    def offer_masked(self, dx: torch.Tensor, call: int, dx_masked: torch.Tensor):
        self._masked[dx.data_ptr()] = (call, dx_masked, dx)   # (dx itself is kept so that its address cannot be reused meanwhile)

    def masked_grad(self, dy: torch.Tensor, p: float, call: int) -> torch.Tensor:
        """dropout(dy) with the mask (p, call): taken from the LayerNorm backward that produced dy when it offered one"""
        hit = self._masked.pop(dy.data_ptr(), None)
        if hit is not None and hit[0] == call and hit[1].numel() == dy.numel():
            return hit[1].view(dy.shape)   # (the LayerNorm wrote it in ITS view of the rows: [B, S, d] against the GEMM's [B*S, d])
        return ops.dropout(dy, p, self.seed, call)

    def defer_wgrad(self, dy2d, x2d, into) -> bool:
        """queue dW (`into`, fp32 [n_out, n_in] view of the flat gradient buffer) += dy^T x for the next grouped launch; False when
        the shapes do not fit the grouped kernel (the caller then issues the product itself)"""
        rows, n_out = dy2d.shape
        n_in = x2d.shape[1]
        if not (self.group_wgrads and self.dtype == torch.bfloat16 and rows % 64 == 0 and n_out % 128 == 0 and n_in % 128 == 0
                and dy2d.stride(1) == 1 and x2d.stride(1) == 1 and dy2d.stride(0) % 8 == 0 and x2d.stride(0) % 8 == 0
                and dy2d.data_ptr() % 16 == 0 and x2d.data_ptr() % 16 == 0 and rows <= 65536):
            return False
        self._wjobs.append((dy2d, x2d, into.view(n_out, n_in), 1))
        self._wtiles += (n_out // 128) * (n_in // 128)
        self._note_producer()
        if self._wtiles >= self.group_tiles:
            self.flush_wgrads()
        return True

    def defer_bgrad(self, dy2d, into) -> bool:
        if not (self.group_wgrads and self.dtype == torch.bfloat16 and dy2d.shape[1] % 256 == 0 and dy2d.stride(1) == 1 and dy2d.stride(0) % 8 == 0
                and dy2d.data_ptr() % 16 == 0):
            return False
        self._bjobs.append((dy2d, into))
        self._note_producer()
        return True

    def _note_producer(self):
        """the queued operand was produced on the CURRENT stream (main, or a branch's stream during its backward): the grouped
        launch must wait for this point of THAT stream only -- waiting for whole streams would park the weight gradients of the
        big layers behind the chain of tiny kernels of an unrelated branch"""
        if self.overlap and self.direct_grads:
            st = torch.cuda.current_stream()
            if self.defer_side:   # one external record per stream at FLUSH time (a record node per queued operand would be hundreds per step)
                self._wdep_streams[st.cuda_stream] = st
                return
            ev = torch.cuda.Event()
            ev.record(st)
            self._wdeps[st.cuda_stream] = ev   # a later event on the same stream covers the earlier ones

    def flush_wgrads(self):
        """issue the queued weight / bias gradients (on the side stream when the engine overlaps them with the data-gradient chain)"""
        wj, bj = self._wjobs, self._bjobs
        if not (wj or bj):
            return
        # reduction splits: one workgroup sustains ~1 TFLOP/s on a 128x128 tile (measured), so a launch wants ~1000 workgroups;
        # every split walks at least 4096 rows (slab traffic; 2048: +0.05 ms per step), tiny reductions stay whole
        want = max(1, self.group_target_wgs // max(self._wtiles, 1))
        # a bias gradient whose dy is also the operand of a queued weight gradient rides on that product (ralf_wgrad_grouped: the column sums
        # of the dy tiles it reads anyway); the others keep the column-sum launch
        by_dy = {}
        if self.fold_bgrads:
            for i, (dy, into) in enumerate(bj):
                by_dy.setdefault((dy.data_ptr(), dy.shape[0], dy.shape[1], dy.stride(0)), []).append(i)
        taken = set()

        def bias_of(dy):
            hit = by_dy.get((dy.data_ptr(), dy.shape[0], dy.shape[1], dy.stride(0)))
            if not hit or dy.shape[1] % 256 != 0:
                return None
            i = hit.pop(0)
            if bj[i][1].data_ptr() % 16 != 0 or not bj[i][1].is_contiguous():
                hit.insert(0, i)
                return None
            taken.add(i)
            return bj[i][1]
        wj = [(dy, x, dw, max(1, min(want, dy.shape[0] // _GROUP_MIN_ROWS)), bias_of(dy)) for dy, x, dw, _ in wj]
        bj = [b for i, b in enumerate(bj) if i not in taken]
        self._wjobs, self._bjobs, self._wtiles = [], [], 0

        def run():
            if wj:
                ops.wgrad_grouped(wj)
            if bj:
                ops.colsum_grouped(bj)
        if self.overlap and self.direct_grads and self.defer_side:
            evs = []
            for st in self._wdep_streams.values():
                evs.append(ExternalEvent().record(st))
            self._wdep_streams = {}
            self._deferred.append((evs, run, (wj, bj)))
        elif self.overlap and self.direct_grads:
            if not self._side:
                self._side = self._make_side()
            st = self._side[0]
            for ev in self._wdeps.values():
                st.wait_event(ev)
            self._wdeps = {}
            with torch.cuda.stream(st):
                run()
            self._keep.append((wj, bj))
        else:
            run()

    def join_side(self):
        self.flush_wgrads()
        if self._side and self._keep:
            for st in self._side:
                torch.cuda.current_stream().wait_stream(st)
            self._keep.clear()

    def branch(self, name: str):
        """context manager: the enclosed forward work runs on the branch stream `name`, ordered only after the START of the step
        (begin_step): it must depend on nothing but the step's inputs and the weights.  Issue a branch LATE in program order
        (just before its output is consumed): on the device it still starts with the step, and autograd -- which runs the
        most recently recorded nodes first -- then starts its backward EARLY, beside the backward of the main branch, instead
        of as a serial tail.  Call join_branch(name, outputs...) before consuming its outputs on the current stream."""
        return _Branch(self, name)

    def join_branch(self, name: str, *outputs):
        st = self._branch_streams.get(name)
        if st is None:
            return
        torch.cuda.current_stream().wait_stream(st)
        # tensors that cross streams stay allocated until the end of the step: the caching allocator only orders reuse per stream
        self._branch_keep.extend(o for o in outputs if torch.is_tensor(o))

    def ahead(self, fns, name: str):
        """results of the independent products `fns` (all inputs ready on the current stream), issued NOW on the stream `name`: each result r carries
        r._ralf_ready, the event its consumer on the current stream has to wait for (ops.tlayer_fwd does, just before the launch that reads r).
        None unless the engine runs sub-networks on graph branches (then the caller computes them in line)."""
        if not (self.branches and self.kv_ahead and self._main_stream is not None and torch.is_grad_enabled()):
            return None
        st = self._branch_streams.get(name)
        if st is None:
            st = self._branch_streams[name] = ops.own_stream(("branch", name))
        st.wait_stream(torch.cuda.current_stream())
        outs = []
        with torch.cuda.stream(st):
            for fn in fns:
                r = fn()
                ev = torch.cuda.Event()
                ev.record(st)
                r._ralf_ready = ev
                outs.append(r)
        self._branch_keep.extend(outs)   # allocated on `st`, read on the current stream: alive until the streams meet again (join_all_branches)
        return outs

    def join_all_branches(self):
        """end of the backward: the branches' parameter-gradient kernels wrote into the flat buffer on their own streams"""
        cur = torch.cuda.current_stream()
        for st in self._branch_streams.values():
            cur.wait_stream(st)
        self._branch_keep.clear()

    def gview(self, p: torch.Tensor):
        """flat-buffer gradient view of a parameter when the engine owns the gradients (kernels accumulate into it
        and the autograd function returns None), else None (gradients are returned to autograd as tensors)."""
        if self.direct_grads and p.grad is not None and p.grad.dtype == torch.float32 and p.grad.is_contiguous():
            return p.grad
        if self.direct_grads and p.grad is None:
            hit = self._gviews.get(id(p))
            if hit is not None and hit[0]() is p:
                return hit[1]
        return None

    def register_grad_view(self, p: torch.Tensor, view: torch.Tensor):
        self._gviews[id(p)] = (weakref.ref(p), view)

    def refresh_conv_shadows(self, weights):
        """re-layout shadows ([Co][kh][kw][Ci] and [Ci][kh][kw][Co]) of all k>1 convolution weights in ONE launch when the
        masters changed (every optimizer step): ~40 separate few-microsecond launches otherwise.  `weights`: the conv
        weight Parameters (OIHW fp32).  lp(w, "ohwi"/"ikwo") then hits the cache."""
        ws = [w for w in weights if w.dim() == 4 and w.shape[2] > 1]
        if not ws:
            return
        sig = tuple((w.data_ptr(), w._version) for w in ws)
        st = self._conv_table
        if st is None or st["ptrs"] != tuple(w.data_ptr() for w in ws) or st["dtype"] != self.dtype:
            jobs, outs = [], []
            for w in ws:
                Co, Ci, kh, kw = w.shape
                cip = (Ci + 7) // 8 * 8
                o1 = torch.empty(Co, kh, kw, cip, dtype=self.dtype, device=w.device)
                o2 = torch.empty(Ci, kh, kw, Co, dtype=self.dtype, device=w.device)
                jobs.append((w.detach(), o1, o2))
                outs.append((o1, o2))
            table, n, blocks = ops.conv_relayout_table(jobs, ws[0].device)
            st = self._conv_table = {"ptrs": tuple(w.data_ptr() for w in ws), "dtype": self.dtype, "table": table, "n": n, "blocks": blocks,
                                     "outs": outs, "sig": None, "token": None}
        if st["sig"] == sig and st["token"] == self._wtoken:
            return
        ops.conv_relayout_batched(st["table"], st["n"], st["blocks"])
        for w, (o1, o2) in zip(ws, st["outs"]):
            stamp = ((w._version, self._wtoken), w.data_ptr())
            self._lp[(id(w), "ohwi", self.dtype)] = (stamp[0], stamp[1], o1, weakref.ref(w))
            self._lp[(id(w), "ikwo", self.dtype)] = (stamp[0], stamp[1], o2, weakref.ref(w))
        st["sig"], st["token"] = sig, self._wtoken

    def packed(self, weights, with_transposes: bool):
        """fragment-order copies (ops.tlayer_pack) of the row-major bf16 shadows of `weights`, followed -- with_transposes -- by the packed
        transposes in REVERSE order (what the strip-wise backward streams): one launch for all of them, repeated only when a master changed
        (every optimizer step in training, never for frozen / inference weights)"""
        key = tuple(id(w) for w in weights) + (bool(with_transposes),)
        stamp = tuple((w.data_ptr(), w._version) for w in weights) + (self._wtoken, self.dtype)
        hit = self._packs.get(key)
        if hit is not None and hit[0] == stamp and all(r() is w for r, w in zip(hit[2], weights)):
            return hit[1]
        mats = [self.lp(w) for w in weights]
        n = len(mats)
        out = ops.tlayer_pack(mats + (mats[::-1] if with_transposes else []), transpose=tuple(range(n, 2 * n)) if with_transposes else ())
        self._packs[key] = (stamp, out, [weakref.ref(w) for w in weights])
        return out

    # low-precision / re-laid-out shadows of fp32 master weights, refreshed when the master changes
    def lp(self, w: torch.Tensor, kind: str = "cast") -> torch.Tensor:
        if kind == "cast":
            if w.dtype == self.dtype:
                return w.detach()
            sh = self._shadow.get(id(w))
            if sh is not None and sh[3]() is w and sh[0].dtype == self.dtype:
                if sh[1] != w._version or sh[2] != w.data_ptr():   # master rewritten behind the optimizer: refresh the view
                    ops.cast_into(w.detach().contiguous().view(-1), sh[0].view(-1))
                    sh[1], sh[2] = w._version, w.data_ptr()
                    self._wtoken += 1
                return sh[0]
        key = (id(w), kind, self.dtype)
        hit = self._lp.get(key)
        # _wtoken covers rewrites through raw pointers (the fused optimizer), i.e. tensors the optimizer manages; a plain cast of anything else
        # (the FROZEN layout encoder's 18 matrices) follows torch's own version counter only -- they were re-cast after every optimizer step
        tok = self._wtoken if (kind != "cast" or id(w) in self._gviews) else 0
        if hit is not None and hit[3]() is w and hit[0] == (w._version, tok) and hit[1] == w.data_ptr():
            return hit[2]
        wd = w.detach()
        if kind == "cast":
            if hit is not None and hit[3]() is w and hit[2].shape == wd.shape and hit[2].dtype == self.dtype:
                t = ops.cast_into(wd.contiguous(), hit[2])   # refreshed IN PLACE: a captured graph that read the old copy sees the new values
            else:
                t = ops.cast(wd, self.dtype)
        else:
            Co, Ci, kh, kw = wd.shape
            if kind == "ohwi":      # [Co][kh][kw][Ci padded to 8]   (forward / weight-gradient layout)
                cip = (Ci + 7) // 8 * 8
                t = ops.permute4(wd, (Co, kh, kw, cip), (Ci * kh * kw, kw, 1, kh * kw), Ci, self.dtype)
            elif kind == "ikwo":    # [Ci][kh][kw][Co]               (data-gradient layout)
                t = ops.permute4(wd, (Ci, kh, kw, Co), (kh * kw, kw, 1, Ci * kh * kw), Co, self.dtype)
            else:
                raise ValueError(kind)
        self._lp[key] = ((w._version, tok), w.data_ptr(), t, weakref.ref(w))  # id() can be recycled: keep a weakref
        return t


class _Branch:
    def __init__(self, rt, name):
        self.rt, self.name, self.ctx = rt, name, None

    def __enter__(self):
        rt = self.rt
        if not (rt.branches and torch.cuda.is_available()):
            return self
        st = rt._branch_streams.get(self.name)
        if st is None:
            st = rt._branch_streams[self.name] = ops.own_stream(("branch", self.name))
        if rt._step_start is not None:
            st.wait_event(rt._step_start)      # inputs and weights of this step are ready there
        else:
            st.wait_stream(torch.cuda.current_stream())
        self.ctx = torch.cuda.stream(st)
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False


def _2d(x):
    return x.reshape(-1, x.shape[-1])


class _AliasFn(Function):
    """identity on a view of x (no kernel): Runtime.fanout_alias"""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g


_WGRAD_MIN_ROWS = int(os.environ.get("RALF_WGRAD_MIN_ROWS", "1024"))
_GROUP_MIN_ROWS = int(os.environ.get("RALF_GROUP_MIN_ROWS", "4096"))
_WGRAD_WG = int(os.environ.get("RALF_WGRAD_WG", "256"))   # tuning knob (re-measured after the lean gather loaders: 256 | 512 | 1024 = 15.89 | 15.97 | 16.07 ms per step)


def _splitk_for(out_rows: int, out_cols: int, red: int) -> int:
    tiles = ((out_rows + 127) // 128) * ((out_cols + 127) // 128)
    want = max(1, _WGRAD_WG // tiles)      # workgroups in flight; more splits only add slab traffic
    return max(1, min(want, red // _WGRAD_MIN_ROWS))  # each split reduces >= 1024 rows


def wgrad(dy2d, x2d, N, K, rows, into=None, rt=None):
    """dW[N,K] (fp32) = dy^T @ x, reduction over `rows` split across workgroups.  `into`: accumulate into this
    fp32 [N,K] view (flat gradient buffer) and return None."""
    if into is not None:
        if rt is not None and rt.defer_wgrad(dy2d, x2d, into):
            return None
        # (fp32 atomics straight into the buffer -- atomic=True -- were measured SLOWER than slabs + reduce:
        #  ~19 G atomics/s in L2 vs millions of adds per weight gradient: 29.6 -> 36.9 ms/step)
        def run():
            ops.gemm(dy2d, x2d, N, K, rows, a_kcontig=False, b_kcontig=False, out=into.view(N, K), accumulate=True, splitk=_splitk_for(N, K, rows))
        if rt is not None:
            rt.side(run, into, dy2d, x2d)
        else:
            run()
        return None
    return ops.gemm(dy2d, x2d, N, K, rows, a_kcontig=False, b_kcontig=False, out_dtype=torch.float32,
                    splitk=_splitk_for(N, K, rows))


def bgrad(dy2d, rows, N, into=None, rt=None):
    if into is not None:
        if rt is not None and rt.defer_bgrad(dy2d, into):
            return None
        if rt is not None:
            rt.side(lambda: ops.colsum(dy2d, rows, N, out=into), into, dy2d)
        else:
            ops.colsum(dy2d, rows, N, out=into)
        return None
    return ops.colsum(dy2d, rows, N)


# ----------------------------------------------------------------------------------------------
class LinearFn(Function):
    """y = x W[r0:r1]^T (+ b[r0:r1]) (+ res); `rows` selects a block of a packed weight (nn.MultiheadAttention
    in_proj for cross-attention) without materialising a sliced parameter; out_f32 -> fp32 logits."""

    @staticmethod
    def forward(ctx, x, W, b, res, rt, out_f32, rows, p=0.0):
        r0, r1 = rows if rows is not None else (0, W.shape[0])
        N, K = r1 - r0, W.shape[1]
        x2 = _2d(x)
        nrow = x2.shape[0]
        call = rt.next_call() if p > 0.0 else 0
        y = ops.gemm(x2, rt.lp(W)[r0:r1], nrow, N, K, bias=b.detach()[r0:r1] if b is not None else None,
                     res=_2d(res) if res is not None else None,
                     out_dtype=torch.float32 if out_f32 else None, drop_p=p, seed=rt.seed if p > 0.0 else None, call_id=call)
        ctx.save_for_backward(x2, W)
        ctx.bias, ctx.p, ctx.call = b, p, call
        ctx.rt, ctx.has_b, ctx.has_res, ctx.xshape, ctx.rows = rt, b is not None, res is not None, x.shape, (r0, r1)
        ctx.fan = getattr(x, "_ralf_fan", None)   # (Runtime.fanout_alias)
        if ctx.fan is not None:
            ctx.fan[1] += 1
        rt.tag_dropout(y, p, call)
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, W = ctx.saved_tensors
        rt = ctx.rt
        r0, r1 = ctx.rows
        N, K = r1 - r0, W.shape[1]
        full = N == W.shape[0]
        dy2 = _2d(dy.contiguous())
        if dy2.dtype != rt.dtype:
            dy2 = ops.cast(dy2, rt.dtype)
        if ctx.p > 0.0:  # y = res + drop(xW^T + b): regenerate the mask on the incoming gradient
            dy2 = rt.masked_grad(dy2, ctx.p, ctx.call)
        nrow = dy2.shape[0]
        dx = dW = db = None
        if ctx.needs_input_grad[0]:
            acc = ctx.fan
            if acc is not None and acc[0] is not None:
                # x feeds several linear layers (the memory of the 6 cross-attentions): the first backward to run handed its
                # product to autograd, the others add theirs into THAT buffer in the GEMM epilogue and return nothing
                ops.gemm(dy2, rt.lp(W)[r0:r1], nrow, K, N, b_kcontig=False, out=acc[0], accumulate=True)
            else:
                dx = ops.gemm(dy2, rt.lp(W)[r0:r1], nrow, K, N, b_kcontig=False)
                if acc is not None:
                    acc[0] = dx
            if acc is not None:
                # The buffer goes to autograd from the LAST of the fan's backwards to run: the consumer of d(x) may sit on another stream than
                # these products (Runtime.ahead), and autograd orders it behind the node that RETURNS the gradient only.
                acc[1] -= 1
                dx = acc[0] if acc[1] == 0 else None
            if dx is not None:
                dx = dx.view(ctx.xshape)
        if ctx.needs_input_grad[1]:
            gv = rt.gview(W)
            dW = wgrad(dy2, x2, N, K, nrow, gv[r0:r1] if gv is not None else None, rt)
            if dW is not None and not full:  # place the block into a full-size gradient (plumbing copy)
                g = torch.zeros(W.shape, dtype=torch.float32, device=dW.device)
                g[r0:r1] = dW
                dW = g
        if ctx.has_b and ctx.needs_input_grad[2]:
            gv = rt.gview(ctx.bias)
            db = bgrad(dy2, nrow, N, gv[r0:r1] if gv is not None else None, rt)
            if db is not None and not full:
                g = torch.zeros(W.shape[0], dtype=torch.float32, device=db.device)
                g[r0:r1] = db
                db = g
        dres = dy if ctx.has_res else None
        return dx, dW, db, dres, None, None, None, None


def linear(x, W, b=None, res=None, rt=None, out_f32=False, rows=None, p=0.0):
    """y = res + dropout_p(x W^T + b): bias, dropout and residual all live in the GEMM epilogue."""
    return LinearFn.apply(x, W, b, res, rt, out_f32, rows, p)


class FFNFn(Function):
    """y = res + drop_p(W2 drop_p(act(W1 x + b1)) + b2): two GEMMs whose epilogues carry bias, activation, both
    dropouts and the residual; hand-written backward (activation gradient fused into the data-gradient GEMM)."""

    @staticmethod
    def forward(ctx, x, W1, b1, W2, b2, res, act, p, rt):
        x2 = _2d(x)
        rows, K = x2.shape
        Hd, N = W1.shape[0], W2.shape[0]
        z = torch.empty(rows, Hd, dtype=rt.dtype, device=x.device) if act == "gelu" else None
        c1 = rt.next_call() if p > 0.0 else 0
        c2 = rt.next_call() if p > 0.0 else 0
        sd = rt.seed if p > 0.0 else None
        h = ops.gemm(x2, rt.lp(W1), rows, Hd, K, bias=b1.detach(), act=act, out2=z, drop_p=p, seed=sd, call_id=c1)
        y = ops.gemm(h, rt.lp(W2), rows, N, Hd, bias=b2.detach(), res=_2d(res) if res is not None else None, drop_p=p, seed=sd, call_id=c2)
        ctx.save_for_backward(x2, W1, W2, h, z)
        ctx.b1, ctx.b2 = b1, b2
        ctx.rt, ctx.act, ctx.p, ctx.has_res, ctx.xshape, ctx.c2 = rt, act, p, res is not None, x.shape, c2
        rt.tag_dropout(y, p, c2)
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, W1, W2, h, z = ctx.saved_tensors
        rt = ctx.rt
        dy = dy.contiguous()
        dy2 = _2d(dy)
        if ctx.p > 0.0:
            dy2 = rt.masked_grad(dy2, ctx.p, ctx.c2)   # mask of the output dropout
        rows, K = x2.shape
        Hd, N = W1.shape[0], W2.shape[0]
        if ctx.act == "gelu":
            dz = ops.gemm(dy2, rt.lp(W2), rows, Hd, N, b_kcontig=False, aux=z, aux_mode="gelu_grad")
        else:  # relu (+ dropout): h > 0 <=> pre-activation > 0 and kept
            dz = ops.gemm(dy2, rt.lp(W2), rows, Hd, N, b_kcontig=False, aux=h, aux_mode="relu_mask", aux_scale=1.0 / (1.0 - ctx.p))
        dW2 = wgrad(dy2, h, N, Hd, rows, rt.gview(W2), rt)
        db2 = bgrad(dy2, rows, N, rt.gview(ctx.b2), rt)
        dW1 = wgrad(dz, x2, Hd, K, rows, rt.gview(W1), rt)
        db1 = bgrad(dz, rows, Hd, rt.gview(ctx.b1), rt)
        dx = ops.gemm(dz, rt.lp(W1), rows, K, Hd, b_kcontig=False).view(ctx.xshape) if ctx.needs_input_grad[0] else None
        return dx, dW1, db1, dW2, db2, (dy if ctx.has_res else None), None, None, None


def _ln_backward(ctx, dy, skip):
    """shared backward of LayerNormFn / LayerNormSkipFn; when the normalised tensor came out of `res + dropout(..)` (tagged), the
    kernel also writes the masked gradient that the producer's backward will ask for"""
    x, g, mean, rstd = ctx.saved_tensors
    rt = ctx.rt
    gg, gb = rt.gview(g), rt.gview(ctx.beta)
    drop = (ctx.tag[0], rt.seed, ctx.tag[1]) if ctx.tag is not None else None
    into = (gg, gb) if (gg is not None and gb is not None) else None
    out = ops.layernorm_bwd(dy.contiguous(), x, g.detach(), mean, rstd, need_wgrad=True, into=into, skip=skip, drop=drop)
    dx = out[0]
    if drop is not None:
        rt.offer_masked(dx, ctx.tag[1], out[3])
    if into is not None:
        return dx, None, None
    return dx, out[1], out[2]


class LayerNormFn(Function):
    @staticmethod
    def forward(ctx, x, g, b, rt):
        xc = x.contiguous()
        y, mean, rstd = ops.layernorm_fwd(xc, g.detach(), b.detach())
        ctx.save_for_backward(xc, g, mean, rstd)
        ctx.rt, ctx.beta, ctx.tag = rt, b, rt.dropout_tag(xc)
        return y

    @staticmethod
    def backward(ctx, dy):
        dx, dg, db = _ln_backward(ctx, dy, None)
        return dx, dg, db, None


def layer_norm(x, g, b, rt):
    return LayerNormFn.apply(x, g, b, rt)


class LayerNormSkipFn(Function):
    """(LN(x), x): the pre-norm residual pattern.  The gradient of the skip output is added inside the
    LayerNorm backward kernel instead of by a separate accumulate pass."""

    @staticmethod
    def forward(ctx, x, g, b, rt):
        xc = x.contiguous()
        y, mean, rstd = ops.layernorm_fwd(xc, g.detach(), b.detach())
        ctx.save_for_backward(xc, g, mean, rstd)
        ctx.rt, ctx.beta, ctx.tag = rt, b, rt.dropout_tag(xc)
        return y, xc.view_as(xc)

    @staticmethod
    def backward(ctx, dy, dskip):
        skip = dskip.contiguous() if dskip is not None else None
        dx, dg, db = _ln_backward(ctx, dy, skip)
        return dx, dg, db, None


def layer_norm_skip(x, g, b, rt):
    return LayerNormSkipFn.apply(x, g, b, rt)


class DropAddFn(Function):
    """y = res + dropout(x) (plain add when p == 0)."""

    @staticmethod
    def forward(ctx, x, res, p, rt):
        call = rt.next_call() if p > 0.0 else 0
        ctx.rt, ctx.p, ctx.call, ctx.has_res = rt, p, call, res is not None
        return ops.dropout(x.contiguous(), p, rt.seed, call, res.contiguous() if res is not None else None)

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        dx = ops.dropout(dy, ctx.p, ctx.rt.seed, ctx.call) if ctx.p > 0.0 else dy
        return dx, (dy if ctx.has_res else None), None, None


def drop_add(x, res, p, rt):
    if p == 0.0 and res is None:
        return x
    return DropAddFn.apply(x, res, p, rt)


class ConcatRowsFn(Function):
    """cat(dim=1) of [B, S_i, d] tensors, with a learned scalar table[idx_i] added to source i where idx_i is not None (the
    flag embedding task_emb = nn.Embedding(2, 1), retrieval_augmented_autoreg.py:1022-1028): one launch forward, one launch
    backward (contiguous gradient pieces + the scalars' gradients)."""

    @staticmethod
    def forward(ctx, table, idxs, rt, *srcs):
        srcs = [t.contiguous() for t in srcs]
        scal = None
        if table is not None:
            tv = table.detach().view(-1)
            scal = [tv[i:i + 1] if i is not None else None for i in idxs]
        ctx.shapes, ctx.idxs, ctx.rt, ctx.table = [tuple(t.shape) for t in srcs], idxs, rt, table
        ctx.dtype = srcs[0].dtype
        return ops.concat_rows(srcs, scal)

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        pieces = [torch.empty(shp, dtype=ctx.dtype, device=dy.device) for shp in ctx.shapes]
        dtab, dsc = None, None
        if ctx.table is not None:
            gv = ctx.rt.gview(ctx.table) if ctx.rt is not None else None
            acc = gv.view(-1) if gv is not None else torch.zeros(ctx.table.numel(), dtype=torch.float32, device=dy.device)
            dsc = [acc[i:i + 1] if i is not None else None for i in ctx.idxs]
            dtab = None if gv is not None else acc.view_as(ctx.table)
        ops.concat_rows(pieces, out=dy, backward=True, dscalars=dsc)
        return (dtab, None, None) + tuple(pieces)


def concat_rows(srcs, rt, table=None, idxs=None):
    return ConcatRowsFn.apply(table, idxs if idxs is not None else [None] * len(srcs), rt, *srcs)


class Fork2Fn(Function):
    """(x, x): two aliases of a tensor with two consumers; their gradients are summed by the library's own kernel instead of autograd's
    accumulation add"""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, g1, g2):
        if g1 is None or g2 is None:
            return g1 if g2 is None else g2
        return ops.dropout(g1.contiguous(), 0.0, None, 0, g2.contiguous())


def fork2(x):
    if not (torch.is_grad_enabled() and x.requires_grad):
        return x, x
    return Fork2Fn.apply(x)


class ConcatColsFn(Function):
    """cat(dim=-1) of two [..., Ca] / [..., Cb] tensors (the FPN's channel concat, common/image.py:108): two strided copies by the
    library's own kernel each way"""

    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        Ca, Cb = a.shape[-1], b.shape[-1]
        rows = a.numel() // Ca
        out = torch.empty(*a.shape[:-1], Ca + Cb, dtype=a.dtype, device=a.device)
        ops.copy2d(a, out, rows, Ca, Ca, Ca + Cb)
        ops.copy2d(b, out.view(-1)[Ca:], rows, Cb, Cb, Ca + Cb)
        ctx.dims = (Ca, Cb, rows, tuple(a.shape), tuple(b.shape))
        return out

    @staticmethod
    def backward(ctx, g):
        Ca, Cb, rows, sa, sb = ctx.dims
        g = g.contiguous()
        ga = torch.empty(sa, dtype=g.dtype, device=g.device)
        gb = torch.empty(sb, dtype=g.dtype, device=g.device)
        ops.copy2d(g, ga, rows, Ca, Ca + Cb, Ca)
        ops.copy2d(g.view(-1)[Ca:], gb, rows, Cb, Ca + Cb, Cb)
        return ga, gb


class ScalePEDropFn(Function):
    """dropout_p(x * s + pe[:S]) on [B, S, d]: PositionalEncoding1d (common/positional_encoding.py:92-107) of the retrieved features"""

    @staticmethod
    def forward(ctx, x, pe, scale, p, rt):
        call = rt.next_call() if p > 0.0 else 0
        ctx.cfg = (scale, p, call, rt)
        return ops.scale_pe_dropout(x.contiguous(), pe, x.shape[-2], scale, p, rt.seed, call)

    @staticmethod
    def backward(ctx, dy):
        scale, p, call, rt = ctx.cfg
        return ops.scale_pe_dropout(dy.contiguous(), None, 1, scale, p, rt.seed, call), None, None, None, None


class EmbedFn(Function):
    """emb(idx) * sqrt(d) + pe[:S]  (nn.Embedding -> PositionalEncoding1d)."""

    @staticmethod
    def forward(ctx, idx, W, pe, rt):
        S, d = idx.shape[-1], W.shape[1]
        ctx.save_for_backward(idx)
        ctx.vocab, ctx.scale, ctx.rt, ctx.W = W.shape[0], math.sqrt(d), rt, W
        return ops.embed_fwd(idx.contiguous(), W.detach(), pe, S, ctx.scale, rt.dtype)

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        gv = ctx.rt.gview(ctx.W)
        return None, ops.embed_bwd(idx.contiguous(), dy.contiguous(), ctx.vocab, ctx.scale, into=gv), None, None


class AddScalarFn(Function):
    """x + s[i] with s a learned scalar table (task_emb = nn.Embedding(2, 1))."""

    @staticmethod
    def forward(ctx, x, table, i, rt=None):
        ctx.i, ctx.n, ctx.rt, ctx.table = i, table.shape[0], rt, table
        return ops.add_scalar(x.contiguous(), table.detach().view(-1)[i:i + 1].contiguous())

    @staticmethod
    def backward(ctx, dy):
        gv = ctx.rt.gview(ctx.table) if ctx.rt is not None else None
        if gv is not None:
            ops.sum_all(dy.contiguous(), out=gv.view(-1)[ctx.i:ctx.i + 1])
            return dy, None, None, None
        g = torch.zeros(ctx.n, 1, dtype=torch.float32, device=dy.device)
        g[ctx.i] = ops.sum_all(dy.contiguous())  # slice assignment: plumbing
        return dy, g, None, None


class AttnFn(Function):
    """softmax(scale Q K^T + mask) V on packed projections.
    mode 'self':  a = qkv [B,S,3*H*dh];            mode 'cross': a = q [B,Sq,H*dh], b = kv [B,Sk,2*H*dh]"""

    @staticmethod
    def forward(ctx, a, b, H, dh, causal, kpm, p, rt):
        a = a.contiguous()
        inner = H * dh
        if b is None:
            B, Sq, _ = a.shape
            Sk, q, k, v, offs = Sq, a, a, a, (0, inner, 2 * inner)
        else:
            b = b.contiguous()
            B, Sq, _ = a.shape
            Sk, q, k, v, offs = b.shape[1], a, b, b, (0, 0, inner)
        call = rt.next_call() if p > 0.0 else 0
        o, lse = ops.attention_fwd(q, k, v, B, H, Sq, Sk, dh, *offs, causal=causal, kpm=kpm, p_drop=p, seed=rt.seed, call_id=call)
        ctx.save_for_backward(a, b, o, lse, kpm)
        ctx.cfg = (B, H, Sq, Sk, dh, offs, causal, p, call, rt)
        return o

    @staticmethod
    def backward(ctx, do):
        a, b, o, lse, kpm = ctx.saved_tensors
        B, H, Sq, Sk, dh, offs, causal, p, call, rt = ctx.cfg
        do = do.contiguous()
        da = torch.empty_like(a)
        if b is None:
            ops.attention_bwd(do, a, a, a, o, lse, da, da, da, B, H, Sq, Sk, dh, *offs, *offs, causal=causal, kpm=kpm, p_drop=p, seed=rt.seed, call_id=call)
            return da, None, None, None, None, None, None, None
        db = torch.empty_like(b)
        ops.attention_bwd(do, a, b, b, o, lse, da, db, db, B, H, Sq, Sk, dh, *offs, *offs, causal=causal, kpm=kpm, p_drop=p, seed=rt.seed, call_id=call)
        return da, db, None, None, None, None, None, None


class XentFn(Function):
    """nn.CrossEntropyLoss(label_smoothing, ignore_index) on fp32 logits; dlogits produced by the same pass."""

    @staticmethod
    def forward(ctx, logits, target, ignore_index, eps, rt):
        cl, dl = ops.xent(logits.contiguous(), target.contiguous(), ignore_index, eps, rt.dtype)
        ctx.save_for_backward(dl)
        return cl[1]

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        # scalar chain rule on the device by the library's own kernel (the logits are fp32, so is their gradient)
        return ops.scale_dev(dl, g.detach().float().reshape(1).contiguous()), None, None, None, None


# ----------------------------------------------------------------------------------------------
# cross-attention K/V projections of ALL decoder layers over the same memory in one product
# ----------------------------------------------------------------------------------------------
class CrossKVPlan:
    """The 6 decoder layers of BaseDecoder project the SAME memory with their own packed in_proj rows [d:3d]
    (common/common.py:25-34: nn.TransformerDecoderLayer.multihead_attn).  When the engine keeps all parameters in one flat
    buffer, consecutive layers sit a constant stride apart, so the 6 products are ONE batched GEMM launch (forward, weight
    gradient) and -- through a K range chained over the 6 weight blocks -- ONE data-gradient GEMM instead of 6 products and 5
    full-size accumulations of d(memory).  `make` returns None when the layout does not allow it (plain torch parameters:
    the per-layer path of MHAParams.cross_attn is used)."""

    def __init__(self, L, d, sW, sWg, sB, sBg):
        self.L, self.d, self.sW, self.sWg, self.sB, self.sBg = L, d, sW, sWg, sB, sBg
        self.dkv = None   # gradient of the stacked K/V projections, filled slice by slice by the layers' attention backward

    # OFF by default -- measured on MI355X (B = 64, same run): encoder-decoder step 6.92 ms per-layer vs 7.52 ms stacked.  The
    # stacked K/V tensor (209 MB) and its gradient are produced by one launch and consumed layer by layer much later, i.e. from
    # HBM instead of the 256 MB Infinity Cache the per-layer products (35 MB each, consumed at once) live in.  Kept for tests
    # and for re-measuring (RALF_STACKED_KV=1).
    enabled = os.environ.get("RALF_STACKED_KV", "0") == "1"

    @staticmethod
    def make(attns, rt):
        if not (CrossKVPlan.enabled and rt.direct_grads) or len(attns) < 2:
            return None
        Ws, bs = [a.in_proj_weight for a in attns], [a.in_proj_bias for a in attns]
        if any(rt.gview(w) is None for w in Ws) or any(rt.gview(b) is None for b in bs):
            return None

        def stride(ts):
            e = ts[0].element_size()
            ds = {(ts[i + 1].data_ptr() - ts[i].data_ptr()) for i in range(len(ts) - 1)}
            return (ds.pop() // e) if len(ds) == 1 and next(iter(ds)) % (16) == 0 and next(iter(ds)) > 0 else None

        lw = [rt.lp(w) for w in Ws]
        st = [stride(lw), stride([rt.gview(w) for w in Ws]), stride([b.detach() for b in bs]), stride([rt.gview(b) for b in bs])]
        if any(x is None for x in st) or len({tuple(w.shape) for w in Ws}) != 1:
            return None
        return CrossKVPlan(len(attns), Ws[0].shape[1], *st)


class CrossKVFn(Function):
    """kv_all [B*M, L*2d] = memory @ [W_0[d:3d]; ...; W_{L-1}[d:3d]]^T + biases  (layer l owns columns [l*2d, (l+1)*2d))"""

    @staticmethod
    def forward(ctx, mem, plan, rt, W0, b0, *rest):
        L, d = plan.L, plan.d
        mem2 = _2d(mem.contiguous())
        rows = mem2.shape[0]
        out = torch.empty(rows, L * 2 * d, dtype=rt.dtype, device=mem.device)
        ops.gemm(mem2, rt.lp(W0)[d:], rows, 2 * d, d, bias=b0.detach()[d:], batch=(L, 1), sA=(0, 0), sB=(plan.sW, 0), sC=(2 * d, 0),
                 ldc=L * 2 * d, out=out, sBias0=plan.sB)
        ctx.save_for_backward(mem2, W0)
        ctx.plan, ctx.rt, ctx.b0, ctx.mshape = plan, rt, b0, mem.shape
        plan.dkv = None
        return out.view(*mem.shape[:-1], L * 2 * d)

    @staticmethod
    def backward(ctx, dkv):
        mem2, W0 = ctx.saved_tensors
        plan, rt = ctx.plan, ctx.rt
        L, d = plan.L, plan.d
        rows = mem2.shape[0]
        dkv2 = _2d(dkv.contiguous())
        K = L * 2 * d
        if rt.dtype == torch.bfloat16:   # one product over the chained K range
            dmem = ops.gemm(dkv2, rt.lp(W0)[d:], rows, d, K, b_kcontig=False, kseg=2 * d, sBk=plan.sW - 2 * d * d)
        else:                            # parity mode: per layer, accumulated in the GEMM epilogue
            dmem = torch.empty(rows, d, dtype=rt.dtype, device=dkv2.device)
            w0 = rt.lp(W0)
            for l in range(L):
                wl = torch.as_strided(w0, (3 * d, d), (d, 1), w0.storage_offset() + l * plan.sW)[d:]
                ops.gemm(dkv2[:, l * 2 * d:], wl, rows, d, 2 * d, lda=K, b_kcontig=False, out=dmem, accumulate=l > 0)
        gW, gb = rt.gview(W0), rt.gview(ctx.b0)

        def wg():   # dW_l[d:3d] += dkv_l^T mem for all layers at once (batched over l, split over the rows)
            ops.gemm(dkv2, mem2, 2 * d, d, rows, a_kcontig=False, b_kcontig=False, lda=K, batch=(L, 1), sA=(2 * d, 0), sB=(0, 0), sC=(plan.sWg, 0),
                     out=gW[d:], accumulate=True, splitk=_splitk_for(2 * d * L, d, rows))
            tmp = ops.colsum(dkv2, rows, K)
            ops.copy2d_acc(tmp, gb[d:], L, 2 * d, 2 * d, plan.sBg)
        rt.side(wg, gW, dkv2, mem2)
        return (dmem.view(ctx.mshape), None, None, None, None) + (None,) * (2 * L - 2)


class AttnCrossSliceFn(Function):
    """cross-attention of decoder layer `li` reading its K/V from the stacked projection kv_all; the backward writes dK/dV into
    the shared gradient buffer of kv_all (plan.dkv) and only layer 0 -- whose backward runs last -- hands that buffer to
    autograd, so no per-layer zero-filled full-size gradients are summed."""

    @staticmethod
    def forward(ctx, q, kv_all, li, plan, H, dh, p, rt):
        q = q.contiguous()
        B, Sq, _ = q.shape
        Sk = kv_all.shape[1]
        d = plan.d
        call = rt.next_call() if p > 0.0 else 0
        o, lse = ops.attention_fwd(q, kv_all, kv_all, B, H, Sq, Sk, dh, 0, li * 2 * d, li * 2 * d + d, p_drop=p, seed=rt.seed, call_id=call)
        ctx.save_for_backward(q, kv_all, o, lse)
        ctx.cfg = (B, H, Sq, Sk, dh, li, plan, p, call, rt)
        return o

    @staticmethod
    def backward(ctx, do):
        q, kv_all, o, lse = ctx.saved_tensors
        B, H, Sq, Sk, dh, li, plan, p, call, rt = ctx.cfg
        d = plan.d
        if plan.dkv is None:
            plan.dkv = torch.empty_like(kv_all)
        dq = torch.empty_like(q)
        ops.attention_bwd(do.contiguous(), q, kv_all, kv_all, o, lse, dq, plan.dkv, plan.dkv, B, H, Sq, Sk, dh, 0, li * 2 * d, li * 2 * d + d,
                          0, li * 2 * d, li * 2 * d + d, p_drop=p, seed=rt.seed, call_id=call)
        out = plan.dkv if li == 0 else None
        if li == 0:
            plan.dkv = None
        return dq, out, None, None, None, None, None, None


# ----------------------------------------------------------------------------------------------
# whole pre-norm transformer layer, forward in one launch (ops.tlayer_fwd)
# ----------------------------------------------------------------------------------------------
class _Ctx:
    """stand-in for an autograd ctx: TLayerFn.backward runs the unfused Functions' OWN backward code on the tensors the fused forward
    saved, so the two paths cannot drift apart"""

    def __init__(self, saved, needs=None, **kw):
        self.saved_tensors, self.needs_input_grad = saved, needs
        self.__dict__.update(kw)


def tlayer_supported(x, rt, d, nhead, dim_ff, allow_long=False) -> bool:
    """allow_long (decoder layers): sequences of more than 64 tokens take the strip-wise kernels for their row-wise parts when the rows of all
    samples split into 64-row strips (TLayerFn forward: _tlayer_fwd_long)"""
    if not (rt.fused_layers and rt.dtype == torch.bfloat16 and x.is_cuda and x.dim() == 3 and d == 256 and nhead == 8 and dim_ff == 1024 and x.shape[2] == d):
        return False
    return x.shape[1] <= ops.TLAYER_MAX_ROWS or (allow_long and rt.fused_ffn and _FUSED_LONG and (x.shape[0] * x.shape[1]) % 64 == 0)


_FUSED_LONG = os.environ.get("RALF_FUSED_LONG", "1") != "0"   # A/B runs: long decoder layers per operation


def _tlayer_fwd_long(x, W, rowmajor, kv, kpm, causal, p, seed, calls):
    """a decoder layer on MORE than 64 tokens per sample: the same tensors as ops.tlayer_fwd writes, from [LayerNorm 1 + q|k|v] on strips,
    attention, the self out-projection / LayerNorm 2 / q projection per operation, cross-attention, [out-projection 2 + LayerNorm 3 + FFN] on
    strips.  rowmajor = (self out_proj weight, cross q-projection weight) as bf16 row-major matrices (these two stay ralf_gemm)."""
    B, S, d = x.shape
    rows, H = B * S, 8
    sd = seed if p > 0.0 else None
    t = ops.tlayer_lnqkv(x, {"ln1": W["ln1"], "sa_in": W["sa_in"]})
    t["o1"], t["lse1"] = ops.attention_fwd(t["qkv"], t["qkv"], t["qkv"], B, H, S, S, d // H, 0, d, 2 * d, causal=causal, kpm=kpm, p_drop=p, seed=seed, call_id=calls[0])
    t["x1"] = ops.gemm(t["o1"].view(rows, d), rowmajor[0], rows, d, d, bias=W["sa_out"][1], res=x.view(rows, d), drop_p=p, seed=sd, call_id=calls[1]).view(B, S, d)
    t["h2"], t["mean2"], t["rstd2"] = ops.layernorm_fwd(t["x1"], *W["ln2"])
    t["q"] = ops.gemm(t["h2"].view(rows, d), rowmajor[1], rows, d, d, bias=W["q_proj"][1]).view(B, S, d)
    t["o2"], t["lse2"] = ops.attention_fwd(t["q"], kv, kv, B, H, S, kv.shape[1], d // H, 0, 0, d, p_drop=p, seed=seed, call_id=calls[2])
    t.update(ops.tlayer_ffn(t["x1"], {"out": W["out2"], "ln3": W["ln3"], "ffn1": W["ffn1"], "ffn2": W["ffn2"]}, o=t["o2"], p=p, seed=seed, calls=(calls[3], calls[4], calls[5])))
    return t


def tlayer_matrices(params, rt):
    """the row-major bf16 weight matrices of a layer in the order TLayerFn takes their packed forms: self in_proj, self out_proj,
    [cross in_proj rows 0..d-1 (the q projection), cross out_proj,] linear1, linear2"""
    cross = len(params) == 18
    d = params[2].shape[1]
    mats = [rt.lp(params[2]), rt.lp(params[4])]
    if cross:
        mats += [rt.lp(params[8])[:d], rt.lp(params[10])]
    return mats + [rt.lp(params[-4]), rt.lp(params[-2])]


def tlayer_matrices_t(params, rt):
    """the matrices whose TRANSPOSES the strip-wise backward streams, in its order: linear2, linear1, the last out-projection (cross, or self
    for an encoder layer), [cross q-projection rows, self out-projection,] self in-projection"""
    cross = len(params) == 18
    d = params[2].shape[1]
    mats = [rt.lp(params[-2]), rt.lp(params[-4])]
    if cross:
        mats += [rt.lp(params[10]), rt.lp(params[8])[:d], rt.lp(params[4])]
    else:
        mats += [rt.lp(params[4])]
    return mats + [rt.lp(params[2])]


class TLayerFn(Function):
    """nn.TransformerDecoderLayer (kv = the memory's packed k | v projections [B, M, 2d]) or nn.TransformerEncoderLayer (kv None), norm_first:
    the forward is ONE launch (decoder: two, around the cross-attention's own) that also writes what LayerNormSkipFn / LinearFn / AttnFn / FFNFn would have saved; the backward calls their
    backward code in reverse order.  params = norm1 (w, b), self in_proj (w, b), self out_proj (w, b), [norm2 (w, b), cross in_proj (w, b),
    cross out_proj (w, b),] last norm (w, b), linear1 (w, b), linear2 (w, b); packed = ops.tlayer_pack(tlayer_matrices(params, rt)) when the caller
    packed several layers in one launch, else None.  Dropout call ids are drawn in the unfused order."""

    @staticmethod
    def forward(ctx, x, kv, kpm, causal, p, rt, packed, *params):
        cross = kv is not None
        assert len(params) == (18 if cross else 12)
        x = x.contiguous()
        n1w, n1b, siw, sib, sow, sob = params[:6]
        n3w, n3b, w1, b1, w2, b2 = params[-6:]
        nc = lambda: rt.next_call() if p > 0.0 else 0
        a1, o1 = nc(), nc()
        a2, o2 = (nc(), nc()) if cross else (0, 0)
        calls = (a1, o1, a2, o2, nc(), nc())
        if packed is None:
            packed = ops.tlayer_pack(tlayer_matrices(params, rt))
        nfw = 6 if cross else 4   # packed = the forward matrices [+ the transposes of tlayer_matrices_t for the strip-wise backward]
        ctx.packed_t = tuple(packed[nfw:]) if len(packed) > nfw else None
        W = {"ln1": (n1w.detach(), n1b.detach()), "sa_in": (packed[0], sib.detach()), "sa_out": (packed[1], sob.detach()),
             "ln3": (n3w.detach(), n3b.detach()), "ffn1": (packed[nfw - 2], b1.detach()), "ffn2": (packed[nfw - 1], b2.detach())}
        if cross:
            n2w, n2b, ciw, cib, cow, cob = params[6:12]
            W.update({"ln2": (n2w.detach(), n2b.detach()), "q_proj": (packed[2], cib.detach()[:x.shape[2]]), "out2": (packed[3], cob.detach())})
            ready = getattr(kv, "_ralf_ready", None)   # (Runtime.ahead: kv was projected on another stream)
            kv = kv.contiguous()
        if x.shape[1] <= ops.TLAYER_MAX_ROWS:
            t = ops.tlayer_fwd(x, W, causal=causal, kpm=kpm, kv=kv, p_attn=p, p_res=p, seed=rt.seed if p > 0.0 else None, calls=calls,
                               kv_ready=ready if cross else None)
        else:
            assert cross, "long encoder layers take TLNQKVFn / TFFNFn"
            t = _tlayer_fwd_long(x, W, (rt.lp(sow), rt.lp(params[8])[:x.shape[2]]), kv, kpm, causal, p, rt.seed if p > 0.0 else None, calls)
        ctx.keys = tuple(k for k in t if k != "out")
        ctx.save_for_backward(x, kv, kpm, *params, *[t[k] for k in ctx.keys])
        ctx.cfg = (cross, causal, p, calls, rt, rt.dropout_tag(x))
        rt.tag_dropout(t["out"], p, calls[5])
        return t["out"]

    @staticmethod
    def backward(ctx, dy):
        cross, causal, p, calls, rt, tag_in = ctx.cfg
        sv = ctx.saved_tensors
        x, kv, kpm = sv[:3]
        npar = 18 if cross else 12
        params = sv[3:3 + npar]
        t = dict(zip(ctx.keys, sv[3 + npar:]))
        n1w, n1b, siw, sib, sow, sob = params[:6]
        n3w, n3b, w1, b1, w2, b2 = params[-6:]
        B, S, d = x.shape
        H, rows = 8, B * S
        need = ctx.needs_input_grad[7:]
        tag = (lambda call: (p, call)) if (p > 0.0 and rt.ln_dropout) else (lambda call: None)
        two = lambda a: a.view(rows, a.shape[-1])
        grads = [None] * npar
        if rt.fused_ffn_bwd and rt.ln_dropout and rows % 64 == 0 and rows >= 1024 and dy.is_cuda:   # (a handful of strips would leave the chip empty: 28 us per launch at 4 workgroups)
            # the rows of all samples as 64-row strips (ops.tlayer_bwd / tlayer_bwd_lnqkv): [feed-forward + LayerNorm 3 + out-projection data
            # gradients] - attention - [q projection + LayerNorm 2 + self out-projection] - attention - [in-projection + LayerNorm 1]: 7 launches
            # (encoder layer: 4) instead of 13 (8); weight / bias gradients from the tensors they write, through the usual (grouped) launches
            pt = ctx.packed_t or ops.tlayer_pack(tlayer_matrices_t(params, rt), transpose=tuple(range(6 if cross else 4)))

            def ln_into(wi):
                gg, gb = rt.gview(params[wi]), rt.gview(params[wi + 1])
                if gg is not None and gb is not None:
                    return gg, gb, True
                return (torch.zeros(params[wi].numel(), dtype=torch.float32, device=x.device), torch.zeros(params[wi + 1].numel(), dtype=torch.float32, device=x.device), False)

            def lin_grads(g2, xin, wi, prows):   # the weight / bias gradient of params[wi][prows] from (dy, x), as LinearFn.backward
                W, bias = params[wi], params[wi + 1]
                r0, r1 = prows
                N, K, full = r1 - r0, W.shape[1], (r1 - r0) == W.shape[0]
                if need[wi]:
                    gv = rt.gview(W)
                    dW = wgrad(g2, two(xin), N, K, rows, gv[r0:r1] if gv is not None else None, rt)
                    if dW is not None and not full:
                        gfull = torch.zeros(W.shape, dtype=torch.float32, device=dW.device)
                        gfull[r0:r1] = dW
                        dW = gfull
                    grads[wi] = _acc(grads[wi], dW)
                if need[wi + 1]:
                    gv = rt.gview(bias)
                    db = bgrad(g2, rows, N, gv[r0:r1] if gv is not None else None, rt)
                    if db is not None and not full:
                        gfull = torch.zeros(W.shape[0], dtype=torch.float32, device=db.device)
                        gfull[r0:r1] = db
                        db = gfull
                    grads[wi + 1] = _acc(grads[wi + 1], db)

            dy2 = _2d(dy.contiguous())
            if dy2.dtype != rt.dtype:
                dy2 = ops.cast(dy2, rt.dtype)
            dy_m = rt.masked_grad(dy2, p, calls[5]) if p > 0.0 else dy2
            r = t["x2"] if cross else t["x1"]
            o_last, wo_i, call_last = (t["o2"], 10, calls[3]) if cross else (t["o1"], 4, calls[1])
            gg, gb, direct = ln_into(npar - 6)
            tb = ops.tlayer_bwd(dy_m, two(t["hid"]), {"w2t": pt[0], "w1t": pt[1], "wot": pt[2]}, p=p, dy=dy2, x2=two(r), mean3=t["mean3"], rstd3=t["rstd3"],
                                gamma=n3w.detach(), dgamma=gg, dbeta=gb, seed=rt.seed if p > 0.0 else None, call_out=call_last)
            if not direct:
                grads[npar - 6], grads[npar - 5] = gg, gb
            lin_grads(dy_m, t["hid"], npar - 2, (0, d))
            lin_grads(tb["dz"], t["h3"], npar - 4, (0, w1.shape[0]))
            lin_grads(tb["g_m"], o_last, wo_i, (0, d))
            g, do = tb["g"], tb["d_o"]
            dkv = None
            if cross:
                n2w, n2b, ciw, cib, cow, cob = params[6:12]
                c = _Ctx((t["q"], kv, t["o2"], t["lse2"], None), cfg=(B, H, S, kv.shape[1], d // H, (0, 0, d), False, p, calls[2], rt))
                dq, dkv = AttnFn.backward(c, do.view(x.shape))[:2]
                gg, gb, direct = ln_into(6)
                dx1, dx1m, do = ops.tlayer_bwd_lnqkv(two(dq), pt[3], two(t["x1"]), t["mean2"], t["rstd2"], n2w.detach(), skip=g, dgamma=gg, dbeta=gb, p=p,
                                                    seed=rt.seed if p > 0.0 else None, call=calls[1], wo_t=pt[4])
                if not direct:
                    grads[6], grads[7] = gg, gb
                lin_grads(two(dq), t["h2"], 8, (0, d))
                lin_grads(dx1m if dx1m is not None else dx1, t["o1"], 4, (0, d))
                g = dx1
            c = _Ctx((t["qkv"], None, t["o1"], t["lse1"], kpm), cfg=(B, H, S, S, d // H, (0, d, 2 * d), causal, p, calls[0], rt))
            dqkv = AttnFn.backward(c, do.view(x.shape))[0]
            gg, gb, direct = ln_into(0)
            pin, cin = tag_in if tag_in is not None else (0.0, 0)
            dx, dxm = ops.tlayer_bwd_lnqkv(two(dqkv), pt[-1], two(x), t["mean1"], t["rstd1"], n1w.detach(), skip=g, dgamma=gg, dbeta=gb, p=pin,
                                           seed=rt.seed if pin > 0.0 else None, call=cin)
            if not direct:
                grads[0], grads[1] = gg, gb
            if dxm is not None:
                rt.offer_masked(dx, cin, dxm.view(x.shape))
            lin_grads(two(dqkv), t["h1"], 2, (0, 3 * d))
            return (dx.view(x.shape), dkv, None, None, None, None, None, *grads)

        def lin_bwd(g, xin, W, b, wi, prows, has_res, call):
            c = _Ctx((two(xin), W), (True, need[wi], need[wi + 1]), rt=rt, rows=prows, p=(p if call else 0.0), call=call, fan=None, bias=b,
                     has_b=True, has_res=has_res, xshape=x.shape)
            dx, dW, db = LinearFn.backward(c, g)[:3]
            grads[wi], grads[wi + 1] = _acc(grads[wi], dW), _acc(grads[wi + 1], db)
            return dx

        def ln_bwd(g, skip, xin, gw, gb, mean, rstd, wi, tg):
            dx, dg, db = _ln_backward(_Ctx((xin, gw, mean, rstd), rt=rt, beta=gb, tag=tg), g, skip)
            grads[wi], grads[wi + 1] = dg, db
            return dx

        # feed-forward block
        r = t["x2"] if cross else t["x1"]
        c = _Ctx((two(t["h3"]), w1, w2, two(t["hid"]), None), (True,), b1=b1, b2=b2, rt=rt, act="relu", p=p, has_res=True, xshape=x.shape, c2=calls[5])
        dh, grads[npar - 4], grads[npar - 3], grads[npar - 2], grads[npar - 1], dres = FFNFn.backward(c, dy)[:6]
        g = ln_bwd(dh, dres, r, n3w, n3b, t["mean3"], t["rstd3"], npar - 6, tag(calls[3] if cross else calls[1]))
        dkv = None
        if cross:   # cross-attention block
            n2w, n2b, ciw, cib, cow, cob = params[6:12]
            do = lin_bwd(g, t["o2"], cow, cob, 10, (0, d), True, calls[3])
            c = _Ctx((t["q"], kv, t["o2"], t["lse2"], None), cfg=(B, H, S, kv.shape[1], d // H, (0, 0, d), False, p, calls[2], rt))
            dq, dkv = AttnFn.backward(c, do)[:2]
            dh = lin_bwd(dq, t["h2"], ciw, cib, 8, (0, d), False, 0)
            g = ln_bwd(dh, g, t["x1"], n2w, n2b, t["mean2"], t["rstd2"], 6, tag(calls[1]))
        # self-attention block
        do = lin_bwd(g, t["o1"], sow, sob, 4, (0, d), True, calls[1])
        c = _Ctx((t["qkv"], None, t["o1"], t["lse1"], kpm), cfg=(B, H, S, S, d // H, (0, d, 2 * d), causal, p, calls[0], rt))
        dqkv = AttnFn.backward(c, do)[0]
        dh = lin_bwd(dqkv, t["h1"], siw, sib, 2, (0, 3 * d), False, 0)
        dx = ln_bwd(dh, g, x, n1w, n1b, t["mean1"], t["rstd1"], 0, tag_in)
        return (dx, dkv, None, None, None, None, None, *grads)


class TFFNFn(Function):
    """the tail of a pre-norm encoder layer for ANY number of rows, forward in one launch on 64-row strips (ops.tlayer_ffn): with the attention
    output o, r = x + drop(o Wo^T + bo) first, then r + drop(W2 drop(relu(W1 LN(r) + b1)) + b2) -- the layers whose attention is too long for
    TLayerFn (the image encoder: 256 tokens per sample).  Saves what LinearFn + LayerNormSkipFn + FFNFn save; the backward is theirs.
    packed = ops.tlayer_pack([out_proj (when o is given), linear1, linear2]) or None.  params = [out_proj (w, b),] norm (w, b), linear1 (w, b),
    linear2 (w, b).  Returns the gradients of o (if given) and x."""

    @staticmethod
    def forward(ctx, x, o, p, rt, packed, *params):
        x = x.contiguous()
        with_o = o is not None
        assert len(params) == (8 if with_o else 6)
        n3w, n3b, w1, b1, w2, b2 = params[-6:]
        nc = lambda: rt.next_call() if p > 0.0 else 0
        co = nc() if with_o else 0
        c1, c2 = nc(), nc()
        if packed is None:
            packed = ops.tlayer_pack(([rt.lp(params[0])] if with_o else []) + [rt.lp(w1), rt.lp(w2)])
        k0 = 1 if with_o else 0   # packed = [out_proj,] linear1, linear2 [, W2^T, W1^T, Wo^T]
        W = {"ln3": (n3w.detach(), n3b.detach()), "ffn1": (packed[k0], b1.detach()), "ffn2": (packed[k0 + 1], b2.detach())}
        if with_o:
            o = o.contiguous()
            W["out"] = (packed[0], params[1].detach())
        t = ops.tlayer_ffn(x, W, o=o, p=p, seed=rt.seed if p > 0.0 else None, calls=(co, c1, c2))
        ctx.save_for_backward(x, o, *params, t["h3"], t["mean3"], t["rstd3"], t["hid"], t.get("x2"))
        ctx.cfg = (with_o, p, co, c2, rt, rt.dropout_tag(x))
        ctx.packed_t = tuple(packed[3:6]) if (with_o and len(packed) >= 6) else None   # (W2^T, W1^T, Wo^T in fragment order, when the caller packed them)
        rt.tag_dropout(t["out"], p, c2)
        return t["out"]

    @staticmethod
    def backward(ctx, dy):
        with_o, p, co, c2, rt, tag_in = ctx.cfg
        sv = ctx.saved_tensors
        x, o = sv[:2]
        npar = 8 if with_o else 6
        params = sv[2:2 + npar]
        h3, mean3, rstd3, hid, x2 = sv[2 + npar:]
        n3w, n3b, w1, b1, w2, b2 = params[-6:]
        rows = x.numel() // x.shape[-1]
        need = ctx.needs_input_grad[5:]
        if with_o and rt.fused_ffn_bwd and rows % 64 == 0:
            # one launch for the four data-gradient steps (dz, dh, LayerNorm backward + skip, d o); the weight / bias gradients from the tensors
            # it wrote, through the same (grouped) launches as the per-operation path
            dy2 = _2d(dy.contiguous())
            if dy2.dtype != rt.dtype:
                dy2 = ops.cast(dy2, rt.dtype)
            dy_m = rt.masked_grad(dy2, p, c2) if p > 0.0 else dy2
            wo, bo = params[0], params[1]
            pt = ctx.packed_t or ops.tlayer_pack([rt.lp(w2), rt.lp(w1), rt.lp(wo)], transpose=(0, 1, 2))
            gg, gb = rt.gview(n3w), rt.gview(n3b)
            direct = gg is not None and gb is not None
            if not direct:
                gg, gb = torch.zeros(n3w.numel(), dtype=torch.float32, device=x.device), torch.zeros(n3b.numel(), dtype=torch.float32, device=x.device)
            t = ops.tlayer_bwd(dy_m, hid.view(rows, -1), {"w2t": pt[0], "w1t": pt[1], "wot": pt[2]}, p=p, dy=dy2, x2=x2.view(rows, -1), mean3=mean3, rstd3=rstd3,
                               gamma=n3w.detach(), dgamma=gg, dbeta=gb, seed=rt.seed if p > 0.0 else None, call_out=co)
            d_model, ff = x.shape[-1], hid.shape[-1]
            dW2 = wgrad(dy_m, hid.view(rows, -1), d_model, ff, rows, rt.gview(w2), rt)
            db2 = bgrad(dy_m, rows, d_model, rt.gview(b2), rt)
            dW1 = wgrad(t["dz"], h3.view(rows, -1), ff, d_model, rows, rt.gview(w1), rt)
            db1 = bgrad(t["dz"], rows, ff, rt.gview(b1), rt)
            dWo = wgrad(t["g_m"], o.view(rows, -1), d_model, d_model, rows, rt.gview(wo), rt) if need[0] else None
            dbo = bgrad(t["g_m"], rows, d_model, rt.gview(bo), rt) if need[1] else None
            return (t["g"].view(x.shape), t["d_o"].view(o.shape), None, None, None, dWo, dbo, None if direct else gg, None if direct else gb, dW1, db1, dW2, db2)
        c = _Ctx((h3.view(rows, -1), w1, w2, hid.view(rows, -1), None), (True,), b1=b1, b2=b2, rt=rt, act="relu", p=p, has_res=True, xshape=x.shape, c2=c2)
        dh, dW1, db1, dW2, db2, dres = FFNFn.backward(c, dy)[:6]
        if not with_o:
            dx, dg, db = _ln_backward(_Ctx((x, n3w, mean3, rstd3), rt=rt, beta=n3b, tag=tag_in), dh, dres)
            return dx, None, None, None, None, dg, db, dW1, db1, dW2, db2
        tag = (p, co) if (p > 0.0 and rt.ln_dropout) else None
        g, dg, db = _ln_backward(_Ctx((x2, n3w, mean3, rstd3), rt=rt, beta=n3b, tag=tag), dh, dres)   # gradient of r = x + drop(out-projection)
        c = _Ctx((o.view(rows, -1), params[0]), (True, need[0], need[1]), rt=rt, rows=(0, params[0].shape[0]), p=p, call=co, fan=None, bias=params[1],
                 has_b=True, has_res=True, xshape=o.shape)
        do, dWo, dbo = LinearFn.backward(c, g)[:3]
        return g, do, None, None, None, dWo, dbo, dg, db, dW1, db1, dW2, db2


class TLNQKVFn(Function):
    """(qkv, x) = (LN(x) Win^T + bin, x) for ANY number of rows, forward in one launch on 64-row strips (ops.tlayer_lnqkv): the head of a
    pre-norm layer whose attention is too long for TLayerFn.  Saves what LayerNormSkipFn + LinearFn save; the backward is theirs.
    packed_in = ops.tlayer_pack([in_proj_weight])[0] or None."""

    @staticmethod
    def forward(ctx, x, rt, packed_in, n1w, n1b, siw, sib, packed_in_t=None):
        x = x.contiguous()
        if packed_in is None:
            packed_in = ops.tlayer_pack([rt.lp(siw)])[0]
        t = ops.tlayer_lnqkv(x, {"ln1": (n1w.detach(), n1b.detach()), "sa_in": (packed_in, sib.detach())})
        ctx.save_for_backward(x, n1w, n1b, siw, sib, t["h1"], t["mean1"], t["rstd1"])
        ctx.cfg = (rt, rt.dropout_tag(x))
        ctx.packed_in_t = packed_in_t   # Win^T in fragment order when the caller packed it (the one-launch backward)
        return t["qkv"], x.view_as(x)

    @staticmethod
    def backward(ctx, dqkv, dskip):
        rt, tag_in = ctx.cfg
        x, n1w, n1b, siw, sib, h1, mean1, rstd1 = ctx.saved_tensors
        rows = x.numel() // x.shape[-1]
        if rt.fused_ffn_bwd and rows % 64 == 0:
            # in-projection data gradient + LayerNorm backward (+ skip gradient, + the masked copy the producing block's backward asks for)
            dq2 = _2d(dqkv.contiguous())
            if dq2.dtype != rt.dtype:
                dq2 = ops.cast(dq2, rt.dtype)
            wt = ctx.packed_in_t if ctx.packed_in_t is not None else ops.tlayer_pack([rt.lp(siw)], transpose=(0,))[0]
            gg, gb = rt.gview(n1w), rt.gview(n1b)
            direct = gg is not None and gb is not None
            if not direct:
                gg, gb = torch.zeros(n1w.numel(), dtype=torch.float32, device=x.device), torch.zeros(n1b.numel(), dtype=torch.float32, device=x.device)
            p, call = tag_in if tag_in is not None else (0.0, 0)
            dx, dxm = ops.tlayer_bwd_lnqkv(dq2, wt, x, mean1, rstd1, n1w.detach(), skip=dskip.contiguous() if dskip is not None else None, dgamma=gg, dbeta=gb,
                                           p=p, seed=rt.seed if p > 0.0 else None, call=call)
            if dxm is not None:
                rt.offer_masked(dx, call, dxm)
            N, K = siw.shape
            dW = wgrad(dq2, h1.view(rows, -1), N, K, rows, rt.gview(siw), rt) if ctx.needs_input_grad[5] else None
            db = bgrad(dq2, rows, N, rt.gview(sib), rt) if ctx.needs_input_grad[6] else None
            return dx, None, None, None if direct else gg, None if direct else gb, dW, db, None
        c = _Ctx((h1.view(rows, -1), siw), (True, ctx.needs_input_grad[5], ctx.needs_input_grad[6]), rt=rt, rows=(0, siw.shape[0]), p=0.0, call=0, fan=None,
                 bias=sib, has_b=True, has_res=False, xshape=x.shape)
        dh, dW, db = LinearFn.backward(c, dqkv)[:3]
        skip = dskip.contiguous() if dskip is not None else None
        dx, dg, dbt = _ln_backward(_Ctx((x, n1w, mean1, rstd1), rt=rt, beta=n1b, tag=tag_in), dh, skip)
        return dx, None, None, dg, dbt, dW, db, None


class TFeedForwardFn(Function):
    """W2 gelu(W1 LN(x) + b1) + b2 (the reference's FeedForward, common/attention.py:15-30) for any number of rows, forward in one launch on
    64-row strips (ops.tlayer_ffn, act="gelu", no residual).  Saves what LayerNormFn + FFNFn save; the backward is theirs."""

    @staticmethod
    def forward(ctx, x, rt, packed, lnw, lnb, w1, b1, w2, b2):
        x = x.contiguous()
        ctx.pt = None
        if packed is None:
            # W1, W2 and (when a backward will follow) W2^T, W1^T in ONE launch per step and module, none while the weights stand still
            rows = x.numel() // x.shape[-1]
            with_t = rt.fused_ffn_bwd and rows % 64 == 0 and any(ctx.needs_input_grad)
            packed = rt.packed((w1, w2), with_t)
            if with_t:
                ctx.pt = (packed[2], packed[3])
        t = ops.tlayer_ffn(x, {"ln3": (lnw.detach(), lnb.detach()), "ffn1": (packed[0], b1.detach()), "ffn2": (packed[1], b2.detach())}, act="gelu", residual=False)
        ctx.save_for_backward(x, lnw, lnb, w1, b1, w2, b2, t["h3"], t["mean3"], t["rstd3"], t["hid"], t["z"])
        ctx.cfg = (rt, rt.dropout_tag(x))
        return t["out"]

    @staticmethod
    def backward(ctx, dy):
        rt, tag_in = ctx.cfg
        x, lnw, lnb, w1, b1, w2, b2, h3, mean3, rstd3, hid, z = ctx.saved_tensors
        rows = x.numel() // x.shape[-1]
        if rt.fused_ffn_bwd and rows % 64 == 0:   # dz, dh and the LayerNorm backward in one launch (ops.tlayer_bwd, gelu)
            dy2 = _2d(dy.contiguous())
            if dy2.dtype != rt.dtype:
                dy2 = ops.cast(dy2, rt.dtype)
            pt = ctx.pt if ctx.pt is not None else ops.tlayer_pack([rt.lp(w2), rt.lp(w1)], transpose=(0, 1))
            gg, gb = rt.gview(lnw), rt.gview(lnb)
            direct = gg is not None and gb is not None
            if not direct:
                gg, gb = torch.zeros(lnw.numel(), dtype=torch.float32, device=x.device), torch.zeros(lnb.numel(), dtype=torch.float32, device=x.device)
            p, call = tag_in if tag_in is not None else (0.0, 0)
            t = ops.tlayer_bwd(dy2, z.view(rows, -1), {"w2t": pt[0], "w1t": pt[1]}, p=p, dy=None, x2=x.view(rows, -1), mean3=mean3, rstd3=rstd3, gamma=lnw.detach(),
                               dgamma=gg, dbeta=gb, seed=rt.seed if p > 0.0 else None, call_out=call, gelu=True)
            if p > 0.0:
                rt.offer_masked(t["g"], call, t["g_m"])
            d_model, ff = x.shape[-1], hid.shape[-1]
            dW2 = wgrad(dy2, hid.view(rows, -1), w2.shape[0], ff, rows, rt.gview(w2), rt)
            db2 = bgrad(dy2, rows, w2.shape[0], rt.gview(b2), rt)
            dW1 = wgrad(t["dz"], h3.view(rows, -1), ff, d_model, rows, rt.gview(w1), rt)
            db1 = bgrad(t["dz"], rows, ff, rt.gview(b1), rt)
            return t["g"].view(x.shape), None, None, None if direct else gg, None if direct else gb, dW1, db1, dW2, db2
        c = _Ctx((h3.view(rows, -1), w1, w2, hid.view(rows, -1), z.view(rows, -1)), (True,), b1=b1, b2=b2, rt=rt, act="gelu", p=0.0, has_res=False, xshape=x.shape, c2=0)
        dh, dW1, db1, dW2, db2 = FFNFn.backward(c, dy)[:5]
        dx, dg, db = _ln_backward(_Ctx((x, lnw, mean3, rstd3), rt=rt, beta=lnb, tag=tag_in), dh, None)
        return dx, None, None, dg, db, dW1, db1, dW2, db2


def tffn_supported(x, rt, d, dim_ff) -> bool:
    return (rt.fused_layers and rt.fused_ffn and rt.dtype == torch.bfloat16 and x.is_cuda and d == 256 and dim_ff == 1024 and x.shape[-1] == d
            and (x.numel() // d) % 64 == 0)


def _acc(a, b):
    return b if a is None else (a if b is None else a + b)


# ----------------------------------------------------------------------------------------------
# convolutional backbone (NHWC)
# ----------------------------------------------------------------------------------------------
def _conv_splitk(M: int, K: int) -> int:
    """3x3 convolutions of layer4 (M = 4096 output pixels, K = 4608): 32 x 4 tiles of 128 x 128 leave half the chip idle and
    512 tiles of 64 x 64 run at the LDS rate; 4 k-splits on the 128 x 128 tiles: 65 -> 45 us (tools/conv_sweep.py)"""
    return 4 if (M <= 4096 and K >= 4096) else 1


class ConvFn(Function):
    """NHWC conv2d through the implicit-im2col GEMM.  W is the fp32 OIHW master weight.
    `pos` (optional, constant [OH*OW, Co]) is added to every image (fused positional table)."""

    @staticmethod
    def forward(ctx, x, W, b, stride, pad, pos, rt, fork=False, stats=False):
        """fork: also return x itself as a second output (the residual / downsample branch consumes THAT alias);
        its gradient comes back into this backward and is added in the data-gradient GEMM's epilogue instead of a
        separate autograd accumulation pass over the block input."""
        B, H, Wd, C = x.shape
        Co, Ci, kh, kw = W.shape
        OH, OW = (H + 2 * pad - kh) // stride + 1, (Wd + 2 * pad - kw) // stride + 1
        M = B * OH * OW
        x = x.contiguous()
        bias = b.detach() if b is not None else None
        # stats: the following BatchNorm's batch statistics come out of this GEMM's epilogue (per-64-row partial sums)
        sk = _conv_splitk(M, kh * kw * C) if kh > 1 else 1
        cst = ops.colstats_buffer(M, Co, x.device) if (stats and Co % 64 == 0 and b is None and pos is None and sk == 1) else None
        if kh == 1 and stride == 1:
            if pos is not None:  # batched over images so the [hw, Co] table is shared (batch stride 0)
                hw = OH * OW
                y = ops.gemm(x.view(-1, C), rt.lp(W).view(Co, Ci), hw, Co, Ci, bias=bias, res=pos, batch=(B, 1),
                             sA=(hw * Ci, 0), sC=(hw * Co, 0), sR=(0, 0), out=torch.empty(M, Co, dtype=x.dtype, device=x.device))
            elif bias is None and ops.conv1x1_k64_ok(x.view(-1, C), Co) and (cst is not None or not stats):
                # 64 input channels (layer1's conv3 / downsample / first conv1): one k-tile per output tile in front of a store four times its
                # size -- a wave per 64 x 64 tile with the weights in registers (ops.conv1x1_k64), the same bits as the tiled product
                y = ops.conv1x1_k64(x.view(-1, C), rt.lp(W).view(Co, Ci), colstats=cst)
            else:
                y = ops.gemm(x.view(-1, C), rt.lp(W).view(Co, Ci), M, Co, Ci, bias=bias, colstats=cst)
        elif (rt.stem_direct and kh == 7 and kw == 7 and stride == 2 and pad == 3 and C == 8 and Co == 64 and b is None and pos is None
              and x.dtype == torch.bfloat16):
            # the stem: direct form (one output row per tile, the input patch staged once, weights in registers), statistics from its own epilogue
            y, cst = ops.stem7x7_fwd(x, rt.lp(W, "ohwi"), want_stats=stats)
        else:
            assert pos is None
            geom = dict(RH=OH, RW=OW, SH=H, SW=Wd, SC=C, KH=kh, KW=kw, stride=stride, pad=pad, mode=0)
            y = ops.gemm(x, rt.lp(W, "ohwi"), M, Co, kh * kw * C, conv=geom, gather=1, bias=bias, colstats=cst, splitk=sk)
        ctx.save_for_backward(x, W)
        ctx.bias = b
        ctx.cfg = (stride, pad, OH, OW, b is not None, rt)
        ctx.bn = rt.take_bn_tag(x)   # x is a relu(BatchNorm) output: this convolution's data gradient also feeds that BatchNorm's backward
        ctx.fork = fork
        ctx.set_materialize_grads(False)   # no zero-filled "gradient" of the statistics output / an unused fork (59 fills per step)
        out = (y.view(B, OH, OW, Co),) + ((x,) if fork else ())
        if stats:
            if cst is None:   # shape not covered by the fused statistics: BatchNorm computes them itself
                cst = torch.empty(0, dtype=torch.float32, device=x.device)
            ctx.mark_non_differentiable(cst)
            out = out + (cst,)
        return out if len(out) > 1 else out[0]

    @staticmethod
    def backward(ctx, dy, *rest):
        dskip = rest[0] if ctx.fork else None
        x, W = ctx.saved_tensors
        stride, pad, OH, OW, has_b, rt = ctx.cfg
        if dy is None:   # only the fork alias carried a gradient
            return (dskip,) + (None,) * 8
        B, H, Wd, C = x.shape
        Co, Ci, kh, kw = W.shape
        M = B * OH * OW
        dy = dy.contiguous()
        dy2 = dy.view(M, Co)
        dx = dW = db = None
        one = kh == 1 and stride == 1
        if ctx.needs_input_grad[0]:
            sk = dskip.contiguous().view(-1, C) if dskip is not None else None
            Mi = B * H * Wd
            dsk = 1 if one else _conv_splitk(Mi, kh * kw * Co)
            bnb = None
            if ctx.bn is not None and dsk == 1 and ctx.bn[0].shape == (Mi, C) and (one or (Co % (64 if dy.dtype == torch.bfloat16 else 32) == 0 and stride in (1, 2))):
                bnb = ctx.bn + (torch.empty((Mi + 63) // 64, 2, C, dtype=torch.float32, device=dy.device),)
            if one:
                dx = ops.gemm(dy2, rt.lp(W).view(Co, Ci), M, Ci, Co, b_kcontig=False, res=sk, bnb=bnb)
            else:
                geom = dict(RH=H, RW=Wd, SH=OH, SW=OW, SC=Co, KH=kh, KW=kw, stride=stride, pad=pad, mode=1)
                dx = ops.gemm(dy, rt.lp(W, "ikwo"), Mi, C, kh * kw * Co, conv=geom, gather=1, res=sk, splitk=dsk, bnb=bnb)
            if bnb is not None:
                rt.offer_bn_stats(dx, bnb[3])
            dx = dx.view(B, H, Wd, C)
        if ctx.needs_input_grad[1]:
            gv = rt.gview(W)
            if one:
                dW = wgrad(dy2, x.view(-1, C), Co, Ci, M, gv, rt)
                dW = dW.view(Co, Ci, 1, 1) if dW is not None else None
            else:
                geom = dict(RH=OH, RW=OW, SH=H, SW=Wd, SC=C, KH=kh, KW=kw, stride=stride, pad=pad, mode=0)
                direct = rt.conv_wgrad_direct and kw == kh and ops.conv3x3_wgrad_supported(dy, x, stride, pad, kh, kw)
                stem = (rt.stem_direct and kh == 7 and kw == 7 and stride == 2 and pad == 3 and C == 8 and Co == 64 and Ci == 4 and x.dtype == torch.bfloat16
                        and dy.dtype == torch.bfloat16)
                def run(out=None):
                    if stem:     # the stem: direct form (ops.stem7x7_wgrad), straight into OIHW
                        return ops.stem7x7_wgrad(x, dy, out=out)
                    if direct:   # 3x3 / stride 1: the direct form (dy tile and halo patch staged once for all nine taps), straight into OIHW
                        return ops.conv3x3_wgrad(dy, x, out=out, stride=stride)
                    g = ops.gemm(dy2, x, Co, kh * kw * C, M, a_kcontig=False, b_kcontig=False, conv=geom, gather=2,
                                 out_dtype=torch.float32, splitk=_splitk_for(Co, kh * kw * C, M))
                    # fp32 [Co][kh][kw][Cpad] -> OIHW master layout (drops the stem's channel padding)
                    return ops.permute4(g, (Co, Ci, kh, kw), (kh * kw * C, 1, kw * C, C), kw, torch.float32, out=out)
                if gv is not None:   # used once per step: placed straight into the flat buffer (zeroed at step start)
                    rt.side(lambda: run(gv), gv, dy2, x)
                else:
                    dW = run()
        if has_b and ctx.needs_input_grad[2]:
            db = bgrad(dy2, M, Co, rt.gview(ctx.bias), rt)
        return dx, dW, db, None, None, None, None, None, None


def conv2d(x, W, b, stride, pad, rt, pos=None, fork=False, stats=False):
    """returns y, then x itself when fork, then the [M/64, 2, Co] column-statistics partials when stats."""
    return ConvFn.apply(x, W, b, stride, pad, pos, rt, fork, stats)


@torch.no_grad()
def conv_bn_infer(x, W, scale, shift, stride, pad, relu, res, rt):
    """inference: act(conv(x) * scale + shift (+ res)) in ONE GEMM -- the eval-mode BatchNorm (scale / shift from ops.bn_fold_batched) and
    the ReLU live in the convolution's epilogue; no normalisation pass over the activation (at B = 256 the 53 bn_apply launches were
    3.5 ms of a 12 ms backbone forward).  relu: False, True (no residual) or "post" (after the residual: the bottleneck's tail)."""
    B, H, Wd, C = x.shape
    Co, Ci, kh, kw = W.shape
    OH, OW = (H + 2 * pad - kh) // stride + 1, (Wd + 2 * pad - kw) // stride + 1
    M = B * OH * OW
    x = x.contiguous()
    act = "relu_post" if (relu and res is not None) else ("relu" if relu else None)
    r2 = res.contiguous().view(M, Co) if res is not None else None
    if kh == 1 and stride == 1 and ops.conv1x1_k64_ok(x.view(-1, C), Co):
        y = ops.conv1x1_k64(x.view(-1, C), rt.lp(W).view(Co, Ci), scale=scale, shift=shift, res=r2, relu=2 if act == "relu_post" else (1 if act == "relu" else 0))
    elif kh == 1 and stride == 1:
        y = ops.gemm(x.view(-1, C), rt.lp(W).view(Co, Ci), M, Co, Ci, bias=shift, colscale=scale, act=act, res=r2)
    else:
        geom = dict(RH=OH, RW=OW, SH=H, SW=Wd, SC=C, KH=kh, KW=kw, stride=stride, pad=pad, mode=0)
        y = ops.gemm(x, rt.lp(W, "ohwi"), M, Co, kh * kw * C, conv=geom, gather=1, bias=shift, colscale=scale, act=act, res=r2)
    return y.view(B, OH, OW, Co)


class BatchNormFn(Function):
    """y = relu?(BN(x) (+ res)) on NHWC; batch statistics + running-stat update in training."""

    @staticmethod
    def forward(ctx, x, g, b, rm, rv, res, relu, training, rt, counter=None, partials=None):
        shp = x.shape
        x2 = x.contiguous().view(-1, shp[-1])
        if partials is not None and partials.numel() == 0:
            partials = None
        y, mean, rstd, mask = ops.bn_forward(x2, g.detach(), b.detach(), rm, rv, training, relu, res.contiguous().view(-1, shp[-1]) if res is not None else None,
                                             counter=counter, partials=partials, want_mask=True)
        # backward reads x, dy and the 1-bit ReLU mask (not y: 1/16 of the bytes, twice per backward)
        ctx.save_for_backward(x2, mask if relu else None, g, mean, rstd)
        if training and relu:
            rt.tag_bn_output(y, x2, mask, mean)
        ctx.beta, ctx.rt = b, rt
        ctx.cfg = (relu, res is not None, training, shp)
        return y.view(shp)

    @staticmethod
    def backward(ctx, dy):
        x2, mask, g, mean, rstd = ctx.saved_tensors
        relu, has_res, training, shp = ctx.cfg
        gg, gb = ctx.rt.gview(g), ctx.rt.gview(ctx.beta)
        into = (gg, gb) if (gg is not None and gb is not None and training) else None
        dy2 = dy.contiguous().view(-1, shp[-1])
        part = ctx.rt.bn_stats(dy, x2.shape[0], x2.shape[1]) if (training and relu) else None
        dx, dg, db, dres = ops.bn_backward(x2, dy2, None, g.detach(), mean, rstd, relu, has_res, training, into=into, mask=mask, partials=part)
        if into is not None:
            dg = db = None
        return dx.view(shp), dg, db, None, None, (dres.view(shp) if has_res else None), None, None, None, None, None


class StemBNReluPoolFn(Function):
    """maxpool3x3s2(relu(BN(y))) with batch statistics, forward in ONE pass over the stem convolution's output and backward in two
    (ops.bn_relu_maxpool_*): the normalised 64-channel map (134 MB at B = 64, 256 x 256) is neither written nor read back.  Same values as
    BatchNormFn -> MaxPoolFn (pooled output and argmax bit for bit; the backward uses the affine form of the BatchNorm backward apply)."""

    @staticmethod
    def forward(ctx, y, g, b, rm, rv, counter, partials, rt):
        y = y.contiguous()
        if partials is not None and partials.numel() == 0:
            partials = None
        stats = ops.bn_train_stats(y.view(-1, y.shape[-1]), g.detach(), b.detach(), rm, rv, counter, partials)
        out, arg = ops.bn_relu_maxpool_fwd(y, stats[2], stats[3])
        ctx.save_for_backward(y, arg, stats, g)
        ctx.beta, ctx.rt = b, rt
        return out

    @staticmethod
    def backward(ctx, dpool):
        y, arg, stats, g = ctx.saved_tensors
        gg, gb = ctx.rt.gview(g), ctx.rt.gview(ctx.beta)
        into = (gg, gb) if (gg is not None and gb is not None) else None
        dy, dg, db = ops.bn_relu_maxpool_bwd(dpool.contiguous(), arg, y, stats, g.detach(), into)
        if into is not None:
            dg = db = None
        return dy, dg, db, None, None, None, None, None


class MaxPoolFn(Function):
    @staticmethod
    def forward(ctx, x):
        y, arg = ops.maxpool_fwd(x.contiguous())
        ctx.save_for_backward(arg)
        ctx.shape = tuple(x.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        return ops.maxpool_bwd(dy.contiguous(), ctx.saved_tensors[0], ctx.shape)


class UpsampleAddFn(Function):
    """(up, s) = (nearest_upsample(src), nearest_upsample(src) + lateral)  -- FPN top-down step."""

    @staticmethod
    def forward(ctx, src, lateral):
        ctx.shape = tuple(src.shape)
        return ops.upsample_add(src.contiguous(), lateral.contiguous())

    @staticmethod
    def backward(ctx, g_up, g_sum):
        g_up, g_sum = g_up.contiguous(), g_sum.contiguous()
        return ops.upsample_bwd(g_up, g_sum, ctx.shape), g_sum
