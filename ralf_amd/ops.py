"""Thin Python bindings of the C-ABI compute entry points (include/ralf_hip.h) on torch CUDA
tensors.  No arithmetic happens in Python/torch here: every function launches HIP kernels on
torch's current stream.  Autograd wiring lives in ralf_amd/functional.py."""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch

from . import _lib
from ._abi import RalfConvGeom, RalfGemmDesc

F32, BF16 = 0, 1
ACT = {None: 0, "none": 0, "relu": 1, "gelu": 2, "relu_post": 3}   # relu_post: after the residual
AUX = {None: 0, "relu_mask": 1, "gelu_grad": 2}
_TORCH2CODE = {torch.float32: F32, torch.bfloat16: BF16}

_ws_cache: dict = {}


def dtype_code(t: torch.Tensor) -> int:
    return _TORCH2CODE[t.dtype]


_ws_retired: list = []


def workspace(nbytes: int, device) -> torch.Tensor:
    """grow-only scratch buffer per (device, stream): reuse is ordered by the stream the kernels run on.
    A buffer that is outgrown is RETIRED, never freed: a captured graph holds its raw address (a B = 4 graph followed by a B = 64
    capture on the same stream replaced the buffer, the allocator released the old one with the first graph's pool, and the second
    graph -- whose early kernels had still used it -- faulted on replay).  Growth is geometric, so the retired bytes stay below
    the live buffer's size."""
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream)
    w = _ws_cache.get(key)
    if w is None or w.numel() < nbytes:
        if w is not None:
            _ws_retired.append(w)
        w = torch.empty(max(nbytes, 1 << 20, 2 * w.numel() if w is not None else 0), dtype=torch.uint8, device=device)
        _ws_cache[key] = w
    return w


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


_own_streams: dict = {}


def own_stream(slot, device=None, priority=None) -> "torch.cuda.Stream":
    """the library-owned HIP stream `slot` (any hashable name) of `device`, created on first use and kept for the life of the
    process (ralf_stream_create, include/ralf_hip.h).  Every side stream, graph branch and capture stream of the runtime is
    one of these -- NOT torch.cuda.Stream(), which deals out the 32 streams of a shared pool round-robin: RCCL's stream comes
    from that same pool, and when a branch landed on it the collective watchdog's event query hit a capturing stream
    (hipErrorCapturedEvent -> process abort)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (idx, slot)
    st = _own_streams.get(key)
    if st is None:
        with torch.cuda.device(idx):
            raw = ctypes.c_void_p()
            if priority is None:
                _lib.check(_lib.lib().ralf_stream_create(ctypes.byref(raw)), "ralf_stream_create")
            else:   # "high" / "low": a hardware queue of its own (ralf_stream_create_priority)
                _lib.check(_lib.lib().ralf_stream_create_priority(ctypes.byref(raw), 1 if priority == "high" else 0), "ralf_stream_create_priority")
            st = _own_streams[key] = torch.cuda.ExternalStream(raw.value, device=torch.device("cuda", idx))
    return st


def gemm_filter_tile(M: int, N: int, K: int) -> int:
    """column-tile width ralf_gemm uses for a threshold-filter product of this shape (the slot layout of RalfGemmDesc.flt_*)"""
    d = RalfGemmDesc()
    d.M, d.N, d.K, d.dtype, d.splitk = M, N, K, 1, 1
    w = _lib.lib().ralf_gemm_filter_tile(ctypes.byref(d))
    assert w > 0
    return w


def gemm_patch_variant(M: int, N: int, K: int, conv: dict, dtype=torch.bfloat16) -> int:
    """the form ralf_gemm takes for this gather = 1 convolution product: 0 = tap gather, 1 / 2 / 3 = input patch resident in LDS on 128 x 128 / 256 x 128 /
    256 x 64 tiles (ralf_gemm_patch_variant)"""
    d = RalfGemmDesc()
    d.M, d.N, d.K, d.dtype, d.splitk, d.nb0, d.nb1 = M, N, K, _TORCH2CODE[dtype], 1, 1, 1
    d.a_kcontig = d.b_kcontig = 1
    d.lda = d.ldb = K
    d.gather = 1
    d.g = RalfConvGeom(**conv)
    v = _lib.lib().ralf_gemm_patch_variant(ctypes.byref(d))
    _lib.check(min(v, 0), "ralf_gemm_patch_variant")
    return v


def gemm(A: torch.Tensor, B: torch.Tensor, M: int, N: int, K: int, *, a_kcontig=True, b_kcontig=True,
         lda=None, ldb=None, out: Optional[torch.Tensor] = None, ldc=None, out_dtype=None,
         bias=None, act=None, res=None, ldr=None, aux=None, aux_mode=None, aux_scale=1.0, out2=None,
         alpha=1.0, accumulate=False, splitk=1, batch=(1, 1), sA=(0, 0), sB=(0, 0), sC=(0, 0), sR=None,
         conv: Optional[dict] = None, gather=0, drop_p=0.0, seed=None, call_id=0, atomic=False, colstats=None,
         sBias0=0, kseg=0, sBk=0, colscale=None, bnb=None, at=None, flt=None, ln=None, few_row_split=False) -> torch.Tensor:
    """C = epi(alpha * A @ B) through ralf_gemm (see include/ralf_hip.h: RalfGemmDesc).
    ln = (gamma, beta, eps): LayerNorm of the A rows in front of a few-row product (gemm_ln_ok says where it exists).
    at = dict(mode=1|2, c1, c2, c3=None, a2=None, out=None, mask=None, relu=False): the A-operand transform with write-through (at_*)."""
    assert A.is_cuda and B.is_cuda and A.dtype == B.dtype
    d = RalfGemmDesc()
    d.dtype = dtype_code(A)
    nb0, nb1 = batch
    if out is None and flt is not None:
        out = torch.empty(8, dtype=A.dtype, device=A.device)   # (never written: the filter replaces the output matrix)
    if out is None:
        odt = out_dtype or A.dtype
        out = torch.empty((nb1, nb0, M, N) if nb0 * nb1 > 1 else (M, N), dtype=odt, device=A.device)
        if nb0 * nb1 > 1:
            sC = (M * N, nb0 * M * N)
    d.out_f32 = 1 if (out.dtype == torch.float32 and A.dtype != torch.float32) else 0
    if A.dtype == torch.float32:
        assert out.dtype == torch.float32
    d.A, d.B, d.C, d.C2 = _p(A), _p(B), _p(out), _p(out2)
    d.bias, d.res, d.aux = _p(bias), _p(res), _p(aux)
    if bias is not None:
        assert bias.dtype == torch.float32
    d.M, d.N, d.K = M, N, K
    d.lda = lda if lda is not None else (K if a_kcontig else M)
    d.ldb = ldb if ldb is not None else (K if b_kcontig else N)
    d.ldc = ldc if ldc is not None else N
    d.ldr = ldr if ldr is not None else d.ldc
    d.nb0, d.nb1 = nb0, nb1
    d.sA0, d.sA1 = sA
    d.sB0, d.sB1 = sB
    d.sC0, d.sC1 = sC
    d.sR0, d.sR1 = sR if sR is not None else sC
    d.a_kcontig, d.b_kcontig, d.gather = int(a_kcontig), int(b_kcontig), gather
    d.act, d.aux_mode, d.aux_scale = ACT[act], AUX[aux_mode], aux_scale
    d.accumulate, d.splitk, d.alpha = int(accumulate), splitk, alpha
    d.drop_p, d.seed, d.call_id, d.atomic_out = drop_p, _p(seed), call_id, int(atomic)
    d.sBias0, d.kseg, d.sBk = sBias0, kseg, sBk
    d.colscale = _p(colscale)
    if bnb is not None:   # (x, relu mask bits or None, mean, partials [ceil(M/64), 2, N]): BatchNorm-backward reductions from this epilogue
        bx, bm, bmean, bpart = bnb
        assert bx.dtype == A.dtype and bx.is_contiguous() and bx.numel() == M * N and bpart.dtype == torch.float32 and bpart.numel() >= ((M + 63) // 64) * 2 * N
        d.bnb_x, d.bnb_mask, d.bnb_mean, d.bnb_part = _p(bx), _p(bm), _p(bmean), _p(bpart)
    if ln is not None:
        d.ln_g, d.ln_b, d.ln_eps = _p(ln[0]), _p(ln[1]), float(ln[2])
    d.few_row_split = int(bool(few_row_split))
    if flt is not None:   # (thresholds fp32 [M], hit counts int32 [M, T], slots int32 [M, T, cap, 2]; T = ceil(N / gemm_filter_tile)): threshold filter (flt_*)
        th, cnt, lst = flt
        T = lst.shape[1]
        assert th.dtype == torch.float32 and th.numel() >= M and cnt.dtype == torch.int32 and tuple(cnt.shape) == (M, T) and lst.dtype == torch.int32 and lst.dim() == 4 and lst.shape[0] == M and lst.shape[3] == 2
        assert T == (N + gemm_filter_tile(M, N, K) - 1) // gemm_filter_tile(M, N, K), "slot layout does not match the tile width of this product"
        d.flt_thresh, d.flt_count, d.flt_list, d.flt_cap = _p(th), _p(cnt), _p(lst), lst.shape[2]
    if at is not None:
        a2, ao, am = at.get("a2"), at.get("out"), at.get("mask")
        assert A.dtype == torch.bfloat16 and a_kcontig and A.is_contiguous() and A.numel() == M * K
        assert (a2 is None or (a2.dtype == A.dtype and a2.is_contiguous() and a2.numel() == M * K)) and (ao is None or (ao.dtype == A.dtype and ao.is_contiguous() and ao.numel() == M * K))
        assert all(c is None or (c.dtype == torch.float32 and c.numel() >= K and c.is_contiguous()) for c in (at["c1"], at["c2"], at.get("c3")))
        assert am is None or (am.dtype == torch.uint8 and am.numel() >= M * K // 8)
        d.at_mode, d.at_relu = at["mode"], int(bool(at.get("relu", False)))
        d.at_a2, d.at_c1, d.at_c2, d.at_c3, d.at_out, d.at_mask = _p(a2), _p(at["c1"]), _p(at["c2"]), _p(at.get("c3")), _p(ao), _p(am)
    if colstats is not None:   # fp32 [ceil(M/64), 2, N]: per-64-row column sums / sums of squares of the stored output
        assert colstats.dtype == torch.float32 and colstats.numel() >= ((M + 63) // 64) * 2 * N
        d.colstats = _p(colstats)
    if conv is not None:
        g = RalfConvGeom(**conv)
        d.g = g
    L = _lib.lib()
    ws, wsn = None, 0
    if splitk > 1 and not atomic:
        wsn = splitk * nb0 * nb1 * M * N * 4
        ws = workspace(wsn, A.device)
    rc = L.ralf_gemm(ctypes.byref(d), _p(ws), wsn, _lib.stream_ptr())
    _lib.check(rc, "ralf_gemm")
    return out


# ----------------------------------------------------------------------------------------------
# normalisation / pointwise / attention / optimizer bindings (see include/ralf_hip.h)
# ----------------------------------------------------------------------------------------------
from ._abi import RalfAttnDesc  # noqa: E402


def _call(name, *args):
    rc = getattr(_lib.lib(), name)(*args, _lib.stream_ptr())
    _lib.check(rc, name)


def layernorm_fwd(x, gamma, beta, eps=1e-5, save_stats=True):
    rows, cols = x.numel() // x.shape[-1], x.shape[-1]
    y = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device) if save_stats else None
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device) if save_stats else None
    _call("ralf_layernorm_fwd", dtype_code(x), _p(x), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), rows, cols, eps)
    return y, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, need_wgrad=True, into=None, skip=None, drop=None):
    """into = (dgamma, dbeta) fp32 views to ACCUMULATE into (flat gradient buffer).
    drop = (p, seed, call_id): also return dropout(dx) with that mask as a 4th value."""
    rows, cols = x.numel() // x.shape[-1], x.shape[-1]
    dx = torch.empty_like(x)
    if into is not None:
        dg, db = into
    else:
        dg = torch.zeros(cols, dtype=torch.float32, device=x.device) if need_wgrad else None
        db = torch.zeros(cols, dtype=torch.float32, device=x.device) if need_wgrad else None
    dxd = torch.empty_like(x) if drop is not None else None
    p, seed, call = drop if drop is not None else (0.0, None, 0)
    _call("ralf_layernorm_bwd", dtype_code(x), _p(dy), _p(x), _p(gamma), _p(mean), _p(rstd), _p(dx), _p(dg), _p(db), _p(skip), rows, cols,
          _p(dxd), p, _p(seed), call)
    if drop is not None:
        return dx, dg, db, dxd
    return dx, dg, db


def colsum(x2d, rows, cols, ld=None, out=None):
    """out (fp32 [cols]) is accumulated into when given."""
    if out is None:
        out = torch.zeros(cols, dtype=torch.float32, device=x2d.device)
    _call("ralf_colsum", dtype_code(x2d), _p(x2d), ld if ld is not None else cols, _p(out), rows, cols)
    return out


def embed_fwd(idx, W, pe, S, scale, dtype):
    rows, d = idx.numel(), W.shape[1]
    out = torch.empty(*idx.shape, d, dtype=dtype, device=W.device)
    _call("ralf_embed_fwd", _TORCH2CODE[dtype], _p(idx), _p(W), _p(pe), _p(out), rows, S, d, scale)
    return out


def embed_bwd(idx, dy, vocab, scale, into=None):
    d = dy.shape[-1]
    dW = into if into is not None else torch.zeros(vocab, d, dtype=torch.float32, device=dy.device)
    _call("ralf_embed_bwd", dtype_code(dy), _p(idx), _p(dy), _p(dW), idx.numel(), d, scale)
    return None if into is not None else dW


def dropout(x, p, seed, call_id, res=None):
    """y = dropout(x) (+ res); p == 0 -> plain add."""
    y = torch.empty_like(x)
    _call("ralf_dropout", dtype_code(x), _p(x), _p(res), _p(y), x.numel(), p, _p(seed), call_id)
    return y


def xent(logits_f32, target, ignore_index, eps, grad_dtype):
    rows, V = logits_f32.numel() // logits_f32.shape[-1], logits_f32.shape[-1]
    cl = torch.empty(2, dtype=torch.float32, device=logits_f32.device)
    dl = torch.empty(logits_f32.shape, dtype=grad_dtype, device=logits_f32.device) if grad_dtype is not None else None
    _call("ralf_xent_fwd_bwd", _TORCH2CODE[grad_dtype or torch.float32], _p(logits_f32), _p(target), _p(dl), _p(cl), rows, V, ignore_index, eps)
    return cl, dl


def concat_rows(srcs, scalars=None, out=None, backward=False, dscalars=None):
    """forward: out[b, off_i + r, :] = srcs[i][b, r, :] (+ scalars[i][0]); backward: `out` is the gradient, srcs[i] are written with
    its pieces and dscalars[i][0] += sum(piece).  srcs: contiguous [B, S_i, d] tensors of one dtype."""
    n = len(srcs)
    B, d = srcs[0].shape[0], srcs[0].shape[-1]
    rows = [t.shape[1] for t in srcs]
    if out is None:
        out = torch.empty(B, sum(rows), d, dtype=srcs[0].dtype, device=srcs[0].device)
    P = ctypes.c_void_p * n
    sp = P(*[t.data_ptr() for t in srcs])
    rp = (ctypes.c_int * n)(*rows)
    sc = P(*[(s.data_ptr() if s is not None else None) for s in (scalars or [None] * n)])
    ds = P(*[(s.data_ptr() if s is not None else None) for s in (dscalars or [None] * n)])
    _call("ralf_concat_rows", dtype_code(srcs[0]), int(backward), n, sp, rp, sc, ds, _p(out), B, d)
    return out


def scale_pe_dropout(x, pe, S, scale, p=0.0, seed=None, call_id=0):
    y = torch.empty_like(x)
    d = x.shape[-1]
    _call("ralf_scale_pe_dropout", dtype_code(x), _p(x), _p(pe), _p(y), x.numel() // d, S, d, scale, p, _p(seed), call_id)
    return y


def add_scalar(x, s):
    y = torch.empty_like(x)
    cols = x.shape[-1]
    _call("ralf_add_scalar", dtype_code(x), _p(x), _p(s), _p(y), x.numel() // cols, cols, cols, cols)
    return y


def counter_add_(t, inc: int):
    """t[0] += inc on the device (int32 / int64 scalar tensor) through ralf_counter_add"""
    assert t.numel() == 1 and t.dtype in (torch.int32, torch.int64)
    _call("ralf_counter_add", _p(t), int(t.dtype == torch.int64), int(inc))
    return t


def layout_pack(cx, cy, w, h, mask, dtype):
    """-> (bbox [R*N, 8] dtype, kpm uint8 [R, N+1]) for the frozen layout encoder (ralf_layout_pack)"""
    R, N = cx.shape
    f = lambda t: t.contiguous() if t.dtype == torch.float32 else t.float().contiguous()   # noqa: E731
    cx, cy, w, h = f(cx), f(cy), f(w), f(h)
    m = mask.contiguous()
    m = m.view(torch.uint8) if m.dtype == torch.bool else m
    bbox = torch.empty(R * N, 8, dtype=dtype, device=cx.device)
    kpm = torch.empty(R, N + 1, dtype=torch.uint8, device=cx.device)
    _call("ralf_layout_pack", _TORCH2CODE[dtype], _p(cx), _p(cy), _p(w), _p(h), _p(m), _p(bbox), _p(kpm), R, N)
    return bbox, kpm


def copy2d(src, dst, rows, cols, lds, ldd):
    """dst[r * ldd + c] = src[r * lds + c] with dtype conversion (ralf_copy2d); src / dst: tensors whose data_ptr is element (0, 0)"""
    _call("ralf_copy2d", dtype_code(src), dtype_code(dst), _p(src), _p(dst), rows, cols, lds, ldd, 0)


def scale_dev(x, s):
    """fp32 x * s[0] with the fp32 factor on the device (ralf_scale_dev)"""
    y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    _call("ralf_scale_dev", dtype_code(x), _p(x), _p(s), _p(y), x.numel())
    return y


def zero_(t):
    """t[...] = 0 through the library's own kernel (ralf_zero); t contiguous, 16-byte aligned, a multiple of 16 bytes"""
    _call("ralf_zero", _p(t), t.numel() * t.element_size())
    return t


def sum_all(x, out=None):
    if out is None:
        out = torch.zeros(1, dtype=torch.float32, device=x.device)
    cols = x.shape[-1]
    _call("ralf_sum_all", dtype_code(x), _p(x), _p(out), x.numel() // cols, cols, cols)
    return out


def cast(x, dtype):
    """dtype conversion through ralf_copy2d (fp32 <-> bf16)."""
    if x.dtype == dtype:
        return x
    y = torch.empty(x.shape, dtype=dtype, device=x.device)
    n = x.numel()
    _call("ralf_copy2d", dtype_code(x), _TORCH2CODE[dtype], _p(x), _p(y), 1, n, n, n, 0)
    return y


def cast_into(x, out):
    """out[:] = x with dtype conversion (fp32 <-> bf16) through ralf_copy2d; both contiguous, same element count."""
    n = x.numel()
    assert out.numel() == n and x.is_contiguous() and out.is_contiguous()
    _call("ralf_copy2d", dtype_code(x), dtype_code(out), _p(x), _p(out), 1, n, n, n, 0)
    return out


def copy2d_acc(src, dst, rows, cols, lds, ldd):
    """dst[r*ldd + c] += src[r*lds + c] (fp32): a stacked result scattered into equally spaced views of a flat buffer"""
    assert src.dtype == torch.float32 and dst.dtype == torch.float32
    _call("ralf_copy2d", F32, F32, _p(src), _p(dst), rows, cols, lds, ldd, 1)


def wgrad_grouped(jobs):
    """jobs: list of (dy2d bf16 [rows, n_out], x2d bf16 [rows, n_in], dw fp32 view [n_out, n_in], splitk[, db fp32 [n_out] or None]):
    dw += dy^T x (and db += column sums of dy, from the tiles the product reads anyway) for all jobs in one launch (ralf_wgrad_grouped)."""
    from ._abi import RalfWgradJob

    arr = (RalfWgradJob * len(jobs))()
    for r, job in zip(arr, jobs):
        dy, x, dw, sk = job[:4]
        db = job[4] if len(job) > 4 else None
        rows, n_out = dy.shape
        if db is not None:
            assert db.dtype == torch.float32 and db.is_contiguous() and db.numel() == n_out and n_out % 256 == 0
            r.db = db.data_ptr()
        n_in = x.shape[1]
        assert dy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and dw.dtype == torch.float32 and x.shape[0] == rows
        assert dy.stride(1) == 1 and x.stride(1) == 1 and dw.stride(1) == 1 and tuple(dw.shape) == (n_out, n_in)
        r.dy, r.x, r.dw = dy.data_ptr(), x.data_ptr(), dw.data_ptr()
        r.rows, r.ld_dy, r.ld_x, r.ld_dw = rows, dy.stride(0), x.stride(0), dw.stride(0)
        r.n_out, r.n_in, r.splitk = n_out, n_in, sk
    L = _lib.lib()
    need = L.ralf_wgrad_grouped_workspace_bytes(arr, len(jobs))
    ws = workspace(need, jobs[0][0].device) if need else None
    rc = L.ralf_wgrad_grouped(arr, len(jobs), BF16, _p(ws), need, _lib.stream_ptr())
    _lib.check(rc, "ralf_wgrad_grouped")


def colsum_grouped(jobs):
    """jobs: list of (x2d bf16 [rows, cols], out fp32 [cols]): out += column sums, all in one launch"""
    from ._abi import RalfColsumJob

    arr = (RalfColsumJob * len(jobs))()
    for r, (x, out) in zip(arr, jobs):
        assert x.dtype == torch.bfloat16 and out.dtype == torch.float32 and x.stride(1) == 1 and out.is_contiguous()
        r.x, r.out, r.ld, r.rows, r.cols = x.data_ptr(), out.data_ptr(), x.stride(0), x.shape[0], x.shape[1]
    rc = _lib.lib().ralf_colsum_grouped(arr, len(jobs), BF16, _lib.stream_ptr())
    _lib.check(rc, "ralf_colsum_grouped")


def permute4(x, out_dims, strides, valid3, dtype, out=None):
    if out is None:
        out = torch.empty(out_dims, dtype=dtype, device=x.device)
    assert out.is_contiguous() and tuple(out.shape) == tuple(out_dims) and out.dtype == dtype
    _call("ralf_permute4", dtype_code(x), _TORCH2CODE[dtype], _p(x), _p(out), *out_dims, *strides, valid3)
    return out


def permute4_table(jobs, device):
    """jobs: list of (src, dst, out_dims, strides, valid3) -> (device table, njobs, total_blocks) for permute4_batched."""
    from ._abi import RalfPermuteJob

    arr = (RalfPermuteJob * len(jobs))()
    blk = 0
    for j, (src, dst, dims, strides, valid3) in enumerate(jobs):
        assert dst.is_contiguous() and tuple(dst.shape) == tuple(dims)
        r = arr[j]
        r.in_, r.out = src.data_ptr(), dst.data_ptr()
        r.s0, r.s1, r.s2, r.s3 = strides
        r.d0, r.d1, r.d2, r.d3 = dims
        r.valid3, r.src_dtype, r.dst_dtype, r.first_block = valid3, dtype_code(src), dtype_code(dst), blk
        blk += max(1, min(256, (dst.numel() + 2047) // 2048))
    raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)
    return raw, len(jobs), blk


def bn_fold_table(bns, buf, device):
    """bns: list of (gamma, beta, running_mean, running_var) fp32 tensors; buf: fp32 [2 * sum C] receiving scale | shift of each layer in
    turn -> (device job table, [(scale view, shift view)]) for bn_fold_batched"""
    from ._abi import RalfBnFoldJob

    arr = (RalfBnFoldJob * len(bns))()
    views, off = [], 0
    for j, (g, b, m, v) in enumerate(bns):
        C = g.numel()
        sc, sh = buf[off:off + C], buf[off + C:off + 2 * C]
        r = arr[j]
        r.gamma, r.beta, r.mean, r.var, r.scale, r.shift, r.C = g.data_ptr(), b.data_ptr(), m.data_ptr(), v.data_ptr(), sc.data_ptr(), sh.data_ptr(), C
        views.append((sc, sh))
        off += 2 * C
    return torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device), views


def bn_fold_batched(table, njobs, eps=1e-5):
    _call("ralf_bn_fold_batched", _p(table), njobs, eps)


def conv_relayout_table(jobs, device):
    """jobs: list of (w fp32 OIHW, ohwi [Co, kh, kw, Cip], ikwo [Ci, kh, kw, Co]) -> (device table, njobs, total_blocks)"""
    from ._abi import RalfConvRelayoutJob

    arr = (RalfConvRelayoutJob * len(jobs))()
    blk = 0
    for j, (w, o1, o2) in enumerate(jobs):
        Co, Ci, kh, kw = w.shape
        assert w.dtype == torch.float32 and w.is_contiguous() and o1.is_contiguous() and o2.is_contiguous() and kh * kw <= 49
        assert tuple(o1.shape[:3]) == (Co, kh, kw) and tuple(o2.shape) == (Ci, kh, kw, Co) and o1.dtype == o2.dtype
        r = arr[j]
        r.w, r.ohwi, r.ikwo = w.data_ptr(), o1.data_ptr(), o2.data_ptr()
        r.Co, r.Ci, r.KK, r.Cip, r.dst_dtype, r.first_block = Co, Ci, kh * kw, o1.shape[3], dtype_code(o1), blk
        ti = 64 if kh * kw <= 9 else max(1, 576 // (kh * kw))
        blk += max(1, min(512, ((Co + 7) // 8) * ((Ci + ti - 1) // ti)))
    return torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device), len(jobs), blk


def conv_relayout_batched(table, njobs, total_blocks):
    _call("ralf_conv_relayout_batched", _p(table), njobs, total_blocks)


def permute4_batched(table, njobs, total_blocks):
    _call("ralf_permute4_batched", _p(table), njobs, total_blocks)


def stem7x7_fwd(x8, w_ohwi, want_stats=True):
    """the stem convolution in direct form (ralf_stem7x7_fwd): x8 [B,IH,IW,8] bf16, w_ohwi [64,7,7,8] bf16 -> (y [B,OH,OW,64] bf16, partial statistics
    [tiles,2,64] fp32 or None)"""
    B, IH, IW, C = x8.shape
    assert C == 8 and x8.dtype == torch.bfloat16 and x8.is_contiguous() and w_ohwi.dtype == torch.bfloat16 and w_ohwi.numel() == 64 * 7 * 7 * 8 and w_ohwi.is_contiguous()
    OH, OW = (IH - 1) // 2 + 1, (IW - 1) // 2 + 1
    y = torch.empty(B, OH, OW, 64, dtype=torch.bfloat16, device=x8.device)
    part = torch.empty(B * OH * ((OW + 127) // 128), 2, 64, dtype=torch.float32, device=x8.device) if want_stats else None
    _call("ralf_stem7x7_fwd", _p(x8), _p(w_ohwi), _p(y), _p(part), B, IH, IW)
    return y, part


def stem7x7_wgrad(x8, dy, out=None, accumulate=False):
    """weight gradient of the stem convolution in direct form (ralf_stem7x7_wgrad): x8 [B,IH,IW,8], dy [B,OH,OW,64] bf16 -> fp32 OIHW [64, 4, 7, 7]"""
    B, IH, IW, C = x8.shape
    assert C == 8 and x8.dtype == torch.bfloat16 and dy.dtype == torch.bfloat16 and x8.is_contiguous() and dy.is_contiguous()
    assert tuple(dy.shape) == (B, (IH - 1) // 2 + 1, (IW - 1) // 2 + 1, 64)
    if out is None:
        out = torch.empty(64, 4, 7, 7, dtype=torch.float32, device=x8.device)
        accumulate = False
    assert out.dtype == torch.float32 and out.is_contiguous() and out.numel() == 64 * 4 * 49
    need = _lib.lib().ralf_stem7x7_wgrad_workspace_bytes(B, IH, IW)
    ws = workspace(need, x8.device)
    _call("ralf_stem7x7_wgrad", _p(x8), _p(dy), _p(out), B, IH, IW, int(accumulate), _p(ws), need)
    return out


def conv3x3_wgrad(dy, x, out=None, accumulate=False, stride=1):
    """weight gradient of a 3x3 / pad 1 convolution (stride 1 or 2) in the direct form (ralf_conv3x3_wgrad): dy [B,H,W,Co], x [B,IH,IW,Ci] NHWC bf16 ->
    fp32 OIHW [Co, Ci, 3, 3] (written into `out` when given)"""
    B, H, W, Co = dy.shape
    _, IH, IW, Ci = x.shape
    assert dy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and dy.is_contiguous() and x.is_contiguous() and x.shape[0] == B
    if out is None:
        out = torch.empty(Co, Ci, 3, 3, dtype=torch.float32, device=dy.device)
        accumulate = False
    assert out.dtype == torch.float32 and out.is_contiguous() and out.numel() == Co * Ci * 9
    L = _lib.lib()
    need = L.ralf_conv3x3_wgrad_workspace_bytes(B, H, W, Ci, Co)
    assert need > 0, "conv3x3_wgrad: shape not covered"
    ws = workspace(need, dy.device)
    _call("ralf_conv3x3_wgrad", _p(dy), _p(x), _p(out), B, H, W, IH, IW, stride, Ci, Co, int(accumulate), _p(ws), need)
    return out


def conv3x3_wgrad_supported(dy, x, stride, pad, kh, kw) -> bool:
    B, H, W, Co = dy.shape
    IH, IW = x.shape[1], x.shape[2]
    grid_ok = (IH == H and IW == W) if stride == 1 else ((IH + 1) // 2 == H and (IW + 1) // 2 == W and W <= 32)
    return (kh == 3 and kw == 3 and stride in (1, 2) and pad == 1 and dy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and W in (8, 16, 32, 64)
            and H % (64 // W) == 0 and Co % 64 == 0 and x.shape[3] % 64 == 0 and x.shape[0] == B and grid_ok)


def bn_train_stats(x2d, gamma, beta, running_mean, running_var, counter, partials, eps=1e-5, momentum=0.1):
    """batch statistics of a BatchNorm input (from the producing convolution's epilogue partials when given) -> fp32 [4, C] = mean, rstd,
    scale, shift; updates the running statistics and num_batches_tracked like bn_forward(training=True)"""
    M, C = x2d.shape
    out = torch.empty(4, C, dtype=torch.float32, device=x2d.device)
    if partials is not None:
        _call("ralf_bn_stats_from_partials", _p(partials), partials.shape[0], _p(gamma), _p(beta), _p(running_mean), _p(running_var), _p(counter),
              _p(out[0]), _p(out[1]), _p(out[2]), _p(out[3]), M, C, eps, momentum, _p(workspace(1024 * 2 * C * 4, x2d.device)))
    else:
        _call("ralf_bn_batch_stats", dtype_code(x2d), _p(x2d), _p(gamma), _p(beta), _p(running_mean), _p(running_var), _p(counter),
              _p(out[0]), _p(out[1]), _p(out[2]), _p(out[3]), M, C, eps, momentum, _p(workspace(1024 * 2 * C * 4, x2d.device)))
    return out


def bn_relu_maxpool_fwd(y, scale, shift):
    """relu(y * scale + shift) -> 3x3 / 2 max-pool in one pass (ralf_bn_relu_maxpool_fwd); y NHWC"""
    B, H, W, C = y.shape
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    out = torch.empty(B, OH, OW, C, dtype=y.dtype, device=y.device)
    arg = torch.empty(B, OH, OW, C, dtype=torch.int8, device=y.device)
    _call("ralf_bn_relu_maxpool_fwd", dtype_code(y), _p(y), _p(scale), _p(shift), _p(out), _p(arg), B, H, W, C)
    return out, arg


def bn_relu_maxpool_bwd(dpool, arg, y, stats, gamma, into):
    """backward of bn_relu_maxpool_fwd (batch statistics): -> (dy, dgamma, dbeta); into = (dgamma, dbeta) zeroed flat-gradient views or None"""
    B, H, W, C = y.shape
    dev = y.device
    nblk = 1024
    part = torch.empty(nblk, 2, C, dtype=torch.float32, device=dev)
    _call("ralf_bn_relu_maxpool_bwd_reduce", dtype_code(y), _p(dpool), _p(arg), _p(y), _p(stats[2]), _p(stats[3]), _p(stats[0]), _p(part), nblk, B, H, W, C)
    if into is not None:
        dg, db = into
    else:
        st = torch.zeros(2, C, dtype=torch.float32, device=dev)
        dg, db = st[0], st[1]
    coef = torch.empty(3, C, dtype=torch.float32, device=dev)
    _call("ralf_bn_bwd_stats_from_partials", _p(part), nblk, _p(stats[1]), _p(db), _p(dg), C, _p(workspace(128 * 2 * C * 4, dev)),
          _p(gamma), _p(stats[0]), B * H * W, _p(coef))
    dy = torch.empty_like(y)
    _call("ralf_bn_relu_maxpool_bwd_apply", dtype_code(y), _p(dpool), _p(arg), _p(y), _p(stats[2]), _p(stats[3]), _p(coef[0]), _p(coef[1]), _p(coef[2]), _p(dy), B, H, W, C)
    return dy, dg, db


def maxpool_fwd(x):
    B, H, W, C = x.shape
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty(B, OH, OW, C, dtype=x.dtype, device=x.device)
    arg = torch.empty(B, OH, OW, C, dtype=torch.int8, device=x.device)
    _call("ralf_maxpool3x3s2_fwd", dtype_code(x), _p(x), _p(y), _p(arg), B, H, W, C)
    return y, arg


def maxpool_bwd(dy, arg, in_shape):
    B, H, W, C = in_shape
    dx = torch.empty(in_shape, dtype=dy.dtype, device=dy.device)
    _call("ralf_maxpool3x3s2_bwd", dtype_code(dy), _p(dy), _p(arg), _p(dx), B, H, W, C)
    return dx


def upsample_add(src, lateral):
    B, IH, IW, C = src.shape
    _, OH, OW, _ = lateral.shape
    up = torch.empty_like(lateral)
    s = torch.empty_like(lateral)
    _call("ralf_upsample_nearest_add", dtype_code(src), _p(src), _p(lateral), _p(up), C, _p(s), B, IH, IW, OH, OW, C)
    return up, s


def upsample_bwd(g_up, g_sum, src_shape):
    B, IH, IW, C = src_shape
    _, OH, OW, _ = g_sum.shape
    d = torch.empty(src_shape, dtype=g_sum.dtype, device=g_sum.device)
    _call("ralf_upsample_nearest_bwd", dtype_code(g_sum), _p(g_up), C, _p(g_sum), _p(d), B, IH, IW, OH, OW, C)
    return d


_SKINNY_ROWS = int(os.environ.get("RALF_GEMM_SKINNY_ROWS", "512"))


def gemm_ln_ok(A: torch.Tensor, M: int, K: int) -> bool:
    """RalfGemmDesc.ln_*: the LayerNorm -> linear pairs of a decode step as ONE launch (bf16 rows of 256, at most 512 of them)"""
    return (_LN_GEMM and A.dtype == torch.bfloat16 and K == 256 and M <= min(512, _SKINNY_ROWS) and A.stride(-1) == 1 and A.data_ptr() % 16 == 0)


_LN_GEMM = os.environ.get("RALF_LN_GEMM", "1") != "0"   # A/B runs


def conv1x1_k64_ok(x2d: torch.Tensor, N: int, any_k: bool = False) -> bool:
    """ralf_conv1x1_k64: 64 input channels (the entry also takes 128 -- any_k -- where it is level with the tiled product: 28 against 32 us
    with statistics, 26 against 26 without, so the model does not route those), bf16 rows, M % 64 == 0, N = 64 * 2^k <= 2048"""
    M, K = x2d.shape
    n64 = N // 64
    return (_CONV1X1_K64 and x2d.is_cuda and x2d.dtype == torch.bfloat16 and (K == 64 or (any_k and K == 128)) and x2d.is_contiguous() and M % 64 == 0 and N % 64 == 0 and N <= 2048
            and (n64 & (n64 - 1)) == 0 and x2d.data_ptr() % 16 == 0)


_CONV1X1_K64 = os.environ.get("RALF_CONV1X1_K64", "1") != "0"   # A/B runs


def conv1x1_k64(x2d, W, colstats=None, scale=None, shift=None, res=None, relu=0, out=None):
    """y [M, N] = x [M, 64] W [N, 64]^T (ralf_conv1x1_k64: one wave per 64 x 64 tile, weights in registers, no LDS for the operands).
    colstats: the [M/64, 2, N] statistics partials of the following BatchNorm; scale / shift (+ res, relu 0 / 1 / 2 = after the residual): the
    inference epilogue."""
    M, K, N = x2d.shape[0], x2d.shape[1], W.shape[0]
    assert W.dtype == torch.bfloat16 and W.is_contiguous() and W.numel() == N * K
    y = out if out is not None else torch.empty(M, N, dtype=x2d.dtype, device=x2d.device)
    _call("ralf_conv1x1_k64", _p(x2d), _p(W), _p(y), _p(colstats), _p(scale), _p(shift), _p(res), int(relu), M, N, K)
    return y


def colstats_buffer(M, N, device):
    return torch.empty((M + 63) // 64, 2, N, dtype=torch.float32, device=device)


def bn_forward(x2d, gamma, beta, running_mean, running_var, training, relu, res, eps=1e-5, momentum=0.1, counter=None, partials=None, want_mask=False):
    """counter: the module's num_batches_tracked (int64 scalar tensor), bumped on the device in training mode.
    partials: column statistics the producing GEMM already wrote (ops.gemm(colstats=...)): no pass over x for the statistics."""
    M, C = x2d.shape
    dev = x2d.device
    out = torch.empty(4, C, dtype=torch.float32, device=dev)  # mean, rstd, scale, shift
    dt = dtype_code(x2d)
    if training and partials is not None:
        _call("ralf_bn_stats_from_partials", _p(partials), partials.shape[0], _p(gamma), _p(beta), _p(running_mean), _p(running_var), _p(counter),
              _p(out[0]), _p(out[1]), _p(out[2]), _p(out[3]), M, C, eps, momentum, _p(workspace(1024 * 2 * C * 4, dev)))
    elif training:
        _call("ralf_bn_batch_stats", dt, _p(x2d), _p(gamma), _p(beta), _p(running_mean), _p(running_var), _p(counter),
              _p(out[0]), _p(out[1]), _p(out[2]), _p(out[3]), M, C, eps, momentum, _p(workspace(1024 * 2 * C * 4, dev)))
    else:
        _call("ralf_bn_finalize", None, None, _p(gamma), _p(beta), _p(running_mean), _p(running_var),
              _p(out[0]), _p(out[1]), _p(out[2]), _p(out[3]), M, C, eps, momentum, 0)
    y = torch.empty_like(x2d)
    # ReLU mask bits (1/16 of y's bytes): what the backward kernels read instead of y
    mask = torch.empty(M * C // 8, dtype=torch.uint8, device=dev) if (relu and want_mask) else None
    _call("ralf_bn_apply", dt, _p(x2d), _p(out[2]), _p(out[3]), _p(res), _p(y), _p(mask), M, C, int(relu))
    if want_mask:
        return y, out[0], out[1], mask
    return y, out[0], out[1]


def bn_backward(x2d, dy, y, gamma, mean, rstd, relu, want_dres, training=True, into=None, mask=None, partials=None):
    """training: batch-statistics backward.  eval: statistics are constants -> dx = gamma*rstd*g
    (the same apply kernel with zero reduction terms); dgamma/dbeta are the same sums either way.
    partials: dy is the output of a data-gradient GEMM that already masked it and wrote the reductions per 64-row block
    (ops.gemm(bnb=...)): no pass over dy and x for the sums, and the residual branch's gradient is dy itself."""
    M, C = x2d.shape
    dt = dtype_code(x2d)
    if into is not None:   # (dgamma, dbeta) flat-gradient views, zero at this point: reduce straight into them
        s = (into[1], into[0])
    else:
        st = torch.zeros(2, C, dtype=torch.float32, device=x2d.device)
        s = (st[0], st[1])
    if partials is not None:
        _call("ralf_bn_bwd_stats_from_partials", _p(partials), partials.shape[0], _p(rstd), _p(s[0]), _p(s[1]), C, _p(workspace(128 * 2 * C * 4, x2d.device)),
              None, None, 0, None)
        dx = torch.empty_like(x2d)
        _call("ralf_bn_bwd_apply", dt, _p(x2d), _p(dy), None, None, _p(mean), _p(rstd), _p(gamma), _p(s[0]), _p(s[1]), _p(dx), None, M, C, 0)
        return dx, s[1], s[0], (dy if want_dres else None)
    _call("ralf_bn_bwd_reduce", dt, _p(x2d), _p(dy), _p(y), _p(mask), _p(mean), _p(rstd), _p(s[0]), _p(s[1]), M, C, int(relu), _p(workspace(1024 * 2 * C * 4, x2d.device)))
    dx = torch.empty_like(x2d)
    dres = torch.empty_like(x2d) if want_dres else None
    t = s if training else (torch.zeros_like(s[0]), torch.zeros_like(s[1]))
    _call("ralf_bn_bwd_apply", dt, _p(x2d), _p(dy), _p(y), _p(mask), _p(mean), _p(rstd), _p(gamma), _p(t[0]), _p(t[1]), _p(dx), _p(dres), M, C, int(relu))
    return dx, s[1], s[0], dres  # dx, dgamma, dbeta, dres


def _attn_desc(q, k, v, o, B, H, Sq, Sk, dh, q_off, k_off, v_off, causal, kpm, scale, p_drop, seed, call_id, kv_rows=None):
    d = RalfAttnDesc()
    es = q.element_size()

    def base(t, off):
        return ctypes.c_void_p(t.data_ptr() + off * es)

    d.q, d.k, d.v, d.o = base(q, q_off), base(k, k_off), base(v, v_off), _p(o)
    d.q_rs, d.q_bs = q.shape[-1], q.shape[-1] * Sq
    kvr = kv_rows if kv_rows is not None else Sk   # rows allocated per batch entry (a KV cache is longer than Sk)
    d.k_rs, d.k_bs = k.shape[-1], k.shape[-1] * kvr
    d.v_rs, d.v_bs = v.shape[-1], v.shape[-1] * kvr
    d.o_rs, d.o_bs = o.shape[-1], o.shape[-1] * Sq
    d.B, d.H, d.Sq, d.Sk, d.dh, d.dtype, d.causal = B, H, Sq, Sk, dh, dtype_code(q), int(causal)
    d.kpm, d.seed, d.call_id = _p(kpm), _p(seed), call_id
    d.scale, d.p_drop = scale, p_drop
    return d


def attention_fwd(q, k, v, B, H, Sq, Sk, dh, q_off=0, k_off=0, v_off=0, causal=False, kpm=None, scale=None, p_drop=0.0, seed=None, call_id=0, need_lse=True,
                  kv_rows=None, kpm_stride=None):
    """q/k/v: contiguous [B,S,width] tensors (possibly the same packed buffer); head h of q lives at
    column q_off + h*dh.  Returns O [B,Sq,H*dh] and lse [B,H,Sq]."""
    o = torch.empty(B, Sq, H * dh, dtype=q.dtype, device=q.device)
    lse = torch.empty(B, H, Sq, dtype=torch.float32, device=q.device) if need_lse else None
    d = _attn_desc(q, k, v, o, B, H, Sq, Sk, dh, q_off, k_off, v_off, causal, kpm, scale if scale is not None else dh ** -0.5, p_drop, seed, call_id, kv_rows)
    d.lse = _p(lse)
    d.kpm_bs = int(kpm_stride) if kpm_stride else 0   # kpm rows longer than Sk: one [B, max_len] mask for a growing prefix
    _call("ralf_attention_fwd", ctypes.byref(d))
    return o, lse


TLAYER_MAX_ROWS = 64
TLAYER_PACK_MAX = 96   # matrices per ralf_tlayer_pack launch


def tlayer_strip(rows: int) -> int:
    """rows per strip of the strip-wise layer kernels: 64, or 32 where 64 would leave most of the 256 CUs without a workgroup"""
    return 32 if (rows < 8192 and rows % 32 == 0 and not _STRIP64) else 64


_STRIP64 = os.environ.get("RALF_TLAYER_STRIP64", "0") == "1"   # A/B runs: 64-row strips (per-sample part 2) everywhere


def tlayer_pack(mats, transpose=()):
    """row-major bf16 matrices [N % 32 == 0, K % 16 == 0] -> the fragment order ralf_tlayer_fwd streams its weights in (ralf_tlayer_pack):
    ONE launch and one buffer for all of them; returns the flat packed views in order.  transpose: indices of the matrices to pack as
    their transpose (the data-gradient products read W^T)."""
    from ._abi import RalfPackJob

    assert 0 < len(mats) <= TLAYER_PACK_MAX
    sizes = [m.shape[0] * m.shape[1] for m in mats]
    buf = torch.empty(sum(sizes), dtype=torch.bfloat16, device=mats[0].device)
    outs, off = [], 0
    jobs = (RalfPackJob * len(mats))()
    for i, m in enumerate(mats):
        assert m.dtype == torch.bfloat16 and m.dim() == 2 and m.stride(1) == 1 and m.shape[0] % 32 == 0 and m.shape[1] % 32 == 0
        outs.append(buf[off:off + sizes[i]])
        tr = i in transpose
        jobs[i].src, jobs[i].dst, jobs[i].ld = m.data_ptr(), outs[i].data_ptr(), m.stride(0)
        jobs[i].N, jobs[i].K, jobs[i].transpose = (m.shape[1], m.shape[0], 1) if tr else (m.shape[0], m.shape[1], 0)
        off += sizes[i]
    _call("ralf_tlayer_pack", jobs, len(mats))
    return outs


def tlayer_fwd(x, W, *, causal, kpm=None, kpm_stride=0, kv=None, p_attn=0.0, p_res=0.0, seed=None, calls=(0, 0, 0, 0, 0, 0), eps=1e-5, kv_ready=None):
    """one pre-norm transformer layer forward on SHORT sequences (ralf_tlayer_fwd): x [B, S <= 64, 256] bf16.
    W: dict of pairs -- LayerNorms "ln1", "ln3" (and "ln2") = (gamma, beta) fp32; linear layers "sa_in" [768, 256], "sa_out" [256, 256],
    "ffn1" [1024, 256], "ffn2" [256, 1024] (and "q_proj" [256, 256], "out2" [256, 256]) = (weight PACKED by tlayer_pack, fp32 bias).
    kv None: encoder layer, one launch.  kv [B, M, 512] (the memory's k | v projections): decoder layer = part 1, ralf_attention_fwd, part 2.
    calls = dropout call ids (self-attention, its out-projection, cross-attention, its out-projection, ffn1, ffn2).
    Returns the dict of everything the unfused backward reads: h1 mean1 rstd1 qkv o1 lse1 x1 [h2 mean2 rstd2 q o2 lse2 x2] h3 mean3 rstd3 hid out."""
    from ._abi import RalfTLayerDesc

    B, S, dm = x.shape
    H, ff = 8, 1024
    cross = kv is not None
    assert x.dtype == torch.bfloat16 and x.is_contiguous() and dm == 256 and 1 <= S <= TLAYER_MAX_ROWS
    for name, n in (("sa_in", 3 * dm * dm), ("sa_out", dm * dm), ("ffn1", ff * dm), ("ffn2", ff * dm)) + ((("q_proj", dm * dm), ("out2", dm * dm)) if cross else ()):
        w, b = W[name]
        assert w.dtype == torch.bfloat16 and w.is_contiguous() and w.numel() == n and b.dtype == torch.float32 and b.is_contiguous(), name
    dev = x.device

    def act(width=dm):
        return torch.empty(B, S, width, dtype=torch.bfloat16, device=dev)

    def stat():
        return torch.empty(B * S, dtype=torch.float32, device=dev)

    t = {"h1": act(), "mean1": stat(), "rstd1": stat(), "qkv": act(3 * dm), "o1": act(), "lse1": torch.empty(B, H, S, dtype=torch.float32, device=dev), "x1": act(),
         "h3": act(), "mean3": stat(), "rstd3": stat(), "hid": act(ff), "out": act()}
    if cross:
        assert kv.dtype == torch.bfloat16 and kv.is_contiguous() and kv.shape[0] == B and kv.shape[2] == 2 * dm
        t.update({"h2": act(), "mean2": stat(), "rstd2": stat(), "q": act(), "x2": act()})
    d = RalfTLayerDesc()
    d.x = _p(x)
    d.ln1_g, d.ln1_b, d.w_in, d.b_in, d.w_o, d.b_o = _p(W["ln1"][0]), _p(W["ln1"][1]), _p(W["sa_in"][0]), _p(W["sa_in"][1]), _p(W["sa_out"][0]), _p(W["sa_out"][1])
    d.ln3_g, d.ln3_b, d.w1, d.b1, d.w2, d.b2 = _p(W["ln3"][0]), _p(W["ln3"][1]), _p(W["ffn1"][0]), _p(W["ffn1"][1]), _p(W["ffn2"][0]), _p(W["ffn2"][1])
    if cross:
        d.ln2_g, d.ln2_b, d.w_q, d.b_q, d.w_o2, d.b_o2 = _p(W["ln2"][0]), _p(W["ln2"][1]), _p(W["q_proj"][0]), _p(W["q_proj"][1]), _p(W["out2"][0]), _p(W["out2"][1])
    d.kpm, d.kpm_bs = _p(kpm), int(kpm_stride) if kpm_stride else (S if kpm is not None else 0)
    for k, v in t.items():
        setattr(d, k, _p(v))
    d.seed = _p(seed) if (p_attn > 0.0 or p_res > 0.0) else None
    d.call_attn1, d.call_out1, d.call_out2, d.call_ffn1, d.call_ffn2 = int(calls[0]), int(calls[1]), int(calls[3]), int(calls[4]), int(calls[5])
    d.B, d.S, d.causal = B, S, int(bool(causal))
    d.scale, d.p_attn, d.p_res, d.eps = (dm // H) ** -0.5, float(p_attn), float(p_res), eps
    if not cross:
        d.part = 0
        _call("ralf_tlayer_fwd", ctypes.byref(d))
        return t
    d.part = 1
    _call("ralf_tlayer_fwd", ctypes.byref(d))
    if kv_ready is not None:   # kv comes from another stream (functional.Runtime.ahead): part 1 did not need it
        torch.cuda.current_stream().wait_event(kv_ready)
    t["o2"], t["lse2"] = attention_fwd(t["q"], kv, kv, B, H, S, kv.shape[1], dm // H, 0, 0, dm, p_drop=p_attn, seed=seed, call_id=int(calls[2]))
    d.o2, d.part = _p(t["o2"]), 2
    if (B * S) % 32 == 0 and B * S < 8192 and not _STRIP64:   # part 2 is row-wise: 32-row strips of all samples' rows put twice as many (lighter) workgroups on the chip
        d.B, d.S = B * S // 32, 32
    _call("ralf_tlayer_fwd", ctypes.byref(d))
    return t


def tlayer_tail(o2, x1, W, rows_per_strip=None, eps=1e-5):
    """inference tail of a decoder layer on rows [R, 256] bf16 (ralf_tlayer_fwd part 2, nothing kept for a backward pass):
    r = x1 + o2 Wo2^T + bo2;  out = r + W2 relu(W1 LN3(r) + b1) + b2.  W: "out2", "ffn1", "ffn2" = (weight packed by tlayer_pack, fp32 bias),
    "ln3" = (gamma, beta).  The R rows are cut into strips of rows_per_strip (default: the largest of 32, 16, .. 1 that divides R)."""
    from ._abi import RalfTLayerDesc

    R, dm = o2.shape
    assert dm == 256 and o2.dtype == torch.bfloat16 and x1.dtype == torch.bfloat16 and o2.is_contiguous() and x1.is_contiguous() and x1.shape == o2.shape
    S = rows_per_strip or next(s for s in (32, 16, 8, 4, 2, 1) if R % s == 0)
    assert R % S == 0 and 1 <= S <= TLAYER_MAX_ROWS
    out, x2 = torch.empty_like(x1), torch.empty_like(x1)
    d = RalfTLayerDesc()
    d.x1, d.o2, d.x2, d.out = _p(x1), _p(o2), _p(x2), _p(out)
    d.w_o2, d.b_o2 = _p(W["out2"][0]), _p(W["out2"][1])
    d.ln3_g, d.ln3_b, d.w1, d.b1, d.w2, d.b2 = _p(W["ln3"][0]), _p(W["ln3"][1]), _p(W["ffn1"][0]), _p(W["ffn1"][1]), _p(W["ffn2"][0]), _p(W["ffn2"][1])
    d.B, d.S, d.part, d.eps = R // S, S, 2, eps
    _call("ralf_tlayer_fwd", ctypes.byref(d))
    return out


def tlayer_ffn(x, W, *, o=None, p=0.0, seed=None, calls=(0, 0, 0), rows_per_strip=None, eps=1e-5, act="relu", residual=True):
    """the tail of a pre-norm layer on ANY row count, in strips of rows_per_strip rows (rows % rows_per_strip == 0), ralf_tlayer_fwd part 3 / 2:
        o is None:  out = x + drop(W2 drop(relu(W1 LN(x) + b1)) + b2)                                   (the feed-forward block)
        o given:    r = x + drop(o Wo^T + bo);  out = r + drop(W2 drop(relu(W1 LN(r) + b1)) + b2)       (+ the attention's out-projection in front)
    x, o [..., 256] bf16; W: "ln3" (gamma, beta), "ffn1" / "ffn2" (and "out" with o) = (weight packed by tlayer_pack, fp32 bias);
    calls = dropout call ids (out-projection, ffn1, ffn2).  act="gelu", residual=False (o None only): out = W2 gelu(W1 LN(x) + b1) + b2, the
    reference's FeedForward, with the pre-activation z kept for the GELU gradient.  Returns what the unfused Functions would have saved plus
    the output: [x2 (= r),] h3, mean3, rstd3, hid, [z,] out."""
    from ._abi import RalfTLayerDesc

    shape = x.shape
    rows, dm, ff = x.numel() // shape[-1], shape[-1], 1024
    S = int(rows_per_strip) if rows_per_strip else tlayer_strip(rows)
    assert dm == 256 and x.dtype == torch.bfloat16 and x.is_contiguous() and rows % S == 0 and 1 <= S <= TLAYER_MAX_ROWS
    dev = x.device
    t = {"h3": torch.empty(shape, dtype=torch.bfloat16, device=dev), "mean3": torch.empty(rows, dtype=torch.float32, device=dev),
         "rstd3": torch.empty(rows, dtype=torch.float32, device=dev), "hid": torch.empty(*shape[:-1], ff, dtype=torch.bfloat16, device=dev),
         "out": torch.empty(shape, dtype=torch.bfloat16, device=dev)}
    d = RalfTLayerDesc()
    d.ln3_g, d.ln3_b, d.w1, d.b1, d.w2, d.b2 = _p(W["ln3"][0]), _p(W["ln3"][1]), _p(W["ffn1"][0]), _p(W["ffn1"][1]), _p(W["ffn2"][0]), _p(W["ffn2"][1])
    assert act in ("relu", "gelu") and (o is None or (act == "relu" and residual))
    if act == "gelu":
        t["z"] = torch.empty(*shape[:-1], ff, dtype=torch.bfloat16, device=dev)
        d.act = 2
    d.no_res = 0 if residual else 1
    if o is None:
        d.x, d.part = _p(x), 3
    else:
        assert o.dtype == torch.bfloat16 and o.is_contiguous() and o.shape == x.shape
        t["x2"] = torch.empty(shape, dtype=torch.bfloat16, device=dev)
        d.x1, d.o2, d.w_o2, d.b_o2, d.part = _p(x), _p(o), _p(W["out"][0]), _p(W["out"][1]), 2
    for k, v in t.items():
        setattr(d, k, _p(v))
    d.seed = _p(seed) if p > 0.0 else None
    d.call_out2, d.call_ffn1, d.call_ffn2 = int(calls[0]), int(calls[1]), int(calls[2])
    d.B, d.S = rows // S, S
    d.p_res, d.eps = float(p), eps
    _call("ralf_tlayer_fwd", ctypes.byref(d))
    return t


def tlayer_lnqkv(x, W, rows_per_strip=64, eps=1e-5):
    """h1 = LayerNorm(x); qkv = h1 Win^T + bin on strips of rows_per_strip rows of ANY [rows, 256] bf16 tensor (ralf_tlayer_fwd part 4).
    W: "ln1" (gamma, beta), "sa_in" (in_proj weight packed by tlayer_pack, fp32 bias).  Returns h1, mean1, rstd1, qkv."""
    from ._abi import RalfTLayerDesc

    shape = x.shape
    rows, dm, S = x.numel() // shape[-1], shape[-1], int(rows_per_strip)
    assert dm == 256 and x.dtype == torch.bfloat16 and x.is_contiguous() and rows % S == 0 and 1 <= S <= TLAYER_MAX_ROWS
    dev = x.device
    t = {"h1": torch.empty(shape, dtype=torch.bfloat16, device=dev), "mean1": torch.empty(rows, dtype=torch.float32, device=dev),
         "rstd1": torch.empty(rows, dtype=torch.float32, device=dev), "qkv": torch.empty(*shape[:-1], 3 * dm, dtype=torch.bfloat16, device=dev)}
    d = RalfTLayerDesc()
    d.x, d.ln1_g, d.ln1_b, d.w_in, d.b_in = _p(x), _p(W["ln1"][0]), _p(W["ln1"][1]), _p(W["sa_in"][0]), _p(W["sa_in"][1])
    for k, v in t.items():
        setattr(d, k, _p(v))
    d.B, d.S, d.part, d.eps = rows // S, S, 4, eps
    _call("ralf_tlayer_fwd", ctypes.byref(d))
    return t


def tlayer_bwd(dy_m, hid, Wt, *, p=0.0, dy=None, x2=None, mean3=None, rstd3=None, gamma=None, dgamma=None, dbeta=None, seed=None, call_out=0,
               rows_per_strip=None, gelu=False):
    """data gradients of the strip-wise layer tail (ralf_tlayer_bwd): dz = (dy_m W2) o [hid > 0] / (1 - p), dh = dz W1 -- and, with the
    LayerNorm operands (dy, x2, mean3, rstd3, gamma), g = LN-backward(dh) + dy, g_m = g masked by (p, call_out), d_o = g_m Wo.
    gelu=True (FeedForward): hid = the pre-activation z, dz = (dy_m W2) o gelu'(z), no skip gradient needed, no d_o.
    Wt: "w2t", "w1t" (and "wot") = tlayer_pack(.., transpose) of linear2 / linear1 / out_proj weights.  Returns dz and dh (stage 1) or dz, g, g_m, d_o."""
    from ._abi import RalfTLayerBwdDesc

    shape = dy_m.shape
    rows = dy_m.numel() // shape[-1]
    S = int(rows_per_strip) if rows_per_strip else tlayer_strip(rows)
    assert shape[-1] == 256 and dy_m.dtype == torch.bfloat16 and dy_m.is_contiguous() and hid.is_contiguous() and rows % S == 0
    full = x2 is not None
    dev = dy_m.device
    t = {"dz": torch.empty(*shape[:-1], 1024, dtype=torch.bfloat16, device=dev), "g": torch.empty(shape, dtype=torch.bfloat16, device=dev)}
    d = RalfTLayerBwdDesc()
    d.dy_m, d.hid, d.w2t, d.w1t = _p(dy_m), _p(hid), _p(Wt["w2t"]), _p(Wt["w1t"])
    if full:
        t["g_m"] = torch.empty(shape, dtype=torch.bfloat16, device=dev) if p > 0.0 else t["g"]
        if not gelu:
            t["d_o"] = torch.empty(shape, dtype=torch.bfloat16, device=dev)
        d.dy, d.x2, d.mean3, d.rstd3, d.ln3_g, d.wot = _p(dy), _p(x2), _p(mean3), _p(rstd3), _p(gamma), _p(Wt.get("wot"))
        d.dgamma, d.dbeta, d.seed, d.call_out = _p(dgamma), _p(dbeta), _p(seed) if p > 0.0 else None, int(call_out)
    for k, v in t.items():
        setattr(d, k, _p(v))
    assert not gelu or full
    d.B, d.S, d.stage, d.p = rows // S, S, (5 if gelu else 3) if full else 1, float(p)
    _call("ralf_tlayer_bwd", ctypes.byref(d))
    return t


def tlayer_bwd_lnqkv(dqkv, win_t, x, mean, rstd, gamma, *, skip=None, dgamma=None, dbeta=None, p=0.0, seed=None, call=0, rows_per_strip=None, wo_t=None):
    """backward of tlayer_lnqkv's data path (ralf_tlayer_bwd stage 4): dh = dqkv Win, dx = LN-backward(dh; x, mean, rstd, gamma) + skip, and
    (p > 0) dx masked by the dropout (p, call) of the block that produced x.  win_t = tlayer_pack([in_proj_weight], transpose=(0,)).
    dqkv may also be the [rows, 256] gradient of a 256 -> 256 projection (win_t = that weight's transpose).  wo_t (the packed transpose of the
    out-projection that produced x's last addend): also d_o = dx_masked Wo.  Returns (dx, dx_masked or None[, d_o]); dgamma / dbeta are accumulated into."""
    from ._abi import RalfTLayerBwdDesc

    shape = x.shape
    rows = x.numel() // shape[-1]
    S = int(rows_per_strip) if rows_per_strip else tlayer_strip(rows)
    nk = dqkv.numel() // (rows * 256)
    assert shape[-1] == 256 and dqkv.dtype == torch.bfloat16 and dqkv.is_contiguous() and dqkv.numel() == rows * 256 * nk and nk in (1, 3) and x.is_contiguous() and rows % S == 0
    dx = torch.empty(shape, dtype=torch.bfloat16, device=x.device)
    dxm = torch.empty(shape, dtype=torch.bfloat16, device=x.device) if p > 0.0 else None
    d = RalfTLayerBwdDesc()
    d.dy_m, d.w1t, d.x2, d.mean3, d.rstd3, d.ln3_g, d.dy = _p(dqkv), _p(win_t), _p(x), _p(mean), _p(rstd), _p(gamma), _p(skip)
    d.g, d.g_m, d.dgamma, d.dbeta = _p(dx), _p(dxm if dxm is not None else dx), _p(dgamma), _p(dbeta)
    d.seed, d.call_out = _p(seed) if p > 0.0 else None, int(call)
    d.B, d.S, d.stage, d.p, d.nk = rows // S, S, 4, float(p), nk
    d_o = None
    if wo_t is not None:
        d_o = torch.empty(shape, dtype=torch.bfloat16, device=x.device)
        d.wot, d.d_o = _p(wo_t), _p(d_o)
    _call("ralf_tlayer_bwd", ctypes.byref(d))
    return (dx, dxm) if wo_t is None else (dx, dxm, d_o)


_DEC_MAXK = None


def decode_attn_max_keys() -> int:
    """most keys ralf_decode_attn takes (its scores live in LDS)"""
    global _DEC_MAXK
    if _DEC_MAXK is None:
        _DEC_MAXK = int(_lib.lib().ralf_decode_attn_max_keys())
    return _DEC_MAXK


def decode_attn(x, ln_g, ln_b, W, bias, kv, Sk, H, self_attn, kpm=None, kpm_stride=0, eps=1e-5, pos=None, packed_rows=0):
    """one attention block of a KV-cached decode step with LayerNorm and the q (k, v) projections inside (ralf_decode_attn):
    x [B, d] bf16, W bf16 [3d, d] (in_proj_weight), bias fp32 [3d], kv bf16 cache [B, rows, 2d] -> o [B, d].
    pos (self-attention only): int32 [B] on the device, the cached rows PER ELEMENT (Sk is then their upper bound)."""
    from ._abi import RalfDecodeAttnDesc

    B, d_model = x.shape
    assert x.dtype == torch.bfloat16 and W.dtype == torch.bfloat16 and kv.dtype == torch.bfloat16
    o = torch.empty(B, d_model, dtype=x.dtype, device=x.device)
    d = RalfDecodeAttnDesc()
    d.x, d.ln_g, d.ln_b, d.W, d.bias, d.kv, d.kpm, d.o = _p(x), _p(ln_g), _p(ln_b), _p(W), _p(bias), _p(kv), _p(kpm), _p(o)
    if packed_rows:   # head-pair-major cross-attention cache [B, 2 (k | v), H/2, rows, 64] (nn.decoder_init_cache)
        R = int(packed_rows)
        assert not self_attn and kv.is_contiguous() and kv.numel() == B * 2 * (H // 2) * R * 64
        d.x_rs, d.kv_bs, d.kv_rs, d.kpm_bs, d.o_rs = x.stride(0), 2 * (H // 2) * R * 64, 64, int(kpm_stride), d_model
        d.kv_hs, d.kv_vo = R * 64, (H // 2) * R * 64
    else:
        assert kv.shape[2] == 2 * d_model
        d.x_rs, d.kv_bs, d.kv_rs, d.kpm_bs, d.o_rs = x.stride(0), kv.stride(0), kv.stride(1), int(kpm_stride), d_model
    d.B, d.H, d.d, d.Sk, d.self_ = B, H, d_model, int(Sk), int(bool(self_attn))
    d.scale, d.eps = (d_model // H) ** -0.5, eps
    if pos is not None:
        assert self_attn and pos.dtype == torch.int32 and pos.is_contiguous() and pos.numel() == B and pos.device == x.device
    d.pos = _p(pos)
    _call("ralf_decode_attn", ctypes.byref(d))
    return o


_DEC_TOKEN_LIMITS = None


def decode_token_limits():
    """(most cached positions, most memory rows) ralf_decode_token takes"""
    global _DEC_TOKEN_LIMITS
    if _DEC_TOKEN_LIMITS is None:
        a, b = ctypes.c_int(0), ctypes.c_int(0)
        _lib.lib().ralf_decode_token_limits(ctypes.byref(a), ctypes.byref(b))
        _DEC_TOKEN_LIMITS = (a.value, b.value)
    return _DEC_TOKEN_LIMITS


def decode_token(layers, head, emb, pe, emb_scale, tok, pos, self_kv, cross_kv, L, M, V, kpm=None, kpm_stride=0, pos_vec=None, eps=1e-5, keep=None, sample=None):
    """one KV-cached decode step of the whole decoder stack in ONE launch, a workgroup per sample (ralf_decode_token): token ids [B] -> fp32
    logits [B, V].  layers: per layer a dict of device tensors (bf16 weights row-major [n_out, n_in], fp32 biases / LayerNorm parameters):
    w_qkv b_qkv ln1_g ln1_b w_o1 b_o1 ln2_g ln2_b w_q2 b_q2 w_o2 b_o2 ln3_g ln3_b w_f1 b_f1 w_f2 b_f2; head = (ln_g, ln_b, w_head).
    sample (optional): the keyword arguments of mask_sample (allowed, forced, mode, top_k, temperature, seed, call_id, seq_col, pad_flag_col, pad_id, top_p,
    row0) -- the token choice runs in the same launch and the call returns (logits, tokens int64 [B])."""
    from ._abi import RalfDecodeTokenDesc

    B = tok.numel()
    logits = torch.empty(B, V, dtype=torch.float32, device=tok.device)
    d = RalfDecodeTokenDesc()
    d.tok, d.pos_vec, d.kpm, d.emb, d.pe = _p(tok), _p(pos_vec), _p(kpm), _p(emb), _p(pe)
    d.lnh_g, d.lnh_b, d.w_head, d.logits = _p(head[0]), _p(head[1]), _p(head[2]), _p(logits)
    d.kpm_bs, d.B, d.L, d.M, d.V, d.nlayers, d.pos = int(kpm_stride), B, int(L), int(M), int(V), len(layers), int(pos)
    d.emb_scale, d.eps = float(emb_scale), float(eps)
    for i, w in enumerate(layers):
        lw = d.layer[i]
        for n in ("w_qkv", "b_qkv", "ln1_g", "ln1_b", "w_o1", "b_o1", "ln2_g", "ln2_b", "w_q2", "b_q2", "w_o2", "b_o2", "ln3_g", "ln3_b", "w_f1", "b_f1", "w_f2", "b_f2"):
            setattr(lw, n, _p(w[n]))
        lw.self_kv, lw.cross_kv = _p(self_kv[i]), _p(cross_kv[i])
    out = None
    if sample is not None:
        out = torch.empty(B, dtype=torch.int64, device=tok.device)

        def col(t, dtype):
            if t is None:
                return None, 0
            assert t.dtype == dtype and t.dim() == 1 and t.shape[0] == B and t.stride(0) > 0
            return ctypes.c_void_p(t.data_ptr()), t.stride(0)

        sp, sl = col(sample.get("seq_col"), torch.int64)
        fp, fl = col(sample.get("pad_flag_col"), torch.uint8)
        d.s_allowed, d.s_forced, d.s_seed, d.s_out = _p(sample.get("allowed")), _p(sample.get("forced")), _p(sample.get("seed")), _p(out)
        d.s_seq_out, d.s_seq_ld, d.s_flag_out, d.s_flag_ld = sp, sl, fp, fl
        d.s_pad_id, d.s_call, d.s_row0 = int(sample.get("pad_id", -1)), int(sample.get("call_id", 0)), int(sample.get("row0", 0))
        d.s_mode, d.s_top_k = int(sample.get("mode", 0)), int(sample.get("top_k", 1))
        d.s_temperature, d.s_top_p = float(sample.get("temperature", 1.0)), float(sample.get("top_p", 1.0))
    _call("ralf_decode_token", ctypes.byref(d))
    return logits if sample is None else (logits, out)


def attention_bwd(dout, q, k, v, o, lse, dq, dk, dv, B, H, Sq, Sk, dh, q_off=0, k_off=0, v_off=0, dq_off=0, dk_off=0, dv_off=0,
                  causal=False, kpm=None, scale=None, p_drop=0.0, seed=None, call_id=0):
    d = _attn_desc(q, k, v, o, B, H, Sq, Sk, dh, q_off, k_off, v_off, causal, kpm, scale if scale is not None else dh ** -0.5, p_drop, seed, call_id)
    delta = torch.empty(B, H, Sq, dtype=torch.float32, device=q.device)
    es = q.element_size()
    d.lse, d.delta, d.dout = _p(lse), _p(delta), _p(dout)
    d.do_rs, d.do_bs = dout.shape[-1], dout.shape[-1] * Sq
    d.dq = ctypes.c_void_p(dq.data_ptr() + dq_off * es)
    d.dk = ctypes.c_void_p(dk.data_ptr() + dk_off * es)
    d.dv = ctypes.c_void_p(dv.data_ptr() + dv_off * es)
    d.dq_rs, d.dq_bs = dq.shape[-1], dq.shape[-1] * Sq
    d.dk_rs, d.dk_bs = dk.shape[-1], dk.shape[-1] * Sk
    d.dv_rs, d.dv_bs = dv.shape[-1], dv.shape[-1] * Sk
    _call("ralf_attention_bwd", ctypes.byref(d))


def sumsq(flat, out):
    _call("ralf_sumsq", _p(flat), flat.numel(), _p(out))


def clip_coef(sumsq_t, max_norm, coef, norm_out=None):
    _call("ralf_clip_coef", _p(sumsq_t), max_norm, _p(coef), _p(norm_out))


SUMSQ_PARTS = 1024   # RALF_SUMSQ_PARTS


def sumsq_partials(flat, partials):
    """deterministic: partials (fp32 [SUMSQ_PARTS]) <- per-workgroup sums of squares (no atomics)"""
    assert partials.numel() >= SUMSQ_PARTS and partials.dtype == torch.float32
    _call("ralf_sumsq_partials", _p(flat), flat.numel(), _p(partials))


def clip_coef_partials(partials, max_norm, coef, norm_out=None):
    _call("ralf_clip_coef_partials", _p(partials), max_norm, _p(coef), _p(norm_out))


def adamw(p, g, m, v, lr, beta1, beta2, eps, wd, step, coef=None, shadow=None, step_dev=None, lr_scale=None):
    _call("ralf_adamw", _p(p), _p(g), _p(m), _p(v), _p(shadow), p.numel(), lr, beta1, beta2, eps, wd, step, _p(coef), _p(step_dev), _p(lr_scale))


SAMPLING_MODES = {"deterministic": 0, "top_k": 1, "top_p": 2, "random": 3, "gumbel": 4}   # helpers/sampling.py:18-71 -> ralf_mask_sample's mode


def mask_sample(logits, allowed=None, forced=None, mode=0, top_k=1, temperature=1.0, seed=None, call_id=0,
                seq_col=None, pad_flag_col=None, pad_id=-1, top_p=1.0, row0=0):
    """decode-space mask + token choice on the device -> int64 [B].  seq_col / pad_flag_col: COLUMN views of the [B, L] int64
    sequence buffer / uint8 key-padding mask that also receive the token / (token == pad_id)."""
    B, V = logits.shape
    out = torch.empty(B, dtype=torch.int64, device=logits.device)
    if seq_col is None and pad_flag_col is None and row0 == 0:
        _call("ralf_mask_sample", _p(logits.contiguous()), _p(allowed), _p(forced), mode, top_k, temperature, _p(seed), call_id, _p(out), B, V, float(top_p))
        return out

    def col(t, dtype):
        if t is None:
            return None, 0
        assert t.dtype == dtype and t.dim() == 1 and t.shape[0] == B and t.stride(0) > 0
        return ctypes.c_void_p(t.data_ptr()), t.stride(0)

    sp, sl = col(seq_col, torch.int64)
    fp, fl = col(pad_flag_col, torch.uint8)
    _call("ralf_mask_sample_step", _p(logits.contiguous()), _p(allowed), _p(forced), mode, top_k, temperature, _p(seed), call_id, _p(out),
          sp, sl, fp, fl, int(pad_id), B, V, float(top_p), int(row0))
    return out
