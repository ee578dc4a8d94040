"""Thin Python bindings of the C-ABI compute entry points (include/ralf_hip.h) on torch CUDA
tensors.  No arithmetic happens in Python/torch here: every function launches HIP kernels on
torch's current stream.  Autograd wiring lives in ralf_amd/functional.py."""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from . import _lib
from ._abi import RalfConvGeom, RalfGemmDesc

F32, BF16 = 0, 1
ACT = {None: 0, "none": 0, "relu": 1, "gelu": 2}
AUX = {None: 0, "relu_mask": 1, "gelu_grad": 2}
_TORCH2CODE = {torch.float32: F32, torch.bfloat16: BF16}

_ws_cache: dict = {}


def dtype_code(t: torch.Tensor) -> int:
    return _TORCH2CODE[t.dtype]


def workspace(nbytes: int, device) -> torch.Tensor:
    """grow-only scratch buffer per device (stream-ordered reuse: all ops run on one stream)."""
    key = (device.type, device.index)
    w = _ws_cache.get(key)
    if w is None or w.numel() < nbytes:
        w = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = w
    return w


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def gemm(A: torch.Tensor, B: torch.Tensor, M: int, N: int, K: int, *, a_kcontig=True, b_kcontig=True,
         lda=None, ldb=None, out: Optional[torch.Tensor] = None, ldc=None, out_dtype=None,
         bias=None, act=None, res=None, ldr=None, aux=None, aux_mode=None, aux_scale=1.0, out2=None,
         alpha=1.0, accumulate=False, splitk=1, batch=(1, 1), sA=(0, 0), sB=(0, 0), sC=(0, 0), sR=(0, 0),
         conv: Optional[dict] = None, gather=0) -> torch.Tensor:
    """C = epi(alpha * A @ B) through ralf_gemm (see include/ralf_hip.h: RalfGemmDesc)."""
    assert A.is_cuda and B.is_cuda and A.dtype == B.dtype
    d = RalfGemmDesc()
    d.dtype = dtype_code(A)
    nb0, nb1 = batch
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
    d.sR0, d.sR1 = sR if res is not None and sR != (0, 0) else sC
    d.a_kcontig, d.b_kcontig, d.gather = int(a_kcontig), int(b_kcontig), gather
    d.act, d.aux_mode, d.aux_scale = ACT[act], AUX[aux_mode], aux_scale
    d.accumulate, d.splitk, d.alpha = int(accumulate), splitk, alpha
    if conv is not None:
        g = RalfConvGeom(**conv)
        d.g = g
    L = _lib.lib()
    ws, wsn = None, 0
    if splitk > 1:
        wsn = splitk * nb0 * nb1 * M * N * 4
        ws = workspace(wsn, A.device)
    rc = L.ralf_gemm(ctypes.byref(d), _p(ws), wsn, _lib.stream_ptr())
    _lib.check(rc, "ralf_gemm")
    return out
