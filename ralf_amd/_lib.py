"""ctypes loader for libralf_hip.so (the C-ABI boundary declared in include/ralf_hip.h).

The product path has NO CPU fallback: if the HIP library is missing or a call fails, a
RuntimeError is raised.  Build with `python -c "import __graft_entry__ as g; g.build()"` or
`make -C ralf_amd/csrc`.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RALF_HIP_LIB: another build of the same ABI, for same-box A/B timing of kernel changes (tools/ab_encdec.sh); never a fallback
LIB_PATH = os.environ.get("RALF_HIP_LIB") or os.path.join(_HERE, "libralf_hip.so")
_lib = None

c_i64, c_int, c_size, c_void, c_float = ctypes.c_int64, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_float


class RalfHipError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RalfHipError(f"{LIB_PATH} not found: build the HIP extension first (make -C ralf_amd/csrc); there is no CPU fallback")
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.ralf_last_error.restype = ctypes.c_char_p
        _declare(_lib)
    return _lib


def _declare(L):
    from ._abi import SIGNATURES

    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError here = header/library mismatch: fail loudly
        fn.restype = restype
        fn.argtypes = argtypes


def check(rc, what):
    if rc != 0:
        raise RalfHipError(f"{what} failed (code {rc}): {lib().ralf_last_error().decode()}")


def ptr(t):
    """device pointer of a contiguous torch tensor (or None)."""
    if t is None:
        return None
    assert t.is_contiguous(), "ralf_hip ops need contiguous tensors"
    return ctypes.c_void_p(t.data_ptr())


def stream_ptr():
    import torch

    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
