"""ctypes signatures of every symbol declared in include/ralf_hip.h (kept in the same order).
tests/test_abi.py checks that the header, this table and the built library agree."""
import ctypes

i64, i32, sz, vp, f32 = ctypes.c_int64, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_float

SIGNATURES = {
    "ralf_last_error": (ctypes.c_char_p, []),
    "ralf_abi_version": (i32, []),
    # exact inner-product top-k
    "ralf_knn_topk_ip_workspace_bytes": (sz, [i64, i32, i32, i32]),
    "ralf_knn_topk_ip": (i32, [vp, i64, i32, vp, i32, i32, vp, vp, vp, sz, vp]),
    "ralf_knn_scores": (i32, [vp, i64, i32, vp, i32, vp, vp]),
    "ralf_knn_select": (i32, [vp, i64, i32, i32, vp, vp, vp, sz, vp]),
}


class RalfConvGeom(ctypes.Structure):
    _fields_ = [(n, i32) for n in ("RH", "RW", "SH", "SW", "SC", "KH", "KW", "stride", "pad", "mode")]


class RalfGemmDesc(ctypes.Structure):
    _fields_ = (
        [(n, vp) for n in ("A", "B", "C", "C2", "bias", "res", "aux")]
        + [(n, i64) for n in ("lda", "ldb", "ldc", "ldr", "sA0", "sA1", "sB0", "sB1", "sC0", "sC1", "sR0", "sR1")]
        + [(n, i32) for n in ("M", "N", "K", "nb0", "nb1", "dtype", "a_kcontig", "b_kcontig", "gather",
                              "act", "aux_mode", "out_f32", "accumulate", "splitk")]
        + [("alpha", f32), ("aux_scale", f32), ("g", RalfConvGeom)]
    )


SIGNATURES.update({
    "ralf_gemm_workspace_bytes": (sz, [ctypes.POINTER(RalfGemmDesc)]),
    "ralf_gemm": (i32, [ctypes.POINTER(RalfGemmDesc), vp, sz, vp]),
})
