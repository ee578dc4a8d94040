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
