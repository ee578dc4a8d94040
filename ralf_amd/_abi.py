"""ctypes signatures of every symbol declared in include/ralf_hip.h (kept in the same order).
tests/test_abi.py checks that the header, this table and the built library agree."""
import ctypes

i64, i32, sz, vp, f32 = ctypes.c_int64, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_float

SIGNATURES = {
    "ralf_last_error": (ctypes.c_char_p, []),
    "ralf_abi_version": (i32, []),
    # exact inner-product top-k
    "ralf_knn_topk_ip_workspace_bytes": (sz, [i64, i32, i32, i32]),
    "ralf_conv1x1_k64": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i64, i32, i32, vp]),
    "ralf_knn_topk_ip": (i32, [vp, i64, i32, vp, i32, i32, vp, vp, vp, sz, vp]),
    "ralf_knn_scores": (i32, [vp, i64, i32, vp, i32, vp, vp]),
    "ralf_knn_select": (i32, [vp, i64, i32, i32, vp, vp, vp, sz, vp]),
    "ralf_knn_rescore": (i32, [vp, i64, i32, vp, i32, vp, i32, vp, vp]),
    "ralf_knn_select_cand": (i32, [vp, vp, i32, i32, i32, vp, vp, vp, i64, vp, vp, i32, vp, vp]),
    "ralf_knn_rownorms": (i32, [vp, vp, i64, i32, vp, vp, vp]),
    "ralf_decode_token": (i32, [vp, vp]),
    "ralf_decode_token_limits": (i32, [vp, vp]),
    "ralf_knn_two_stage_workspace_bytes": (sz, [i64, i32, i32, i32]),
    "ralf_knn_topk_ip_two_stage": (i32, [vp, vp, i64, i32, vp, i32, i32, i32, vp, vp, vp, vp, vp, sz, vp]),
    "ralf_knn_topk_ip_two_stage_filtered": (i32, [vp, vp, i64, i32, vp, i32, i32, i32, vp, vp, vp, vp, vp, sz, vp]),
    "ralf_knn_list_unpack": (i32, [vp, vp, i32, i32, i32, vp, vp, vp, vp]),
    "ralf_knn_gather_rows": (i32, [vp, i32, vp, i32, i32, vp, vp]),
}


class RalfConvGeom(ctypes.Structure):
    _fields_ = [(n, i32) for n in ("RH", "RW", "SH", "SW", "SC", "KH", "KW", "stride", "pad", "mode")]


class RalfDecodeAttnDesc(ctypes.Structure):
    _fields_ = ([(n, vp) for n in ("x", "ln_g", "ln_b", "W", "bias", "kv", "kpm", "o")]
                + [(n, i64) for n in ("x_rs", "kv_bs", "kv_rs", "kpm_bs", "o_rs")]
                + [(n, i32) for n in ("B", "H", "d", "Sk", "self_")] + [("scale", f32), ("eps", f32), ("kv_hs", i64), ("kv_vo", i64), ("pos", vp)])


class RalfDecodeTokenLayer(ctypes.Structure):
    _fields_ = [(n, vp) for n in ("w_qkv", "b_qkv", "ln1_g", "ln1_b", "w_o1", "b_o1", "ln2_g", "ln2_b", "w_q2", "b_q2", "w_o2", "b_o2",
                                  "ln3_g", "ln3_b", "w_f1", "b_f1", "w_f2", "b_f2", "self_kv", "cross_kv")]


class RalfDecodeTokenDesc(ctypes.Structure):
    _fields_ = ([(n, vp) for n in ("tok", "pos_vec", "kpm", "emb", "pe", "lnh_g", "lnh_b", "w_head", "logits")] + [("kpm_bs", i64)]
                + [(n, i32) for n in ("B", "L", "M", "V", "nlayers", "pos")] + [("emb_scale", f32), ("eps", f32)] + [("layer", RalfDecodeTokenLayer * 8)]
                + [(n, vp) for n in ("s_allowed", "s_forced", "s_seed", "s_out", "s_seq_out", "s_flag_out")] + [(n, i64) for n in ("s_seq_ld", "s_flag_ld", "s_pad_id")]
                + [("s_call", ctypes.c_uint64)] + [(n, i32) for n in ("s_mode", "s_top_k", "s_row0")] + [("s_temperature", f32), ("s_top_p", f32)])


class RalfTLayerDesc(ctypes.Structure):
    _fields_ = ([(n, vp) for n in ("x", "ln1_g", "ln1_b", "w_in", "b_in", "w_o", "b_o", "kpm",
                                   "ln2_g", "ln2_b", "w_q", "b_q", "o2", "w_o2", "b_o2",
                                   "ln3_g", "ln3_b", "w1", "b1", "w2", "b2",
                                   "h1", "mean1", "rstd1", "qkv", "o1", "lse1", "x1",
                                   "h2", "mean2", "rstd2", "q", "x2",
                                   "h3", "mean3", "rstd3", "hid", "out", "seed")]
                + [(n, ctypes.c_uint64) for n in ("call_attn1", "call_out1", "call_out2", "call_ffn1", "call_ffn2")]
                + [("kpm_bs", i64)]
                + [(n, i32) for n in ("B", "S", "causal", "part")]
                + [(n, f32) for n in ("scale", "p_attn", "p_res", "eps")]
                + [("z", vp), ("act", i32), ("no_res", i32)])


class RalfTLayerBwdDesc(ctypes.Structure):
    _fields_ = ([(n, vp) for n in ("dy_m", "dy", "hid", "x2", "mean3", "rstd3", "ln3_g", "w2t", "w1t", "wot", "dz", "g", "g_m", "d_o", "dgamma", "dbeta", "seed")]
                + [("call_out", ctypes.c_uint64)] + [(n, i32) for n in ("B", "S", "stage", "nk")] + [("p", f32), ("pad2_", f32)])


class RalfPackJob(ctypes.Structure):
    _fields_ = [("src", vp), ("dst", vp), ("ld", i64), ("N", i32), ("K", i32), ("transpose", i32), ("pad_", i32)]


class RalfBnFoldJob(ctypes.Structure):
    _fields_ = [(n, vp) for n in ("gamma", "beta", "mean", "var", "scale", "shift")] + [("C", i32), ("pad_", i32)]


class RalfConvRelayoutJob(ctypes.Structure):
    _fields_ = [(n, vp) for n in ("w", "ohwi", "ikwo")] + [(n, i32) for n in ("Co", "Ci", "KK", "Cip", "dst_dtype", "first_block")]


class RalfPermuteJob(ctypes.Structure):
    _fields_ = ([("in_", vp), ("out", vp)] + [(n, i64) for n in ("s0", "s1", "s2", "s3")]
                + [(n, i32) for n in ("d0", "d1", "d2", "d3", "valid3", "src_dtype", "dst_dtype", "first_block")])


class RalfGemmDesc(ctypes.Structure):
    _fields_ = (
        [(n, vp) for n in ("A", "B", "C", "C2", "bias", "res", "aux")]
        + [(n, i64) for n in ("lda", "ldb", "ldc", "ldr", "sA0", "sA1", "sB0", "sB1", "sC0", "sC1", "sR0", "sR1")]
        + [(n, i32) for n in ("M", "N", "K", "nb0", "nb1", "dtype", "a_kcontig", "b_kcontig", "gather",
                              "act", "aux_mode", "out_f32", "accumulate", "splitk")]
        + [("alpha", f32), ("aux_scale", f32), ("g", RalfConvGeom)]
        + [("seed", vp), ("call_id", ctypes.c_uint64), ("drop_p", f32), ("atomic_out", i32), ("colstats", vp)]
        + [("sBias0", i64), ("sBk", i64), ("kseg", i32), ("colscale", vp)]
        + [("bnb_x", vp), ("bnb_mask", vp), ("bnb_mean", vp), ("bnb_part", vp)]
        + [("at_mode", i32), ("at_relu", i32), ("at_a2", vp), ("at_c1", vp), ("at_c2", vp), ("at_c3", vp), ("at_out", vp), ("at_mask", vp)]
        + [("flt_thresh", vp), ("flt_count", vp), ("flt_list", vp), ("flt_cap", i32), ("flt_thresh_ld", i32)]
        + [("ln_g", vp), ("ln_b", vp), ("ln_eps", f32), ("few_row_split", i32)]
    )


class RalfWgradJob(ctypes.Structure):
    _fields_ = [("dy", vp), ("x", vp), ("dw", vp), ("rows", i64), ("ld_dy", i64), ("ld_x", i64), ("ld_dw", i64),
                ("n_out", i32), ("n_in", i32), ("splitk", i32), ("pad", i32), ("db", vp)]


class RalfColsumJob(ctypes.Structure):
    _fields_ = [("x", vp), ("out", vp), ("ld", i64), ("rows", i32), ("cols", i32)]


SIGNATURES.update({
    "ralf_wgrad_grouped_workspace_bytes": (sz, [ctypes.POINTER(RalfWgradJob), i32]),
    "ralf_wgrad_grouped": (i32, [ctypes.POINTER(RalfWgradJob), i32, i32, vp, sz, vp]),
    "ralf_colsum_grouped": (i32, [ctypes.POINTER(RalfColsumJob), i32, i32, vp]),
    "ralf_permute4_batched": (i32, [vp, i32, i32, vp]),
    "ralf_conv_relayout_batched": (i32, [vp, i32, i32, vp]),
    "ralf_gemm_workspace_bytes": (sz, [ctypes.POINTER(RalfGemmDesc)]),
    "ralf_gemm": (i32, [ctypes.POINTER(RalfGemmDesc), vp, sz, vp]),
    "ralf_gemm_filter_tile": (i32, [ctypes.POINTER(RalfGemmDesc)]),
    "ralf_gemm_patch_variant": (i32, [ctypes.POINTER(RalfGemmDesc)]),
})

u64, u8p = ctypes.c_uint64, vp


class RalfAttnDesc(ctypes.Structure):
    _fields_ = (
        [(n, vp) for n in ("q", "k", "v", "o", "dout", "dq", "dk", "dv", "lse", "delta", "kpm", "seed")]
        + [(n, i64) for n in ("q_bs", "q_rs", "k_bs", "k_rs", "v_bs", "v_rs", "o_bs", "o_rs",
                              "do_bs", "do_rs", "dq_bs", "dq_rs", "dk_bs", "dk_rs", "dv_bs", "dv_rs")]
        + [("call_id", u64)]
        + [(n, i32) for n in ("B", "H", "Sq", "Sk", "dh", "dtype", "causal")]
        + [("scale", f32), ("p_drop", f32)]
        + [("kpm_bs", i64)]
    )


SIGNATURES.update({
    "ralf_layernorm_fwd": (i32, [i32, vp, vp, vp, vp, vp, vp, i32, i32, f32, vp]),
    "ralf_layernorm_bwd": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, f32, vp, u64, vp]),
    "ralf_colsum": (i32, [i32, vp, i64, vp, i32, i32, vp]),
    "ralf_bn_stats": (i32, [i32, vp, vp, vp, i64, i32, vp, vp]),
    "ralf_bn_finalize": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, f32, f32, i32, vp]),
    "ralf_bn_batch_stats": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, f32, f32, vp, vp]),
    "ralf_bn_stats_from_partials": (i32, [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, f32, f32, vp, vp]),
    "ralf_bn_fold_batched": (i32, [vp, i32, f32, vp]),
    "ralf_bn_apply": (i32, [i32, vp, vp, vp, vp, vp, vp, i64, i32, i32, vp]),
    "ralf_bn_bwd_reduce": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, vp, vp]),
    "ralf_bn_bwd_stats_from_partials": (i32, [vp, i32, vp, vp, vp, i32, vp, vp, vp, i64, vp, vp]),
    "ralf_bn_bwd_apply_affine": (i32, [i32, vp, vp, vp, vp, vp, vp, i64, i32, vp]),
    "ralf_bn_bwd_apply": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, vp]),
    "ralf_embed_fwd": (i32, [i32, vp, vp, vp, vp, i64, i32, i32, f32, vp]),
    "ralf_embed_bwd": (i32, [i32, vp, vp, vp, i64, i32, f32, vp]),
    "ralf_dropout": (i32, [i32, vp, vp, vp, i64, f32, vp, u64, vp]),
    "ralf_xent_fwd_bwd": (i32, [i32, vp, vp, vp, vp, i64, i32, i32, f32, vp]),
    "ralf_concat_rows": (i32, [i32, i32, i32, vp, vp, vp, vp, vp, i32, i32, vp]),
    "ralf_scale_pe_dropout": (i32, [i32, vp, vp, vp, i64, i32, i32, f32, f32, vp, u64, vp]),
    "ralf_add_scalar": (i32, [i32, vp, vp, vp, i64, i32, i64, i64, vp]),
    "ralf_sum_all": (i32, [i32, vp, vp, i64, i32, i64, vp]),
    "ralf_scale_dev": (i32, [i32, vp, vp, vp, i64, vp]),
    "ralf_counter_add": (i32, [vp, i32, i64, vp]),
    "ralf_layout_pack": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, i64, i32, vp]),
    "ralf_zero": (i32, [vp, i64, vp]),
    "ralf_copy2d": (i32, [i32, i32, vp, vp, i64, i32, i64, i64, i32, vp]),
    "ralf_permute4": (i32, [i32, i32, vp, vp, i32, i32, i32, i32, i64, i64, i64, i64, i32, vp]),
    "ralf_stem7x7_fwd": (i32, [vp, vp, vp, vp, i32, i32, i32, vp]),
    "ralf_stem7x7_wgrad_workspace_bytes": (sz, [i32, i32, i32]),
    "ralf_stem7x7_wgrad": (i32, [vp, vp, vp, i32, i32, i32, i32, vp, sz, vp]),
    "ralf_conv3x3_wgrad_workspace_bytes": (sz, [i32, i32, i32, i32, i32]),
    "ralf_conv3x3_wgrad": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, sz, vp]),
    "ralf_bn_relu_maxpool_fwd": (i32, [i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "ralf_bn_relu_maxpool_bwd_reduce": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "ralf_bn_relu_maxpool_bwd_apply": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "ralf_maxpool3x3s2_fwd": (i32, [i32, vp, vp, vp, i32, i32, i32, i32, vp]),
    "ralf_maxpool3x3s2_bwd": (i32, [i32, vp, vp, vp, i32, i32, i32, i32, vp]),
    "ralf_upsample_nearest_add": (i32, [i32, vp, vp, vp, i64, vp, i32, i32, i32, i32, i32, i32, vp]),
    "ralf_upsample_nearest_bwd": (i32, [i32, vp, i64, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "ralf_mask_sample": (i32, [vp, vp, vp, i32, i32, f32, vp, u64, vp, i32, i32, f32, vp]),
    "ralf_mask_sample_step": (i32, [vp, vp, vp, i32, i32, f32, vp, u64, vp, vp, i64, vp, i64, i64, i32, i32, f32, i32, vp]),
    "ralf_attention_fwd": (i32, [ctypes.POINTER(RalfAttnDesc), vp]),
    "ralf_attention_bwd": (i32, [ctypes.POINTER(RalfAttnDesc), vp]),
    "ralf_decode_attn": (i32, [ctypes.POINTER(RalfDecodeAttnDesc), vp]),
    "ralf_decode_attn_max_keys": (i32, []),
    "ralf_tlayer_fwd": (i32, [ctypes.POINTER(RalfTLayerDesc), vp]),
    "ralf_tlayer_pack": (i32, [ctypes.POINTER(RalfPackJob), i32, vp]),
    "ralf_tlayer_bwd": (i32, [ctypes.POINTER(RalfTLayerBwdDesc), vp]),
    "ralf_sumsq": (i32, [vp, i64, vp, vp]),
    "ralf_clip_coef": (i32, [vp, f32, vp, vp, vp]),
    "ralf_sumsq_partials": (i32, [vp, i64, vp, vp]),
    "ralf_clip_coef_partials": (i32, [vp, f32, vp, vp, vp]),
    "ralf_adamw": (i32, [vp, vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i32, vp, vp, vp, vp]),
    "ralf_stream_create": (i32, [ctypes.POINTER(vp)]),
    "ralf_stream_create_priority": (i32, [ctypes.POINTER(vp), i32]),
    "ralf_stream_destroy": (i32, [vp]),
})
