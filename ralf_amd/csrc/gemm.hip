// MFMA GEMM for gfx950 with fused epilogues; the one contraction kernel behind every Linear,
// attention projection, FFN, vocab head and (through the implicit-im2col gather) every Conv2d
// of the RALF train step.
//
//   C[z][m][n] = epi( alpha * sum_k A[z][m][k] * B[z][k][n] )
//
// Replaces torch.nn.functional.linear / conv2d and their autograd backward as called from
//   nn.TransformerEncoderLayer / DecoderLayer / MultiheadAttention
//       (image2layout/train/models/retrieval_augmented_autoreg.py:116-126, common/common.py:25-34)
//   FeedForward / Attention (common/attention.py:15-71), BaseDecoder.head (common/common.py:38-40)
//   ResnetBackbone convolutions (common/image.py:80-111)
//
// Design:
//   * (64*FM)x(64*FN) output tile per 256-thread workgroup (FM,FN in {1,2}: 128x128 for the big products,
//     64-wide variants for N = 64 convolutions and for small grids that need more workgroups in flight),
//     2x2 waves, each wave FMxFN fragments of 32x32 (v_mfma_f32_32x32x16_bf16 or the exact-fp32
//     v_mfma_f32_32x32x2_f32); fp32 accumulate always.  The matrix core computes the TRANSPOSED tile
//     (weights as the row operand) so every lane ends up with 4 consecutive output columns: 8/16-byte
//     epilogue loads and stores instead of 2-byte ones.
//   * each operand is either "k-contiguous" (row-major [rows][K]) or "row-contiguous" ([K][rows]);
//     tiles are staged global -> registers -> LDS in their MEMORY order (coalesced 16-B loads) and
//     the row-contiguous case is fed to the matrix core with ds_read_b64_tr_b16 (bf16) / plain
//     ds_read_b32 (fp32), so NN / NT / TN products need no transposed copies in HBM.
//   * the "pixel x (kh,kw,c)" operand of a convolution is gathered on the fly from NHWC (implicit
//     im2col; forward / data-gradient / weight-gradient all use the same gather).
//   * split-K for the weight-gradient shapes (tiny output, huge reduction), deterministic:
//     partial slabs + a reduce kernel that also runs the epilogue.
//   * epilogue: alpha, bias, ReLU/GELU(erf), activation-gradient masks, residual add, second
//     (pre-activation) output, fp32 or bf16 stores.
#include "gemm_impl.h"

extern "C" size_t ralf_gemm_workspace_bytes(const RalfGemmDesc* d) {
    if (!d || d->splitk <= 1 || d->atomic_out) return 0;
    const int nb = d->nb0 * (d->nb1 > 0 ? d->nb1 : 1);
    return (size_t)d->splitk * nb * d->M * d->N * sizeof(float);
}

extern "C" int ralf_gemm_filter_tile(const RalfGemmDesc* dp) {
    if (!dp || dp->M <= 0 || dp->N <= 0 || dp->K <= 0) { ralf::set_error("gemm_filter_tile: bad descriptor"); return RALF_ERR_INVALID; }
    RalfGemmDesc d = *dp;
    d.splitk = 1;
    if (!gemm_use128(d, 1)) return 64;
    return gemm_env_glds() && !gemm_env_tile() && gemm_tile256_shape(d, 1) ? 256 : 128;   // (launch_cfg's rule for a bf16 NT product on the aligned path)
}

// the gathered tensor can be addressed with 32-bit element offsets (and 24-bit row multiplies): the lean loaders of gemm_body
static bool gather_lean_ok(const RalfGemmDesc& d) {
    const RalfConvGeom& g = d.g;
    if (!d.gather || g.RH <= 0 || g.RW <= 0 || g.SH <= 0 || g.SW <= 0) return false;
    const int64_t rows = d.gather == 1 ? d.M : d.K, hw = (int64_t)g.RH * g.RW;
    const int64_t img = (int64_t)g.SH * g.SW * g.SC, src = ceil_div(rows, hw) * img + img;
    const int tap = g.mode ? g.stride : 1;
    bool ok = src < (1ll << 30) && (int64_t)g.SW * g.SC < (1 << 22) && g.SH < (1 << 20) && g.RH < (1 << 20) && g.KH < 1024 && g.KW < 1024 && g.stride < 1024;
    if (d.gather == 1 && (tap & (tap - 1)) != 0) ok = false;
    return ok;
}

// 3 x 3 / stride-1 / pad-1 convolution (forward or data gradient) on whole image rows: the patch form (gemm_impl.h GATHER 15).  The tile it takes:
// 0 = none (the tap gather), 1 = 128 x 128, 2 = 256 x 128, 3 = 256 x 64 -- the taller tile where its halo patch fits the variant's LDS and enough tiles
// exist to fill the chip (fewer: the 64 x 64 tap gather spreads wider).
static int patch_variant(const RalfGemmDesc& d, int nbatch) {
    static const int patch_on = [] { const char* e = getenv("RALF_GEMM_PATCH"); return e ? atoi(e) : 1; }();             // 0 = off (A/B runs, tests)
    static const int min_tiles = [] { const char* e = getenv("RALF_GEMM_PATCH_TILES"); return e ? atoi(e) : 192; }();
    const RalfConvGeom& g = d.g;
    auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
    if (!(patch_on && d.gather == 1 && gather_lean_ok(d) && g.stride == 1 && g.KH == 3 && g.KW == 3 && g.pad == 1 && g.RH == g.SH && g.RW == g.SW && d.dtype == RALF_BF16 &&
          d.a_kcontig && d.b_kcontig && d.splitk <= 1 && nbatch == 1 && pow2(g.SW) && g.SW <= 128 && pow2(g.SC) && g.SC >= 64 && d.K == 9 * g.SC && !d.kseg)) return 0;
    auto fits = [&](int bm, int fn) {   // whole image rows per tile, tiles inside one image, the halo patch inside the variant's LDS
        return bm % g.SW == 0 && (g.SH * g.SW) % bm == 0 && d.M % bm == 0 && d.N % (64 * fn) == 0 &&
               (int64_t)(bm / g.SW + 2) * (g.SW + 2) * (2 * g.SC + 16) <= gemm_patch_bytes(fn) && (int64_t)(d.M / bm) * (d.N / (64 * fn)) >= min_tiles;
    };
    return fits(256, 2) ? 2 : fits(128, 2) ? 1 : fits(256, 1) ? 3 : 0;
}

extern "C" int ralf_gemm_patch_variant(const RalfGemmDesc* dp) {
    if (!dp || dp->M <= 0 || dp->N <= 0 || dp->K <= 0) { ralf::set_error("gemm_patch_variant: bad descriptor"); return RALF_ERR_INVALID; }
    return patch_variant(*dp, (dp->nb0 > 0 ? dp->nb0 : 1) * (dp->nb1 > 0 ? dp->nb1 : 1));
}

extern "C" int ralf_gemm(const RalfGemmDesc* dp, void* workspace, size_t workspace_bytes, void* stream) {
    RALF_REQUIRE(dp, "gemm: null descriptor");
    KParams P;
    P.d = *dp;
    RalfGemmDesc& d = P.d;
    RALF_REQUIRE(d.A && d.B && d.C, "gemm: null operand");
    RALF_REQUIRE(d.M > 0 && d.N > 0 && d.K > 0, "gemm: empty problem M=%d N=%d K=%d", d.M, d.N, d.K);
    RALF_REQUIRE(d.dtype == RALF_F32 || d.dtype == RALF_BF16, "gemm: dtype %d", d.dtype);
    if (d.nb0 <= 0) d.nb0 = 1;
    if (d.nb1 <= 0) d.nb1 = 1;
    if (d.splitk <= 0) d.splitk = 1;
    const int nbatch = d.nb0 * d.nb1;
    const int BK = d.dtype == RALF_F32 ? 32 : 64, VEC = d.dtype == RALF_F32 ? 4 : 8;
    if (d.gather) RALF_REQUIRE(d.g.SC % VEC == 0 && d.g.KH > 0 && d.g.KW > 0 && d.g.stride > 0, "gemm: gather needs channels %% %d == 0", VEC);
    if (d.gather) {
        RALF_REQUIRE((int64_t)(d.gather == 1 ? d.M : d.K) < (1ll << 31) && (int64_t)(d.gather == 1 ? d.K : d.N) < (1ll << 31), "gemm: gather index range");
        P.fd_hw.set((uint32_t)(d.g.RH * d.g.RW)); P.fd_rw.set((uint32_t)d.g.RW); P.fd_sc.set((uint32_t)d.g.SC);
        P.fd_kw.set((uint32_t)d.g.KW); P.fd_st.set((uint32_t)d.g.stride); P.fd_tap.set(d.g.mode ? (uint32_t)d.g.stride : 1u);
    } else {
        P.fd_hw.set(1); P.fd_rw.set(1); P.fd_sc.set(1); P.fd_kw.set(1); P.fd_st.set(1); P.fd_tap.set(1);
    }
    P.ln_sgn = 1; P.ln_sh = 0; P.ln_pm = 0; P.img = P.swsc = 0; P.inc_b = P.inc_y = P.inc_x = 0;
    bool lean_ok = false;   // the gathered tensor can be addressed with 32-bit element offsets (and 24-bit row multiplies)
    if (d.gather) {
        const RalfConvGeom& g = d.g;
        RALF_REQUIRE(g.RH > 0 && g.RW > 0 && g.SH > 0 && g.SW > 0 && g.pad >= 0 && (g.mode == 0 || g.mode == 1), "gemm: gather geometry");
        const int64_t rows = d.gather == 1 ? d.M : d.K, hw = (int64_t)g.RH * g.RW;
        const int64_t img = (int64_t)g.SH * g.SW * g.SC, src = ceil_div(rows, hw) * img + img;
        const int tap = g.mode ? g.stride : 1;
        lean_ok = src < (1ll << 30) && (int64_t)g.SW * g.SC < (1 << 22) && g.SH < (1 << 20) && g.RH < (1 << 20) && g.KH < 1024 && g.KW < 1024 && g.stride < 1024;
        if (lean_ok) { P.img = (int)img; P.swsc = g.SW * g.SC; }
        if (d.gather == 1 && (tap & (tap - 1)) == 0) {
            P.ln_sgn = g.mode ? -1 : 1; P.ln_pm = tap - 1;
            while ((1 << P.ln_sh) < tap) ++P.ln_sh;
        } else if (d.gather == 1) {
            lean_ok = false;
        }
        if (d.gather == 2) {
            RALF_REQUIRE(g.mode == 0, "gemm: gather=2 (weight gradient) takes the forward geometry (mode 0)");
            RALF_REQUIRE(lean_ok, "gemm: gather=2 source too large for 32-bit offsets (%lld elements)", (long long)src);
            RALF_REQUIRE((BK / hw) * img < (1ll << 30), "gemm: gather=2 image step");
            P.inc_b = (int)((BK / hw) * img); P.inc_y = (int)((BK % hw) / g.RW); P.inc_x = (int)((BK % hw) % g.RW);
        }
    }
    // data gradient of a stride-2 convolution: the parity-class form (gemm_impl.h GATHER 14) -- even output grid, whole 128-row tiles per class, <= 9 taps
    P.par = 0;
    P.fd_q.set(1); P.fd_hw2.set(1); P.fd_rw2.set(1);
    {
        static const int par_on = [] { const char* e = getenv("RALF_GEMM_PARITY"); return e ? atoi(e) : 1; }();   // 0 = off (A/B runs, tests)
        const RalfConvGeom& g = d.g;
        if (par_on && d.gather == 1 && lean_ok && g.mode == 1 && g.stride == 2 && d.dtype == RALF_BF16 && d.splitk == 1 && nbatch == 1 && g.RH % 2 == 0 && g.RW % 2 == 0 &&
            g.KH * g.KW <= 9 && d.M % 4 == 0 && (d.M / 4) % 128 == 0 && (int64_t)d.M % ((int64_t)g.RH * g.RW) == 0) {
            P.par = 1;
            P.fd_q.set((uint32_t)(d.M / 4)); P.fd_hw2.set((uint32_t)((g.RH / 2) * (g.RW / 2))); P.fd_rw2.set((uint32_t)(g.RW / 2));
        }
    }
    P.patch = lean_ok ? patch_variant(d, nbatch) : 0;   // (3 x 3 / stride 1: the tile's input patch resident in LDS, gemm_impl.h GATHER 15)
    P.p_pw = P.p_str = P.p_swsh = P.p_c8sh = 0;
    P.fd_pw.set(1);
    if (P.patch) {
        const RalfConvGeom& g = d.g;
        P.p_pw = g.SW + 2; P.p_str = 2 * g.SC + 16;
        while ((1 << P.p_swsh) < g.SW) ++P.p_swsh;
        while ((8 << P.p_c8sh) < g.SC) ++P.p_c8sh;
        P.fd_pw.set((uint32_t)P.p_pw);
    }
    const int ktiles = ceil_div(d.K, BK);
    if (d.splitk > ktiles) d.splitk = ktiles;
    P.kchunk = ceil_div(ktiles, d.splitk) * BK;
    d.splitk = ceil_div(d.K, P.kchunk);
    if (d.atomic_out) {
        RALF_REQUIRE((d.dtype == RALF_F32 || d.out_f32) && !d.bias && !d.act && !d.aux && !d.res && !d.C2 && d.drop_p == 0.f,
                     "gemm: atomic_out needs a plain fp32 output (it adds alpha*A@B into C)");
    }
    RALF_REQUIRE(d.drop_p >= 0.f && d.drop_p < 1.f && (d.drop_p == 0.f || (d.seed && nbatch == 1 && d.ldc == d.N)),
                 "gemm: fused dropout needs 0 <= p < 1, a seed, a single batch and a contiguous [M,N] output");
    {   // interior fast path of the operand loaders
        const int es = d.dtype == RALF_F32 ? 4 : 2;
        const bool al = d.lda % VEC == 0 && d.ldb % VEC == 0 && ((uintptr_t)d.A % 16) == 0 && ((uintptr_t)d.B % 16) == 0 &&
                        (d.sA0 * es) % 16 == 0 && (d.sA1 * es) % 16 == 0 && (d.sB0 * es) % 16 == 0 && (d.sB1 * es) % 16 == 0;
        const bool rows_ok = (d.a_kcontig || (d.M % VEC == 0 && d.M >= VEC)) && (d.b_kcontig || (d.N % VEC == 0 && d.N >= VEC));
        P.fast = (!d.gather && al && rows_ok && d.K % BK == 0) ? 1 : 0;
        P.tapuni = (d.gather == 1 && lean_ok && d.g.SC % BK == 0 && P.kchunk % BK == 0 && d.K % BK == 0) ? 1 : 0;
        // convolution GEMMs load whole aligned vectors without an element-wise fallback (load_vec_al): weights [N][K] with
        // K % VEC == 0 for the forward / data gradient, dy [K][M] with M % VEC == 0 for the weight gradient
        if (d.gather == 1) RALF_REQUIRE(al && d.K % VEC == 0, "gemm: gather=1 needs 16-byte aligned operands and K %% %d == 0", VEC);
        if (d.gather == 2) RALF_REQUIRE(al && d.M % VEC == 0 && d.M >= VEC, "gemm: gather=2 needs 16-byte aligned operands and M %% %d == 0", VEC);
    }
    if (d.kseg) {
        RALF_REQUIRE(P.fast && d.dtype == RALF_BF16 && d.splitk == 1 && d.kseg % BK == 0 && d.K % d.kseg == 0 && (d.sBk * 2) % 16 == 0,
                     "gemm: a segmented B (kseg=%d) needs the bf16 interior path, no split-K, kseg %% %d == 0 and K %% kseg == 0", d.kseg, BK);
    }
    RALF_REQUIRE(d.sBias0 == 0 || (d.bias && d.sBias0 % 4 == 0), "gemm: sBias0 needs a bias and a multiple of 4");
    P.partial = nullptr;
    {   // 4-wide epilogue accesses need 4-element-aligned leading dims / batch strides and 16-byte aligned bases
        const int es = (d.dtype == RALF_F32 || d.out_f32) ? 4 : 2;   // element size of C / C2
        const int et = d.dtype == RALF_F32 ? 4 : 2;                   // element size of res / aux
        auto al = [](const void* p, int bytes) { return (((uintptr_t)p) % (uintptr_t)(4 * bytes)) == 0; };
        bool ok = d.ldc % 4 == 0 && d.sC0 % 4 == 0 && d.sC1 % 4 == 0 && al(d.C, es) && (!d.C2 || al(d.C2, es));
        ok = ok && (!d.bias || al(d.bias, 4));
        ok = ok && (!d.aux || al(d.aux, et));
        ok = ok && (!d.res || (d.ldr % 4 == 0 && d.sR0 % 4 == 0 && d.sR1 % 4 == 0 && al(d.res, et)));
        auto al16 = [](const void* p) { return (((uintptr_t)p) % 16) == 0; };
        bool ok8 = ok && d.ldc % 8 == 0 && d.sC0 % 8 == 0 && d.sC1 % 8 == 0 && al16(d.C) && (!d.C2 || al16(d.C2)) && (!d.bias || al16(d.bias)) &&
                   (!d.aux || al16(d.aux)) && (!d.res || (d.ldr % 8 == 0 && d.sR0 % 8 == 0 && d.sR1 % 8 == 0 && al16(d.res)));
        if (d.splitk > 1 && !d.atomic_out) ok8 = ok && d.N % 8 == 0;   // slab output: [M][N] fp32 in the workspace
        P.vec_epi = ok8 ? 2 : ok ? 1 : 0;   // 2: 8-wide (LDS-staged) epilogue, 1: 4-wide direct, 0: scalar
    }
    if (d.colstats) {
        RALF_REQUIRE(P.vec_epi == 2 && d.N % 64 == 0 && d.splitk == 1 && nbatch == 1 && d.alpha == 1.f && !d.bias && !d.act && !d.res && !d.aux &&
                     !d.C2 && !d.accumulate && !d.atomic_out && d.drop_p == 0.f && (d.dtype == RALF_F32 || !d.out_f32),
                     "gemm: colstats needs a plain epilogue, one batch, no split-K, N %% 64 == 0 and 8-wide aligned output");
    }
    if (d.bnb_part || d.bnb_x || d.bnb_mean || d.bnb_mask) {
        const bool same = d.dtype == RALF_F32 || !d.out_f32;
        RALF_REQUIRE(d.bnb_part && d.bnb_x && d.bnb_mean && P.vec_epi == 2 && d.N % 64 == 0 && d.splitk == 1 && nbatch == 1 && same && d.ldc == d.N &&
                     (!d.res || d.ldr % 8 == 0) && !d.act && !d.aux && !d.C2 && !d.accumulate && !d.atomic_out && !d.colstats && !d.colscale && d.drop_p == 0.f &&
                     ((uintptr_t)d.bnb_x % 16) == 0 && ((uintptr_t)d.bnb_mean % 16) == 0 && ((uintptr_t)d.bnb_part % 4) == 0,
                     "gemm: bnb_* needs x, mean and the partial buffer, a contiguous aligned [M,N] output in the operand dtype, N %% 64 == 0, no split-K and an alpha / bias / res epilogue");
    }
    if (d.splitk > 1 && !d.atomic_out) {
        const size_t need = (size_t)d.splitk * nbatch * d.M * d.N * sizeof(float);
        if (!workspace || workspace_bytes < need) {
            ralf::set_error("gemm: split-K workspace %zu < required %zu bytes", workspace_bytes, need);
            return RALF_ERR_WORKSPACE;
        }
        P.partial = (float*)workspace;
    }
    if (d.flt_list) {
        RALF_REQUIRE(d.flt_thresh && d.flt_count && d.flt_cap > 0 && d.dtype == RALF_BF16 && d.a_kcontig && d.b_kcontig && P.fast && nbatch == 1 && d.splitk == 1 &&
                     !d.bias && !d.act && !d.res && !d.aux && !d.C2 && !d.colstats && !d.bnb_part && !d.colscale && !d.accumulate && !d.atomic_out && d.drop_p == 0.f && !d.kseg &&
                     ((uintptr_t)d.flt_list % 8) == 0,
                     "gemm: flt_* needs thresholds, counters and a capacity, bf16 NT operands on the aligned path, one batch, no split-K and a plain epilogue");
    }
    if (d.ln_g) {
        RALF_REQUIRE(d.ln_b && d.dtype == RALF_BF16 && !d.gather && d.a_kcontig && d.b_kcontig && P.fast && d.K == 256 && d.M <= 512 && nbatch == 1 && d.splitk == 1 &&
                     !d.colstats && !d.bnb_part && !d.kseg && !d.atomic_out && !d.flt_list && !d.at_mode && ((uintptr_t)d.ln_g % 16) == 0 && ((uintptr_t)d.ln_b % 16) == 0,
                     "gemm: ln_* (LayerNorm in front of a few-row product) needs bf16 NT operands on the aligned path, K == 256, M <= 512, one batch, no split-K");
    }
    hipStream_t st = (hipStream_t)stream;
    if (d.at_mode) {
        RALF_REQUIRE(d.at_mode == 1 || d.at_mode == 2, "gemm: at_mode %d", d.at_mode);
        RALF_REQUIRE(d.dtype == RALF_BF16 && !d.gather && d.a_kcontig && P.fast && d.lda == d.K && d.K <= 512 && nbatch == 1 && d.splitk == 1 && !d.kseg &&
                     (int64_t)d.M * d.K < (1ll << 31),
                     "gemm: the operand transform needs bf16, a plain k-contiguous A with lda == K on the aligned path, K %% 64 == 0, K <= 512, one batch, no split-K");
        RALF_REQUIRE(d.at_c1 && d.at_c2 && (d.at_mode == 1 || (d.at_c3 && d.at_a2)), "gemm: at_mode %d: missing coefficients / second operand", d.at_mode);
        RALF_REQUIRE((((uintptr_t)d.at_a2 | (uintptr_t)d.at_out) & 15) == 0, "gemm: at_a2 / at_out must be 16-byte aligned");
        RALF_REQUIRE(!d.C2 && !d.aux && !d.act && d.drop_p == 0.f && !d.atomic_out && !d.colscale, "gemm: the operand transform goes with a plain, column-statistics or bnb_* epilogue");
        return ralf_gemm_dispatch_at(&P, nbatch, st);
    }
    int rc = d.dtype == RALF_F32 ? ralf_gemm_dispatch_f32(&P, nbatch, st) : ralf_gemm_dispatch_bf16(&P, nbatch, st);
    if (rc || d.splitk <= 1 || d.atomic_out) return rc;
    const int64_t total = (int64_t)d.M * d.N * nbatch;
    const int blocks = (int)std::min<int64_t>((total + 255) / 256, 2048);
    return d.dtype == RALF_F32 ? ralf_gemm_reduce_f32(&P, nbatch, blocks, st) : ralf_gemm_reduce_bf16(&P, nbatch, blocks, st);
}

extern "C" size_t ralf_wgrad_grouped_workspace_bytes(const RalfWgradJob* jobs, int njobs) {
    size_t n = 0;
    for (int i = 0; jobs && i < njobs; ++i)
        if (jobs[i].splitk > 1) n += (size_t)jobs[i].splitk * ((size_t)jobs[i].n_out * jobs[i].n_in + (jobs[i].db ? jobs[i].n_out : 0)) * sizeof(float);
    return n;
}

extern "C" int ralf_wgrad_grouped(const RalfWgradJob* jobs, int njobs, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
    RALF_REQUIRE(jobs && njobs > 0, "wgrad_grouped: no jobs");
    RALF_REQUIRE(dtype == RALF_BF16, "wgrad_grouped: bf16 operands only (the fp32 parity mode uses ralf_gemm per layer)");
    for (int i = 0; i < njobs; ++i) {
        const RalfWgradJob& w = jobs[i];
        RALF_REQUIRE(w.dy && w.x && w.dw && w.rows > 0 && w.rows % 64 == 0 && w.rows < (1ll << 31), "wgrad_grouped: job %d: rows %lld must be a positive multiple of 64", i, (long long)w.rows);
        RALF_REQUIRE(w.n_out >= 8 && w.n_in >= 8 && w.n_out % 8 == 0 && w.n_in % 8 == 0 && w.ld_dy % 8 == 0 && w.ld_x % 8 == 0 && w.ld_dw % 8 == 0,
                     "wgrad_grouped: job %d: dimensions and leading dimensions must be multiples of 8", i);
        RALF_REQUIRE(((uintptr_t)w.dy % 16) == 0 && ((uintptr_t)w.x % 16) == 0 && ((uintptr_t)w.dw % 16) == 0, "wgrad_grouped: job %d: operands must be 16-byte aligned", i);
        RALF_REQUIRE(!w.db || (w.n_out % 256 == 0 && ((uintptr_t)w.db % 16) == 0), "wgrad_grouped: job %d: a bias gradient needs n_out %% 256 == 0 and a 16-byte aligned buffer", i);
    }
    return ralf_gemm_grouped_bf16(jobs, njobs, workspace, workspace_bytes, (hipStream_t)stream);
}
