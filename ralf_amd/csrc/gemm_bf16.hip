// bf16 instantiations of the GEMM kernel (see gemm_impl.h); split from gemm.hip so the two element types build in parallel
#include <vector>

#include "gemm_impl.h"

int ralf_gemm_dispatch_bf16(void* kparams, int nbatch, hipStream_t st) { return dispatch<bf16>(*(KParams*)kparams, nbatch, st); }
int ralf_gemm_reduce_bf16(void* kparams, int nbatch, int blocks, hipStream_t st) {
    hipLaunchKernelGGL((splitk_reduce_kernel<bf16>), dim3(blocks), dim3(256), 0, st, *(KParams*)kparams, nbatch);
    return ralf::check_launch("gemm splitk reduce");
}

// ---- grouped weight gradients (ralf_wgrad_grouped) ----
// jobs: RalfWgradJob records; launched in chunks of GROUP_MAX jobs: one gemm_grouped_kernel (128x128 tiles, 8 waves) and, when
// some job splits its reduction, one gemm_grouped_reduce_kernel per chunk.
int ralf_gemm_grouped_bf16(const void* jobs_v, int njobs, void* workspace, size_t workspace_bytes, hipStream_t st) {
    const RalfWgradJob* jobs = (const RalfWgradJob*)jobs_v;
    // direct-to-LDS ring (gemm_impl.h GATHER 5, both operands row-contiguous): bit-identical, 81.9 -> 71.0 us per launch inside the step
    static const int glds = [] { const char* e = getenv("RALF_GEMM_GLDS_GROUPED"); return e ? atoi(e) : 1; }();   // 0: off (A/B runs, tests)
    float* ws = (float*)workspace;
    size_t ws_used = 0;
    for (int j0 = 0; j0 < njobs; j0 += GROUP_MAX) {
        const int n = std::min(GROUP_MAX, njobs - j0);
        GParams G;
        GRedParams R;
        G.njobs = n;
        R.njobs = 0;
        int first = 0, rfirst = 0;
        for (int i = 0; i < n; ++i) {
            const RalfWgradJob& w = jobs[j0 + i];
            GJob& J = G.j[i];
            J.A = w.dy; J.B = w.x; J.C = w.dw; J.partial = nullptr;
            J.db = glds ? w.db : nullptr; J.db_partial = nullptr;   // (the register-staged form has no column sums: ralf_colsum_grouped below)
            J.M = w.n_out; J.N = w.n_in; J.K = (int)w.rows; J.lda = (int)w.ld_dy; J.ldb = (int)w.ld_x; J.ldc = (int)w.ld_dw;
            const int ktiles = J.K / 64;
            int sk = w.splitk < 1 ? 1 : (w.splitk > ktiles ? ktiles : w.splitk);
            J.kchunk = ceil_div(ktiles, sk) * 64;
            J.splitk = sk = ceil_div(J.K, J.kchunk);
            J.tiles_n = ceil_div(J.N, 128);
            J.nwg = ceil_div(J.M, 128) * J.tiles_n;
            J.first = first;
            J.pad = 0;
            first += J.nwg * sk;
            if (sk > 1) {
                const size_t need = (size_t)sk * J.M * J.N;
                if ((ws_used + need) * sizeof(float) > workspace_bytes) {
                    ralf::set_error("wgrad_grouped: split-K workspace too small (%zu bytes)", workspace_bytes);
                    return RALF_ERR_WORKSPACE;
                }
                J.partial = ws + ws_used;
                ws_used += need;
                GRed& r = R.j[R.njobs++];
                r.partial = J.partial; r.C = J.C; r.per = (int64_t)J.M * J.N; r.ldc = J.ldc; r.N = J.N; r.splitk = sk; r.first = rfirst;
                rfirst += ceil_div(r.per, 2048);
                if (J.db) {   // the bias slabs [sk][M]: one more record of the same reduce kernel (a [1, M] matrix)
                    if ((ws_used + (size_t)sk * J.M) * sizeof(float) > workspace_bytes) {
                        ralf::set_error("wgrad_grouped: split-K workspace too small (%zu bytes)", workspace_bytes);
                        return RALF_ERR_WORKSPACE;
                    }
                    J.db_partial = ws + ws_used;
                    ws_used += (size_t)sk * J.M;
                    GRed& rb = R.j[R.njobs++];
                    rb.partial = J.db_partial; rb.C = J.db; rb.per = J.M; rb.ldc = J.M; rb.N = J.M; rb.splitk = sk; rb.first = rfirst;
                    rfirst += ceil_div(rb.per, 2048);
                }
            }
        }
        static const int nw4 = [] { const char* e = getenv("RALF_WGRAD_NW4"); return e ? atoi(e) : 0; }();   // A/B: 4 waves of 64 x 64 per 128 x 128 tile
        if (glds && nw4) hipLaunchKernelGGL((gemm_grouped_kernel<bf16, 2, 2, 4, 5>), dim3(first), dim3(256), 0, st, G);
        else if (glds) hipLaunchKernelGGL((gemm_grouped_kernel<bf16, 2, 2, 8, 5>), dim3(first), dim3(512), 0, st, G);
        else hipLaunchKernelGGL((gemm_grouped_kernel<bf16, 2, 2, 8>), dim3(first), dim3(512), 0, st, G);
        if (R.njobs) hipLaunchKernelGGL(gemm_grouped_reduce_kernel, dim3(rfirst), dim3(256), 0, st, R);
    }
    if (!glds) {   // bias gradients of the register-staged form: the separate column-sum launch
        std::vector<RalfColsumJob> cj;
        for (int i = 0; i < njobs; ++i)
            if (jobs[i].db) cj.push_back(RalfColsumJob{jobs[i].dy, jobs[i].db, jobs[i].ld_dy, (int)jobs[i].rows, jobs[i].n_out});
        for (size_t i = 0; i < cj.size(); ++i)   // (its own constraint: cols % 256 == 0 -- checked there)
            if (int rc = ralf_colsum_grouped(&cj[i], 1, RALF_BF16, st)) return rc;
    }
    return ralf::check_launch("wgrad_grouped");
}
