"""row-strip GEMM (ralf_rs_gemm) vs the tiled GEMM (ralf_gemm) on the transformer's shapes: correctness against the tiled kernel and
time per launch (20 launches captured in one hipGraph, so the host is out of the loop)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ralf_amd import ops

dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)


def graph_time(fn, n=20, reps=5):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(n):
            fn()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        gr.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n * reps) * 1e3


def bf(*shape, scale=1.0):
    return (torch.randn(*shape, device=dev, generator=g) * scale).bfloat16()


print(f"{'shape (M,N,K) mode':44s} {'rs us':>8s} {'tiled us':>9s} {'rs TF/s':>8s}  max rel err")
for M in (16384, 3200, 256, 33792):
    for (N, K, kc, extra) in ((768, 256, True, "ln"), (1024, 256, True, "relu+drop"), (256, 256, True, "res+drop"), (256, 1024, True, "res"),
                              (512, 256, True, "plain"), (256, 768, False, "dgrad"), (256, 1024, False, "dgrad"), (1024, 256, False, "dgrad+mask"), (256, 256, False, "dgrad+res")):
        x = bf(M, K)
        w = bf(N, K, scale=K ** -0.5) if kc else bf(K, N, scale=K ** -0.5)
        bias = torch.randn(N, device=dev, generator=g) if kc else None
        res = bf(M, N) if "res" in extra else None
        aux = bf(M, N) if "mask" in extra else None
        seed = torch.zeros(1, dtype=torch.int64, device=dev)
        p = 0.1 if "drop" in extra else 0.0
        act = "relu" if "relu" in extra else None
        ln = (torch.rand(K, device=dev, generator=g) + 0.5, torch.randn(K, device=dev, generator=g) * 0.1) if extra == "ln" else None
        out_rs = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        out_t = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        xln = torch.empty(M, K, dtype=torch.bfloat16, device=dev) if ln else None
        st = (torch.empty(M, device=dev), torch.empty(M, device=dev)) if ln else None

        def run_rs():
            ops.rs_gemm(x, w, M, N, K, w_kcontig=kc, out=out_rs, bias=bias, act=act, res=res, aux=aux, drop_p=p, seed=seed, call_id=3, ln=ln, xln=xln, ln_stats=st)

        def run_tiled():
            xi = x
            if ln:
                xi, _, _ = ops.layernorm_fwd(x, ln[0], ln[1])
            ops.gemm(xi, w, M, N, K, b_kcontig=kc, out=out_t, bias=bias, act=act, res=res, aux=aux, aux_mode="relu_mask" if aux is not None else None,
                     drop_p=p, seed=seed if p else None, call_id=3)
        run_rs(); run_tiled()
        torch.cuda.synchronize()
        err = ((out_rs.float() - out_t.float()).abs().max() / out_t.float().abs().max()).item()
        if ln:
            xr, _, _ = ops.layernorm_fwd(x, ln[0], ln[1])
            err = max(err, ((xln.float() - xr.float()).abs().max() / xr.float().abs().max()).item())
        t_rs, t_t = graph_time(run_rs), graph_time(run_tiled)
        print(f"({M:5d},{N:4d},{K:4d}) {'NT' if kc else 'NN'} {extra:12s}              {t_rs:8.1f} {t_t:9.1f} {2.0 * M * N * K / t_rs / 1e6:8.0f}  {err:.2e}", flush=True)
