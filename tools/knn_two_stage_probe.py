"""component timing of the two-stage exact search at nq = 1024 (61548 x 1792 index)"""
import sys, time, torch
sys.path.insert(0, ".")
from ralf_amd import ops
from ralf_amd.retrieval.knn import knn_select
N, D, k, nq, pool = 61548, 1792, 16, 1024, 64
g = torch.Generator(device="cuda").manual_seed(0)
X = torch.randn(N, D, device="cuda", generator=g); X /= X.norm(dim=1, keepdim=True)
Xb = ops.cast(X, torch.bfloat16)
Q = torch.randn(nq, D, device="cuda", generator=g); Q /= Q.norm(dim=1, keepdim=True)
def t(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
qb = ops.cast(Q, torch.bfloat16)
print("cast q        %.3f ms" % t(lambda: ops.cast(Q, torch.bfloat16)))
print("coarse gemm   %.3f ms" % t(lambda: ops.gemm(qb, Xb, nq, N, D, out_dtype=torch.float32)))
coarse = ops.gemm(qb, Xb, nq, N, D, out_dtype=torch.float32)
print("select 65     %.3f ms" % t(lambda: knn_select(coarse, pool + 1)))
print("select 16     %.3f ms" % t(lambda: knn_select(coarse, 16)))
cval, cidx = knn_select(coarse, pool + 1)
print("sort          %.3f ms" % t(lambda: torch.sort(cidx[:, :pool], dim=1)))
cand, _ = torch.sort(cidx[:, :pool], dim=1)
from ralf_amd.retrieval.knn import knn_rescore
print("rescore       %.3f ms" % t(lambda: knn_rescore(X, Q, cand)))
exact = knn_rescore(X, Q, cand)
print("final select  %.3f ms" % t(lambda: knn_select(exact, k)))
print("norm          %.3f ms" % t(lambda: torch.linalg.vector_norm(Q, dim=1)))
