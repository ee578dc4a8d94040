"""A/B of the exact-scan variants in ONE process per variant (RALF_KNN_GLDS is read once): python tools/knn_lab.py
prints, per variant and query count, the scan time (HIP events, median of rounds) and checks bit-equality with the register-staged scan."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, torch
sys.path.insert(0, %r)
from ralf_amd.retrieval.knn import knn_scores, knn_topk_ip
from ralf_amd import _lib
torch.manual_seed(0)
out = {}
for (N, D) in ((61548, 1792), (61548, 256), (48544, 512), (7734, 100)):
    g = torch.Generator(device="cuda").manual_seed(N + D)
    X = torch.randn(N, D, device="cuda", generator=g); X /= X.norm(dim=1, keepdim=True)
    for nq in (1, 16, 17, 32):
        Q = torch.randn(nq, D, device="cuda", generator=g); Q /= Q.norm(dim=1, keepdim=True)
        S = knn_scores(X, Q)
        ws = torch.empty(_lib.lib().ralf_knn_topk_ip_workspace_bytes(N, D, nq, 16), dtype=torch.uint8, device="cuda")
        ts, tw = [], []
        for r in range(7):
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record()
            for _ in range(10): knn_scores(X, Q)
            e1.record()
            for _ in range(10): knn_topk_ip(X, Q, 16, ws)
            e2.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 100); tw.append(e1.elapsed_time(e2) * 100)
        out[(N, D, nq)] = (sorted(ts)[3], sorted(tw)[3], S.cpu())
torch.save(out, sys.argv[1])
""" % ROOT

res = {}
for v in ("0", "3", "2", "4", "13", "14"):
    path = f"/tmp/knn_lab_{v}.pt"
    r = subprocess.run([sys.executable, "-c", CHILD, path], env=dict(os.environ, RALF_KNN_GLDS=v), capture_output=True, text=True)
    if r.returncode != 0:
        print("variant", v, "FAILED", r.stderr[-1500:])
        continue
    import torch
    res[v] = torch.load(path)
base = res["0"]
print("variant (RALF_KNN_GLDS): scan us / whole-call us; * = scores differ from the register-staged scan")
for key in base:
    N, D, nq = key
    by = N * D * 4 + nq * D * 4 + nq * 16 * 12
    row = f"{N}x{D} nq={nq:2d} |"
    for v in res:
        ts, tw, S = res[v][key]
        same = torch.equal(S, base[key][2])
        row += f"  [{v}] {ts:6.1f}/{tw:6.1f}{' ' if same else '*'} ({by / ts / 1e3 / 8000:.2f})"
    print(row)
