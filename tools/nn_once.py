"""data-gradient (NN) GEMM shapes for rocprofv3 --pmc FETCH_SIZE: operands rotated over > 256 MB so the Infinity Cache does not hide re-reads."""
import sys, torch
sys.path.insert(0, ".")
from ralf_amd import ops
dt = torch.bfloat16
def run(M, Nout, Kred, res=False, reps=4):
    As = [torch.randn(M, Kred, device="cuda").to(dt) for _ in range(reps)]
    W = torch.randn(Kred, Nout, device="cuda").to(dt)
    R = [torch.randn(M, Nout, device="cuda").to(dt) for _ in range(reps)] if res else None
    out = torch.empty(M, Nout, device="cuda", dtype=dt)
    for i in range(reps):
        ops.gemm(As[i], W, M, Nout, Kred, b_kcontig=False, out=out, res=R[i] if res else None)
    torch.cuda.synchronize()
run(262144, 64, 256)          # conv3 dgrad layer1: A 134 MB, out 33.5 MB           grid 4096 blocks
run(16384, 256, 1024)         # FFN2... dgrad: A 33.5 MB, out 8.4 MB               grid 1024
run(262144, 256, 64, True)    # conv1 dgrad layer1: A 33.5 MB, res 134 MB, out 134 MB
run(65536, 128, 512)          # conv3 dgrad layer2: A 67 MB, out 16.8 MB
run(16384, 1024, 256)         # FFN1 dgrad
