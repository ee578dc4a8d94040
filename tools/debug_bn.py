import sys, torch
sys.path.insert(0, ".")
from ralf_amd import ops
import torch.nn.functional as F
torch.manual_seed(0)
for dtype in (torch.float32, torch.bfloat16):
    M, C = 1000, 64
    x = (torch.randn(M, C) * 2 + 0.5).to(dtype).float().requires_grad_(True)
    res = torch.randn(M, C).to(dtype).float().requires_grad_(True)
    g, b = (1 + 0.1 * torch.randn(C)).requires_grad_(True), (0.1 * torch.randn(C)).requires_grad_(True)
    y = torch.relu(F.batch_norm(x, None, None, g, b, True, 0.1, 1e-5) + res)
    go = torch.randn(M, C).to(dtype).float()
    y.backward(go)
    xd, resd = x.detach().to(dtype).cuda(), res.detach().to(dtype).cuda()
    yd, mean, rstd, mask = ops.bn_forward(xd, g.detach().cuda(), b.detach().cuda(), None, None, True, True, resd, want_mask=True)
    print(dtype, "y err", (yd.float().cpu() - y.detach()).abs().max().item())
    for m in (None, mask):
        dx, dg, db, dres = ops.bn_backward(xd, go.to(dtype).cuda(), yd if m is None else None, g.detach().cuda(), mean, rstd, True, True, mask=m)
        print("  mask" if m is not None else "  y   ", "dx err", (dx.float().cpu() - x.grad).abs().max().item(), "dres err", (dres.float().cpu() - res.grad).abs().max().item(),
              "dg err", (dg.cpu() - g.grad).abs().max().item(), "db err", (db.cpu() - b.grad).abs().max().item())
