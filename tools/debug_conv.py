import sys
import torch, torch.nn.functional as F
sys.path.insert(0, ".")
from ralf_amd import nn as RN, functional as RF
torch.manual_seed(0)
rt = RN.Runtime(torch.float32).to(torch.device("cuda")); rt.training = True
for (Ci, Co, k, s, p, H, W) in [(512, 512, 3, 2, 1, 8, 10), (32, 512, 3, 2, 1, 8, 10), (512, 32, 3, 2, 1, 8, 10), (1024, 2048, 1, 2, 0, 8, 10), (64, 64, 3, 1, 1, 8, 10)]:
    x = torch.randn(2, Ci, H, W).requires_grad_(True)
    w = (torch.randn(Co, Ci, k, k) * 0.05).requires_grad_(True)
    y = F.conv2d(x, w, None, s, p)
    go = torch.randn_like(y)
    y.backward(go)
    xd = x.detach().permute(0, 2, 3, 1).contiguous().cuda().requires_grad_(True)
    wd = w.detach().cuda().requires_grad_(True)
    yd = RF.conv2d(xd, wd, None, s, p, rt)
    yd.backward(go.permute(0, 2, 3, 1).contiguous().cuda())
    e = (xd.grad.cpu().permute(0, 3, 1, 2) - x.grad).abs()
    print((Ci, Co, k, s), "fwd", (yd.detach().cpu().permute(0, 3, 1, 2) - y.detach()).abs().max().item(), "dx max err", e.max().item(), "of", x.grad.abs().max().item(),
          "bad frac", (e > 1e-3).float().mean().item(), "dw", ((wd.grad.cpu() - w.grad).abs().max() / w.grad.abs().max()).item())
    if e.max() > 1e-3:
        bad = (e > 1e-3).nonzero()
        print("   bad idx sample", bad[:8].tolist(), "unique h", sorted(set(bad[:, 2].tolist())), "unique w", sorted(set(bad[:, 3].tolist())), "unique c (first 10)", sorted(set(bad[:, 1].tolist()))[:10])
