import sys
import torch, torch.nn.functional as F
sys.path.insert(0, ".")
from ralf_amd import nn as RN
torch.manual_seed(0)
for (inpl, planes, stride, ds, H, W) in [(2048, 512, 1, False, 4, 5), (1024, 512, 2, True, 8, 10), (64, 64, 1, True, 16, 20)]:
    blk = RN.Bottleneck(inpl, planes, stride, ds)
    for n, p in blk.named_parameters():
        with torch.no_grad():
            if p.ndim == 4: p.normal_(0, (2.0 / (p.shape[1] * p.shape[2] * p.shape[3])) ** 0.5)
            elif n.endswith("weight"): p.uniform_(0.5, 1.5)
            else: p.normal_(0, 0.1)
    x = torch.randn(2, inpl, H, W).requires_grad_(True)
    P = {n: p.detach().clone().requires_grad_(True) for n, p in blk.named_parameters()}
    def bn(t, pre): return F.batch_norm(t, None, None, P[pre + ".weight"], P[pre + ".bias"], True, 0.1, 1e-5)
    y = torch.relu(bn(F.conv2d(x, P["conv1.weight"]), "bn1"))
    y = torch.relu(bn(F.conv2d(y, P["conv2.weight"], None, stride, 1), "bn2"))
    y = bn(F.conv2d(y, P["conv3.weight"]), "bn3")
    idn = bn(F.conv2d(x, P["downsample.0.weight"], None, stride), "downsample.1") if ds else x
    out = torch.relu(y + idn)
    go = torch.randn_like(out)
    out.backward(go)
    blk = blk.cuda()
    rt = RN.Runtime(torch.float32).to(torch.device("cuda")); rt.training = True
    xd = x.detach().permute(0, 2, 3, 1).contiguous().cuda().requires_grad_(True)
    od = blk(xd, rt)
    print("cfg", inpl, planes, stride, ds, "fwd err", (od.detach().cpu().permute(0, 3, 1, 2) - out.detach()).abs().max().item())
    od.backward(go.permute(0, 2, 3, 1).contiguous().cuda())
    print("  dx rel", ((xd.grad.cpu().permute(0, 3, 1, 2) - x.grad).abs().max() / x.grad.abs().max()).item())
    for n, p in blk.named_parameters():
        r = P[n].grad
        print("  ", n, ((p.grad.cpu() - r).abs().max() / r.abs().max().clamp_min(1e-9)).item())
