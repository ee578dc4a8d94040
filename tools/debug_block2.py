import sys
import torch, torch.nn.functional as F
sys.path.insert(0, ".")
from ralf_amd import nn as RN
torch.manual_seed(0)
inpl, planes, stride, ds, H, W = 1024, 512, 2, True, 8, 10
blk = RN.Bottleneck(inpl, planes, stride, ds)
for n, p in blk.named_parameters():
    with torch.no_grad():
        if p.ndim == 4: p.normal_(0, (2.0 / (p.shape[1] * p.shape[2] * p.shape[3])) ** 0.5)
        elif n.endswith("weight"): p.uniform_(0.5, 1.5)
        else: p.normal_(0, 0.1)
x = torch.randn(2, inpl, H, W).requires_grad_(True)
P = {n: p.detach().clone().requires_grad_(True) for n, p in blk.named_parameters()}
def bn(t, pre): return F.batch_norm(t, None, None, P[pre + ".weight"], P[pre + ".bias"], True, 0.1, 1e-5)
T = {}
def keep(name, t): t.retain_grad(); T[name] = t; return t
c1 = keep("c1", F.conv2d(x, P["conv1.weight"])); b1 = keep("b1", torch.relu(bn(c1, "bn1")))
c2 = keep("c2", F.conv2d(b1, P["conv2.weight"], None, stride, 1)); b2 = keep("b2", torch.relu(bn(c2, "bn2")))
c3 = keep("c3", F.conv2d(b2, P["conv3.weight"])); 
dsx = keep("ds", F.conv2d(x, P["downsample.0.weight"], None, stride)); idn = keep("idn", bn(dsx, "downsample.1"))
out = torch.relu(bn(c3, "bn3") + idn)
go = torch.randn_like(out); out.backward(go)
blk = blk.cuda(); rt = RN.Runtime(torch.float32).to(torch.device("cuda")); rt.training = True
xd = x.detach().permute(0, 2, 3, 1).contiguous().cuda().requires_grad_(True)
G = {}
def hook(name):
    def f(g): G[name] = g.detach().clone()
    return f
y1 = blk.conv1(xd, rt); y1.register_hook(hook("c1")); z1 = blk.bn1(y1, rt, True); z1.register_hook(hook("b1"))
y2 = blk.conv2(z1, rt); y2.register_hook(hook("c2")); z2 = blk.bn2(y2, rt, True); z2.register_hook(hook("b2"))
y3 = blk.conv3(z2, rt); y3.register_hook(hook("c3"))
d0 = blk.downsample[0](xd, rt); d0.register_hook(hook("ds")); d1 = blk.downsample[1](d0, rt, False); d1.register_hook(hook("idn"))
od = blk.bn3(y3, rt, True, res=d1)
od.backward(go.permute(0, 2, 3, 1).contiguous().cuda())
for k in ["idn", "ds", "c3", "b2", "c2", "b1", "c1"]:
    r = T[k].grad; g = G[k].cpu().permute(0, 3, 1, 2)
    e = (g - r).abs()
    print(k, "rel max err", (e.max() / r.abs().max()).item(), "bad frac", (e > 1e-4 * r.abs().max()).float().mean().item())
    if k == "b1":
        bad = (e > 1e-3 * r.abs().max()).nonzero()
        print("  bad h", sorted(set(bad[:, 2].tolist())), "w", sorted(set(bad[:, 3].tolist())), "n", len(bad), "b", sorted(set(bad[:, 0].tolist())))
        print("  fwd act err", (z1.detach().cpu().permute(0,3,1,2) - b1.detach()).abs().max().item())
