import sys
import torch, torch.nn.functional as F
sys.path.insert(0, ".")
from oracle import ralf_oracle as O
from oracle.detweights import det_state_dict, resnet50_fpn_shapes
from ralf_amd import nn as RN, functional as RF
sd = det_state_dict(resnet50_fpn_shapes())
g = torch.Generator().manual_seed(3)
img = torch.rand(2, 4, 128, 160, generator=g)
bb = RN.ResnetFeatureExtractor(256)
bb.load_state_dict({k[len("encoder."):]: v.clone() for k, v in sd.items()}, strict=True)
bb = bb.cuda()
rt = RN.Runtime(torch.float32).to(torch.device("cuda")); rt.training = True
p = "encoder.extractor"; b = p + ".body"
# oracle layer by layer (train mode)
x = F.conv2d(img, sd[b + ".conv1.weight"], None, 2, 3)
acts = {"conv1": x}
x = torch.relu(O._bn(x, sd, b + ".bn1", True)); acts["bn1"] = x
x = F.max_pool2d(x, 3, 2, 1); acts["pool"] = x
for li, (planes, blocks, stride) in enumerate(O.RESNET50_STAGES, start=1):
    for bi in range(blocks):
        q = f"{b}.layer{li}.{bi}"; s = stride if bi == 0 else 1; idn = x
        y = torch.relu(O._bn(F.conv2d(x, sd[q + ".conv1.weight"]), sd, q + ".bn1", True))
        y = torch.relu(O._bn(F.conv2d(y, sd[q + ".conv2.weight"], None, s, 1), sd, q + ".bn2", True))
        y = O._bn(F.conv2d(y, sd[q + ".conv3.weight"]), sd, q + ".bn3", True)
        if (q + ".downsample.0.weight") in sd:
            idn = O._bn(F.conv2d(x, sd[q + ".downsample.0.weight"], None, s), sd, q + ".downsample.1", True)
        x = torch.relu(y + idn); acts[f"layer{li}.{bi}"] = x
# mine
B, C, H, W = img.shape
from ralf_amd import ops
xm = ops.permute4(img.cuda().contiguous(), (B, H, W, 8), (4 * H * W, W, 1, H * W), 4, torch.float32)
body = bb.extractor.body
def cmp(name, t):
    r = acts[name]; t = t.detach().cpu().permute(0, 3, 1, 2)
    print(f"{name:12s} max|ref| {r.abs().max().item():9.4f} mean/std per-channel ratio max {(r.mean((0,2,3)).abs() / r.std((0,2,3)).clamp_min(1e-9)).max().item():8.2f}  max err {(t - r).abs().max().item():.3e}")
xm = body.conv1(xm, rt); cmp("conv1", xm)
xm = body.bn1(xm, rt, True); cmp("bn1", xm)
xm = RF.MaxPoolFn.apply(xm); cmp("pool", xm)
for li in (1, 2, 3, 4):
    for bi, blk in enumerate(getattr(body, f"layer{li}")):
        xm = blk(xm, rt); cmp(f"layer{li}.{bi}", xm)
