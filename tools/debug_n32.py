"""N = 32 decode: KV-cached step logits against the full-prefix decoder, teacher-forced with the full-prefix tokens"""
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: E402,F401
from test_model_cpu import build  # noqa: E402
from oracle.detweights import det_state_dict, resnet50_fpn_shapes  # noqa: E402
from ralf_amd import nn as RN  # noqa: E402
from ralf_amd.helpers.task import get_condition  # noqa: E402
from ralf_amd.helpers.sampling import forced_tokens_all  # noqa: E402
from ralf_amd.synthetic import make_batch  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
mc = build(task="c", N=N)
shapes = {k: tuple(v.shape) for k, v in mc.state_dict().items()}
shapes.update(resnet50_fpn_shapes())
mc.load_state_dict(det_state_dict(shapes), strict=True)
mc = mc.cuda().eval()
cond, _ = get_condition(make_batch(8, N, H=64, W=64, seed=5), "c", mc.tokenizer)
cond.retrieved = {k: v for k, v in cond.retrieved.items() if k != "image"}
with torch.no_grad():
    mc.set_task_preprocessor("c") if mc.use_multitask else None
    enc_in, _ = mc._create_encoder_inputs(cond.to(torch.device("cuda")) if hasattr(cond, "to") else cond)
    dev = torch.device("cuda")
    ids = mc.special_token_ids
    mc.rt.to(dev).begin_step()
    memory = mc._encode_into_memory(enc_in)["memory"]
    B = memory.shape[0]
    T = mc.tokenizer.max_token_length
    mc._token_mask_dev(dev)
    tm = mc._token_mask_u8
    cond_seq = cond.seq.to(dev)
    forced = forced_tokens_all(cond_seq, "c", ids["pad"], ids["eos"], T)
    seq = torch.full((B, 1), ids["bos"], dtype=torch.long, device=dev)
    cache = RN.decoder_init_cache(mc.decoder, memory, mc.rt, T)
    worst = 0.0
    for i in range(T):
        full = mc.decoder(seq, memory, mc.rt, seq == ids["pad"])[:, i].float()
        kpm = (seq == ids["pad"]).to(torch.uint8).contiguous()
        inc = RN.decoder_step(mc.decoder, seq[:, i].contiguous(), i, cache, mc.rt, kpm).float()
        d = (full - inc).abs().max().item()
        ta = RN.ops.mask_sample(full, tm[i], forced[i], 0, 1, 1.0, mc.rt.seed, 1000 + i)
        tb = RN.ops.mask_sample(inc, tm[i], forced[i], 0, 1, 1.0, mc.rt.seed, 1000 + i)
        ne = (ta != tb).nonzero().flatten().tolist()
        worst = max(worst, d)
        if ne or d > 1e-3 or i % 20 == 0:
            top = full.clone()
            top[:, ~mc._token_mask_cache[i]] = -1e30
            t2 = top.topk(2, dim=1).values
            print(f"step {i:3d} max|dlogit| {d:.3e} token mismatch rows {ne} npad {int(kpm.sum())} top2 gap (min over rows) {(t2[:, 0] - t2[:, 1]).min().item():.3e}", flush=True)
        seq = torch.cat([seq, ta.view(B, 1)], dim=1)
    print("worst", worst)

with torch.no_grad():
    cfg = {"name": "deterministic"}
    a = mc.decode_tokens(enc_in, cond_seq, "c", cfg, True)
    b = mc.decode_tokens(enc_in, cond_seq, "c", cfg, False)
    ne = (a != b).nonzero()
    print("decode_tokens cached(static buffers) vs full: mismatches", ne.shape[0], ne[:10].tolist())
    if ne.shape[0]:
        r, c = ne[0].tolist()
        print("row", r, "col", c, "cached", a[r, max(0, c - 3):c + 3].tolist(), "full", b[r, max(0, c - 3):c + 3].tolist(), "teacher", seq[r, 1 + max(0, c - 3):1 + c + 3].tolist())
    print("full == teacher-forced", bool((b == seq[:, 1:]).all()), " cached == teacher-forced", bool((a == seq[:, 1:]).all()))
