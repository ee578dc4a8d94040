import gc, os, sys, torch
sys.path.insert(0, '.')
import bench
from ralf_amd.engine import GraphedAdamW
from ralf_amd.helpers import task
from ralf_amd.synthetic import make_batch
dev = torch.device("cuda", 0)
model = bench.build_model(dev, 10, "bfloat16")
opt = GraphedAdamW(params=model.optim_groups(base_lr=1e-4, weight_decay=0.01), max_norm=0.1, loss_lag=1)
batches = [make_batch(64, 10, seed=21 + i) for i in range(3)]
for it in range(8):
    inputs, targets = model.preprocess(batches[it % 3])
    ring = list(task._PINNED.values())[0]["bufs"]
    print(it, len(ring), [torch._C._storage_Use_Count(s[0].untyped_storage()._cdata) for s in ring], {k: (v.device.type if torch.is_tensor(v) else type(v).__name__) for k, v in inputs.items()})
    if it == 1:
        s0 = ring[0][0]
        refs = [type(r).__name__ + ":" + (str(list(r.keys())[:6]) if isinstance(r, dict) else str(r)[:80]) for r in gc.get_referrers(s0)]
        print("referrers of ring[0] base:", refs)
        for o in gc.get_objects():
            if torch.is_tensor(o) and o is not s0 and o.device.type == "cpu" and o.numel() and o.untyped_storage().data_ptr() == s0.untyped_storage().data_ptr():
                print("  alias:", tuple(o.shape), [type(r).__name__ + ":" + (str(list(r.keys())[:8]) if isinstance(r, dict) else str(r)[:60]) for r in gc.get_referrers(o)][:4])
