import gc, os, sys, torch
sys.path.insert(0, '.')
import bench
from ralf_amd.engine import GraphedAdamW
from ralf_amd.helpers import task
from ralf_amd.helpers.task import get_condition
from ralf_amd.synthetic import make_batch
dev = torch.device("cuda", 0)
model = bench.build_model(dev, 10, "bfloat16")
opt = GraphedAdamW(params=model.optim_groups(base_lr=1e-4, weight_decay=0.01), max_norm=0.1, loss_lag=1)
b = make_batch(64, 10, seed=21)
cnt = lambda: [torch._C._storage_Use_Count(s[0].untyped_storage()._cdata) for s in list(task._PINNED.values())[0]["bufs"]]
cond, inputs = get_condition(b, model.auxilary_task, model.tokenizer)
print("after get_condition (cond alive)", cnt())
seqc = model.preprocessor(cond); print("after preprocessor", cnt())
data = model.tokenizer.encode(inputs); print("after encode", cnt())
image = cond.image
_inputs = {"seq": data["seq"][:, :-1], "tgt_key_padding_mask": ~data["mask"][:, :-1], "image": image, "retrieved": inputs["retrieved"],
           "seq_layout_const": seqc["seq"], "seq_layout_const_pad_mask": seqc["pad_mask"]}
out = model._upload_batch(_inputs, {"seq": data["seq"][:, 1:]}); print("after upload", cnt())
del _inputs, image; print("del _inputs,image", cnt())
del cond; print("del cond", cnt())
del seqc, data; print("del seqc,data", cnt())
del inputs; print("del inputs", cnt())
del out; torch.cuda.synchronize(); gc.collect(); print("del out + sync + gc", cnt())
