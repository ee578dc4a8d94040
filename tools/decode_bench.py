"""Constrained decode benchmark (BASELINE config 5): B = 256, tasks c / cwh, deterministic and top-k sampling;
`relation` (per-sample back-tracking decode with host-side constraint logic, like the reference) on a synthetic relationship table."""
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from ralf_amd.helpers.task import get_condition  # noqa: E402
from ralf_amd.synthetic import make_batch  # noqa: E402


def main(B=256, N=10):
    dev = torch.device("cuda")
    for dtype in ("bfloat16",):
        for task in ("c", "cwh"):
            model = bench.build_model(dev, N, dtype, task).eval()
            batch = make_batch(B, N, seed=9)
            cond, _ = get_condition(batch, task, model.tokenizer)
            cond.retrieved = {k: v for k, v in cond.retrieved.items() if k != "image"}
            for name, cfg in (("deterministic", {"name": "deterministic"}), ("top_k5", {"name": "top_k", "top_k": 5, "temperature": 1.0})):
                from ralf_amd.engine import GraphedDecode
                for kv, graph in ((True, True), (True, False), (False, False)):
                    torch.manual_seed(0)
                    dec = GraphedDecode(model, task, cfg, kv) if graph else None
                    for _ in range(2):
                        model.sample(cond=cond, sampling_cfg=cfg, cond_type=task, use_kv_cache=kv, decoder=dec)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    reps = 3
                    for _ in range(reps):
                        out = model.sample(cond=cond, sampling_cfg=cfg, cond_type=task, use_kv_cache=kv, decoder=dec)
                    torch.cuda.synchronize()
                    t = (time.perf_counter() - t0) / reps
                    print(f"{dtype} task={task:3s} sampling={name:13s} kv_cache={kv!s:5s} graph={graph!s:5s}: {t*1e3:8.1f} ms/batch  {t/B*1e3:6.3f} ms/sample  {B*5*N/t:9.0f} tokens/s", flush=True)


def relation(B=64, N=10, dtype="bfloat16"):
    import random

    from ralf_amd.helpers.layout_tokenizer import LabelFeature, LayoutSequenceTokenizer
    from ralf_amd.helpers.relationships import relationship_table
    from ralf_amd.models.generator import ConcateAuxilaryTaskConcateCrossAttnRetrievalAugmentedAutoreg as RALF

    dev = torch.device("cuda")
    labels = ["text", "logo", "underlay"]
    batch = make_batch(B, N, seed=9)
    random.seed(0)
    torch.manual_seed(0)
    model = RALF(features={"label": LabelFeature(labels)}, tokenizer=LayoutSequenceTokenizer(labels, N), dataset_name="pku", max_seq_length=N, top_k=16,
                 retrieval_backbone="dreamsim", saliency_k="None", auxilary_task="relation", compute_dtype=dtype,
                 relation_table=relationship_table(batch, labels)).to(dev).eval()
    cond, _ = get_condition(batch, "relation", model.tokenizer)
    cond.retrieved = {k: v for k, v in cond.retrieved.items() if k != "image"}
    cfg = {"name": "top_k", "top_k": 5, "temperature": 1.0}
    model.sample(cond=cond, sampling_cfg=cfg, cond_type="relation", RELATION_SIZE=10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out, vio = model.sample(cond=cond, sampling_cfg=cfg, cond_type="relation", RELATION_SIZE=10, return_violation=True)
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    print(f"{dtype} task=relation (back-tracking, RELATION_SIZE=10, synthetic table) B={B}: {t*1e3:8.1f} ms/batch  {t/B*1e3:6.3f} ms/sample  "
          f"violations {vio['viorated']}/{vio['total']}", flush=True)


if __name__ == "__main__":
    main()
    relation()
