"""relation decode probe: decoder steps per sample and time per step (random weights -> many back-tracks)"""
import sys, time, random, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import bench
from ralf_amd.helpers.layout_tokenizer import LabelFeature, LayoutSequenceTokenizer
from ralf_amd.helpers.relationships import relationship_table
from ralf_amd.helpers.task import get_condition
from ralf_amd.models.generator import ConcateAuxilaryTaskConcateCrossAttnRetrievalAugmentedAutoreg as RALF
from ralf_amd.synthetic import make_batch
from ralf_amd import nn as RN
dev = torch.device("cuda"); labels = ["text", "logo", "underlay"]; N = 10; B = 8
batch = make_batch(B, N, seed=9)
random.seed(0); torch.manual_seed(0)
model = RALF(features={"label": LabelFeature(labels)}, tokenizer=LayoutSequenceTokenizer(labels, N), dataset_name="pku", max_seq_length=N, top_k=16,
             retrieval_backbone="dreamsim", saliency_k="None", auxilary_task="relation", compute_dtype="bfloat16", relation_table=relationship_table(batch, labels)).to(dev).eval()
cond, _ = get_condition(batch, "relation", model.tokenizer)
cond.retrieved = {k: v for k, v in cond.retrieved.items() if k != "image"}
calls = [0]; orig = RALF._StepGraphs.__call__
def counted(self, *a, **k):
    calls[0] += 1
    return orig(self, *a, **k)
RALF._StepGraphs.__call__ = counted
cfg = {"name": "top_k", "top_k": 5, "temperature": 1.0}
model.sample(cond=cond, sampling_cfg=cfg, cond_type="relation", RELATION_SIZE=10)
calls[0] = 0
torch.cuda.synchronize(); t0 = time.perf_counter()
model.sample(cond=cond, sampling_cfg=cfg, cond_type="relation", RELATION_SIZE=10)
torch.cuda.synchronize(); t = time.perf_counter() - t0
print(f"B={B}: {t:.2f} s, {calls[0]} decoder steps = {calls[0]/B:.0f} per sample, {t/calls[0]*1e3:.2f} ms per step")
