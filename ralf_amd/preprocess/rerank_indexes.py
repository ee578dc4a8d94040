"""python -m ralf_amd.preprocess.rerank_indexes  --  the reference's image2layout/preprocess/rerank_indexes.py:86-148: re-rank every
sample's retrieved pool (top rerank_pool_size) down to top_k by maximal marginal relevance over the cosine similarity of the pooled
exemplars' layout-encoder features, or at random.  Same arguments and file names as the reference script; the layout features of
the whole database are computed once in large batches (ralf_amd.retrieval.embed.layout_features) and the pool Gram matrices by one
batched GEMM per chunk of samples, instead of K feature extractions per dataloader item."""
from __future__ import annotations

import argparse
import os

import torch


def _load_fidnet(num_label: int, max_bbox: int, weight_dir: str, device):
    """FIDNetV3 encoder weights as the reference loads them (fid/model.py:131-147: model_best.pth.tar with a `state_dict` entry)"""
    from ..nn import LayoutEncoder

    enc = LayoutEncoder(num_label, 256, 4, 4)
    path = os.path.join(weight_dir, "model_best.pth.tar")
    state = torch.load(path, map_location="cpu", weights_only=True)
    state = state.get("state_dict", state)
    state = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in state.items()}   # a DDP-wrapped checkpoint
    own = enc.state_dict()
    res = enc.load_state_dict({k: v for k, v in state.items() if k in own}, strict=False)
    # the features ARE the encoder: a checkpoint that fills none / part of it would silently re-rank with random features
    need = ("emb_label.", "fc_bbox.", "enc_fc_in.", "enc_transformer.")
    missing = [k for k in res.missing_keys if k.startswith(need)]
    if missing:
        raise RuntimeError(f"{path}: FIDNetV3 encoder keys missing from the checkpoint (first: {missing[:4]}, {len(missing)} in all); "
                           f"checkpoint keys look like {list(state)[:3]}")
    return enc.to(device).eval()


def main(args) -> None:
    from ..functional import Runtime
    from ..retrieval.embed import layout_features, rerank_tables
    from ..retrieval.retriever import RetrievalDatasetWrapper, load_cache_table, table_path
    from ._data import load_splits

    device = torch.device("cuda")
    datasets, features = load_splits(args.dataset_path, args.dataset, args.max_seq_length)
    num_label = features["label"].feature.num_classes
    enc = _load_fidnet(num_label, args.max_seq_length, args.fid_weight_dir, device)
    rt = Runtime(torch.bfloat16 if args.bf16 else torch.float32).to(device)
    fields = RetrievalDatasetWrapper._layout_table(datasets["train"], args.max_seq_length)     # database layouts, padded [n_db, N]
    feats = layout_features(enc, fields, rt, device=device)                                       # [n_db, 256] once for all splits
    params = f"rerank_{args.rerank_type}_lam_{args.rerank_mmr_lam}" if args.rerank_type == "mmr" else f"rerank_{args.rerank_type}"
    for split in ["train", "val", "test"]:
        src = table_path(args.dataset, split, args.retrieval_backbone, args.rerank_pool_size, args.cache_dir)
        table_indexes = load_cache_table(src, args.rerank_pool_size)
        table_scores = load_cache_table(src.replace("indexes", "scores"), args.rerank_pool_size) if args.rerank_type == "mmr" else None
        out = rerank_tables(table_indexes, table_scores, feats, args.top_k, args.rerank_type, args.rerank_mmr_lam)
        dst = os.path.join(args.cache_dir, f"{args.dataset}_{split}_{args.retrieval_backbone}_{params}_wo_head_table_between_dataset_indexes_top_k{args.top_k}.pt")
        torch.save(out, dst)
        print(f"{split}: {len(out)} samples re-ranked ({args.rerank_type}) -> {dst}")


def parse(argv=None):
    parser = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    parser.add_argument("--max_seq_length", type=int, default=10)
    parser.add_argument("--dataset", type=str, default="pku")
    parser.add_argument("--dataset_path", type=str, default="/datasets/PosterLayout")
    parser.add_argument("--top_k", type=int, default=32)
    parser.add_argument("--retrieval_backbone", type=str, default="dreamsim")
    parser.add_argument("--rerank_pool_size", type=int, default=128)
    parser.add_argument("--rerank_type", type=str, default="mmr", choices=["mmr", "random"])
    parser.add_argument("--rerank_mmr_lam", type=float, default=1.0)
    parser.add_argument("--fid_weight_dir", type=str, default="tmp/fidnet/pku10")
    parser.add_argument("--cache_dir", type=str, default="cache")
    parser.add_argument("--bf16", action="store_true", help="layout features in bf16 (default fp32, like the reference)")
    return parser.parse_args(argv)


if __name__ == "__main__":
    main(parse())
