"""python -m ralf_amd.preprocess.build_retrieval_indexes_cross_dataset  --  the reference's
image2layout/preprocess/build_retrieval_indexes_cross_dataset.py:11-105 on the MI355X scan: every sample of the `with_no_annotation` split of
one dataset searches the TRAIN-split index of the other one (pku -> cgl and cgl -> pku), all top_k + 1 hits are kept (a sample is never
part of the other dataset), written as cache/source_{src}_reference_{ref}_{split}_{backbone}_cross_dataset_indexes_top_k{K}.pt
(+ ..._scores_... with --save_scores) -- models/retrieval/cross_retriever.py:133-207.

Same arguments as the reference script.  As in build_retrieval_indexes, the embeddings of the third-party backbones come from the
reference's own cache files (database: cache/{dataset}_{backbone}_wo_head_index.faiss; queries: --query_embeddings, an .npz with one
[n, D] fp32 array per SOURCE dataset, keys "pku" and "cgl"); the `saliency` backbone is computed here."""
from __future__ import annotations

import argparse

import numpy as np

DATASETS = ["pku", "cgl"]
RETRIEVAL_BACKBONES = ["saliency", "clip", "vgg", "dreamsim"]   # (the reference lists the first three and defaults to the fourth)
SPLIT = "with_no_annotation"


def preprocess_cross_retriever(dataset_path: str = "/datasets/PosterLayout", max_seq_length: int = 10, retrieval_backbone: str = "saliency",
                               top_k: int = 32, save_scores: bool = False, query_embeddings: str | None = None, cache_dir: str = "cache",
                               device: str = "cuda", split: str = SPLIT) -> dict:
    from ..retrieval import Retriever
    from ..retrieval.retriever import _ids_of, cross_dataset_table
    from ._data import load_splits

    data, retr = {}, {}
    for name in DATASETS:   # build_retrieval_indexes_cross_dataset.py:47-76: both datasets, each one's train split is the other's database
        data[name], features = load_splits(dataset_path, name, max_seq_length)
        retr[name] = Retriever(features=features, db_dataset=data[name]["train"], max_seq_length=max_seq_length, dataset_name=name,
                               retrieval_backbone=retrieval_backbone, cache_dir=cache_dir, device=device)
    queries = np.load(query_embeddings) if query_embeddings else None
    tables = {}
    for source, reference in (("pku", "cgl"), ("cgl", "pku")):   # :80-102
        ds_src = data[source][split]
        if queries is not None:
            q = np.asarray(queries[source], np.float32)
        elif retr[source].feature_fn is not None:
            q = np.stack([np.asarray(retr[source].feature_fn(ds_src[i]), np.float32) for i in range(len(ds_src))])
        else:
            raise SystemExit(f"--query_embeddings is needed for backbone {retrieval_backbone}")
        assert q.shape[0] == len(ds_src), f"{source}: {q.shape[0]} query embeddings for {len(ds_src)} samples"
        tables[(source, reference)] = cross_dataset_table(source, reference, split, retrieval_backbone, retr[reference].index.vectors.cpu().numpy(), q,
                                                          list(_ids_of(ds_src)), top_k, save_scores=save_scores, cache_dir=cache_dir, device=device)
        print(f"source {source} -> reference {reference} / {split}: {len(ds_src)} samples x top-{top_k + 1} written to {cache_dir}/")
    return tables


def main(argv=None) -> None:
    """Pre-compute and cache indexes (and optionally similarity scores) for nearest neighbour search."""
    parser = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    parser.add_argument("--dataset_path", type=str)
    parser.add_argument("--retrieval_backbone", type=str, default="dreamsim", choices=RETRIEVAL_BACKBONES)
    parser.add_argument("--top_k", type=int, default=16)
    parser.add_argument("--save_scores", action="store_true", help="some reranking methods needs similarity scores between query and retrieved data")
    parser.add_argument("--max_seq_length", type=int, default=10)
    parser.add_argument("--query_embeddings", type=str, default=None, help=".npz with one [n, D] float32 array per source dataset (third-party backbones)")
    parser.add_argument("--cache_dir", type=str, default="cache")
    args = parser.parse_args(argv)
    preprocess_cross_retriever(dataset_path=args.dataset_path, max_seq_length=args.max_seq_length, retrieval_backbone=args.retrieval_backbone, top_k=args.top_k,
                               save_scores=args.save_scores, query_embeddings=args.query_embeddings, cache_dir=args.cache_dir)


if __name__ == "__main__":
    main()
