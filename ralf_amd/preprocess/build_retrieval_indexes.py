"""python -m ralf_amd.preprocess.build_retrieval_indexes  --  the reference's image2layout/preprocess/build_retrieval_indexes.py:14-78 on
the MI355X scan: for every split, the top-(k+1) inner-product neighbours of each sample in the train-split index, written as
cache/{dataset}_{split}_{backbone}_wo_head_table_between_dataset_indexes_top_k{K}.pt (+ the score table with --save_scores).

Same arguments as the reference script.  The reference embeds one image per call with DreamSim / CLIP / VGG (third-party models, out
of scope here): for those backbones the embeddings are taken from the reference's own cache file
cache/{dataset}_{backbone}_wo_head_index.faiss (database) and from --query_embeddings (an .npz with one [n, D] fp32 array per split);
the `saliency` backbone is computed here (16x16 coarse saliency, retrieval/image.py:35-44)."""
from __future__ import annotations

import argparse

import numpy as np

DATASETS = ["pku", "cgl"]
RETRIEVAL_BACKBONES = ["saliency", "clip", "vgg", "dreamsim"]   # (the reference lists the first three and defaults to the fourth)


def preprocess_retriever(dataset_path: str = "/datasets/PosterLayout", dataset_name: str = "pku", max_seq_length: int = 10,
                         retrieval_backbone: str = "saliency", top_k: int = 32, save_scores: bool = False, query_embeddings: str | None = None,
                         cache_dir: str = "cache", device: str = "cuda") -> None:
    from ..retrieval import Retriever
    from ._data import load_splits

    datasets, features = load_splits(dataset_path, dataset_name, max_seq_length)
    retriever = Retriever(features=features, db_dataset=datasets["train"], max_seq_length=max_seq_length, dataset_name=dataset_name,
                          retrieval_backbone=retrieval_backbone, cache_dir=cache_dir, device=device)
    queries = np.load(query_embeddings) if query_embeddings else None
    for split in datasets.keys():
        q = None
        if queries is not None:
            q = np.asarray(queries[split], np.float32)
        elif retrieval_backbone != "saliency":
            if split != "train":
                raise SystemExit(f"--query_embeddings is needed for the {split} split of backbone {retrieval_backbone}")
            q = retriever.index.vectors.cpu().numpy()   # train split: the queries are the database rows themselves
        table = retriever.preprocess_retrieval_cache(split=split, dataset=datasets[split], top_k=top_k, run_on_local=True, save_scores=save_scores, queries=q)
        print(f"{dataset_name}/{split}: {len(table)} samples x top-{top_k} written to {cache_dir}/")


def main(argv=None) -> None:
    """Pre-compute and cache indexes (and optionally similarity scores) for nearest neighbour search."""
    parser = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    parser.add_argument("--dataset_name", type=str, default="pku", choices=DATASETS)
    parser.add_argument("--dataset_path", type=str)
    parser.add_argument("--retrieval_backbone", type=str, default="dreamsim", choices=RETRIEVAL_BACKBONES)
    parser.add_argument("--top_k", type=int, default=32)
    parser.add_argument("--save_scores", action="store_true", help="some reranking methods needs similarity scores between query and retrieved data")
    parser.add_argument("--max_seq_length", type=int, default=10)
    parser.add_argument("--query_embeddings", type=str, default=None, help=".npz with one [n, D] float32 array per split (third-party backbones)")
    parser.add_argument("--cache_dir", type=str, default="cache")
    args = parser.parse_args(argv)
    preprocess_retriever(dataset_path=args.dataset_path, dataset_name=args.dataset_name, max_seq_length=args.max_seq_length,
                         retrieval_backbone=args.retrieval_backbone, top_k=args.top_k, save_scores=args.save_scores,
                         query_embeddings=args.query_embeddings, cache_dir=args.cache_dir)


if __name__ == "__main__":
    main()
