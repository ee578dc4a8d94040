"""GPU: end-to-end retrieval-table build on the HIP scan (Retriever.preprocess_retrieval_cache) and the
dataset wrapper that materialises the K exemplar layouts."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def fake_dataset(n, seed, N=10):
    rng = np.random.default_rng(seed)
    rows = []
    for i in range(n):
        m = int(rng.integers(1, N + 1))
        rows.append({"id": str(1000 + seed * 10000 + i), "saliency": torch.from_numpy(rng.random((1, 350, 240)).astype(np.float32)),
                     "label": rng.integers(0, 3, m).tolist(), "center_x": rng.random(m).tolist(), "center_y": rng.random(m).tolist(),
                     "width": rng.random(m).tolist(), "height": rng.random(m).tolist()})
    return rows


def test_table_build_and_wrapper(tmp_path):
    from oracle import knn_oracle
    from ralf_amd.retrieval import RetrievalDatasetWrapper, Retriever, coarse_saliency, load_cache_table, table_path

    db, val = fake_dataset(300, 1), fake_dataset(40, 2)
    r = Retriever(db_dataset=db, max_seq_length=10, top_k=16, dataset_name="pku", retrieval_backbone="saliency", cache_dir=str(tmp_path))
    X = np.stack([coarse_saliency(e["saliency"]) for e in db])
    t_train = r.preprocess_retrieval_cache("train", db, top_k=32)
    t_val = r.preprocess_retrieval_cache("val", val, top_k=32, save_scores=True)
    i_ref, _ = knn_oracle.topk_ip(X, X, 33)
    for i, e in enumerate(db):   # train split: rank 0 (the sample itself) dropped
        assert i_ref[i, 0] == i and t_train[int(e["id"])] == i_ref[i, 1:].tolist()
    Qv = np.stack([coarse_saliency(e["saliency"]) for e in val])
    iv, _ = knn_oracle.topk_ip(X, Qv, 33)
    for i, e in enumerate(val):
        assert t_val[int(e["id"])] == iv[i].tolist()
    assert load_cache_table(table_path("pku", "val", "saliency", 32, str(tmp_path)), 16)[int(val[0]["id"])] == iv[0, :16].tolist()
    w = RetrievalDatasetWrapper("pku", val, db, "val", 16, 10, "saliency", cache_dir=str(tmp_path))
    item = w[3]
    ret = item["retrieved"][0]
    assert ret["label"].shape == (16, 10) and ret["mask"].dtype == torch.bool and ret["image"].shape == (16, 4, 1, 1)
    j = ret["index"][5]
    n = len(db[j]["label"])
    assert ret["mask"][5].sum().item() == n and ret["label"][5, :n].tolist() == db[j]["label"]
    assert torch.allclose(ret["center_x"][5, :n], torch.tensor(db[j]["center_x"]))
