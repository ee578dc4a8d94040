"""CPU: host side of the retrieval pipeline -- re-rankers vs known answers recorded from the reference,
coarse saliency feature, table file format."""
import pytest
import os

import numpy as np
import torch

from ralf_amd.retrieval.reranker import maximal_marginal_relevance, reranker_top_k
from ralf_amd.retrieval.retriever import coarse_saliency, load_cache_table, table_path


def test_rerankers_match_reference(golden):
    g = golden("reranker.npz")
    for i in range(5):
        r = g.sub(f"case{i}")
        n, k, lam, sim = r["cfg"].tolist()
        st = "similarity" if sim else "distance"
        q, pair = r["q"].numpy(), r["pair"].numpy()
        assert np.array_equal(maximal_marginal_relevance(q, pair, lam, int(k), st), r["mmr"].numpy())
        assert np.array_equal(reranker_top_k(q, int(k), st), r["topk"].numpy())


def test_coarse_saliency_and_table_format(tmp_path):
    s = torch.rand(1, 350, 240)
    f = coarse_saliency(s)
    ref = torch.nn.functional.interpolate(s[None], size=(16, 16)).flatten()   # nearest, like the reference
    assert f.shape == (256,) and np.allclose(f, (2 * ref.clamp(0, 1) - 1).numpy())
    table = {7: list(range(32)), 9: list(range(32, 64))}
    path = table_path("pku", "train", "dreamsim", 32, str(tmp_path))
    assert path.endswith("pku_train_dreamsim_wo_head_table_between_dataset_indexes_top_k32.pt")   # retriever.py:149
    torch.save(table, path)
    assert load_cache_table(path, 16) == {7: list(range(16)), 9: list(range(32, 48))}           # retrieval_dataset_wrapper.py:32


def test_load_cache_table_reads_reference_written_tables(tmp_path, monkeypatch):
    """the reference saves collections.defaultdict(list) tables with string or int ids (models/retrieval/retriever.py:188-221);
    a missing file falls back to PRECOMPUTED_WEIGHT_DIR/retrieval_indexes/<name> (helpers/retrieval_dataset_wrapper.py:21-27)"""
    import collections

    from ralf_amd.retrieval import retriever as R

    table = collections.defaultdict(list)
    for i in range(3):
        table[f"id{i}"] = [int(j) for j in range(i, i + 32)]
    path = table_path("cgl", "val", "dreamsim", 32, str(tmp_path))
    torch.save(table, path)
    got = load_cache_table(path, 16)
    assert type(got) is dict and got == {f"id{i}": list(range(i, i + 16)) for i in range(3)}
    # fallback directory
    pre = tmp_path / "pre"
    (pre / "retrieval_indexes").mkdir(parents=True)
    name = os.path.basename(path)
    torch.save(table, str(pre / "retrieval_indexes" / name))
    monkeypatch.setattr(R, "PRECOMPUTED_WEIGHT_DIR", str(pre))
    assert load_cache_table(str(tmp_path / "nowhere" / name), 2) == {f"id{i}": [i, i + 1] for i in range(3)}
    with pytest.raises(ValueError, match="Cache not found"):
        load_cache_table(str(tmp_path / "nowhere" / "other.pt"), 2)


def test_coarse_saliency_batch_equals_per_image():
    from ralf_amd.retrieval import coarse_saliency_batch

    s = torch.rand(5, 1, 350, 240) * 1.2 - 0.1   # values outside [0,1] exercise the clamp
    f = coarse_saliency_batch(s)
    assert f.shape == (5, 256)
    for b in range(5):
        assert np.array_equal(f[b].numpy(), coarse_saliency(s[b]))


def test_faiss_flat_index_file_layout_and_round_trip(tmp_path):
    """the on-disk layout of a faiss flat index (faiss 1.7 index_write.cpp), spelled out byte by byte; the reader returns the
    matrix, the writer reproduces the same bytes; non-flat / truncated / inconsistent files fail loudly"""
    import struct

    import numpy as np
    from ralf_amd.retrieval.faiss_io import METRIC_INNER_PRODUCT, METRIC_L2, read_flat_index, write_flat_index

    x = (np.arange(5 * 3, dtype=np.float32).reshape(5, 3) - 7) / 4
    raw = b"IxFI" + struct.pack("<i", 3) + struct.pack("<q", 5) + struct.pack("<qq", 1 << 20, 1 << 20) + b"\x01" + struct.pack("<i", 0)
    raw += struct.pack("<Q", 15) + x.astype("<f4").tobytes()
    p = tmp_path / "pku_saliency_wo_head_index.faiss"
    p.write_bytes(raw)
    got, metric = read_flat_index(str(p))
    assert metric == METRIC_INNER_PRODUCT and got.dtype == np.float32 and np.array_equal(got, x)
    q = tmp_path / "w.faiss"
    write_flat_index(str(q), x)
    assert q.read_bytes() == raw
    write_flat_index(str(q), x, METRIC_L2)
    assert q.read_bytes()[:4] == b"IxF2" and read_flat_index(str(q))[1] == METRIC_L2
    big = np.random.default_rng(0).standard_normal((1000, 257)).astype(np.float32)
    write_flat_index(str(q), big)
    assert np.array_equal(read_flat_index(str(q))[0], big)
    for bad in (raw[:20], raw[:-4], b"IwFl" + raw[4:], raw[:45 - 8] + struct.pack("<Q", 14) + raw[45:], b"IxF2" + raw[4:]):
        p.write_bytes(bad)
        with pytest.raises(ValueError):
            read_flat_index(str(p))
    with pytest.raises(ValueError):
        write_flat_index(str(q), np.zeros(4, np.float32))


def test_preprocess_drivers_expose_the_reference_arguments():
    """python -m ralf_amd.preprocess.{build_retrieval_indexes,rerank_indexes}: option names and defaults of the reference's scripts
    (image2layout/preprocess/build_retrieval_indexes.py:16-31, rerank_indexes.py:151-174)"""
    import argparse

    from ralf_amd.preprocess import build_retrieval_indexes as B
    from ralf_amd.preprocess import rerank_indexes as R

    got = {}

    def fake(**kw):
        got.update(kw)

    orig, B.preprocess_retriever = B.preprocess_retriever, fake
    try:
        B.main(["--dataset_path", "/data", "--save_scores"])
    finally:
        B.preprocess_retriever = orig
    assert got["dataset_name"] == "pku" and got["retrieval_backbone"] == "dreamsim" and got["top_k"] == 32 and got["save_scores"] is True
    assert got["dataset_path"] == "/data"
    with pytest.raises(SystemExit):
        B.main(["--dataset_name", "imagenet"])
    a = R.parse([])
    assert (a.max_seq_length, a.dataset, a.dataset_path, a.top_k, a.retrieval_backbone, a.rerank_pool_size, a.rerank_type, a.rerank_mmr_lam, a.fid_weight_dir) == \
        (10, "pku", "/datasets/PosterLayout", 32, "dreamsim", 128, "mmr", 1.0, "tmp/fidnet/pku10")
    with pytest.raises(SystemExit):
        R.parse(["--rerank_type", "bm25"])
    assert isinstance(a, argparse.Namespace)


def test_score_tables_with_numpy_rows_load(tmp_path):
    """the reference saves its score tables with one numpy array per sample (retriever.py:216-224); load_cache_table reads them
    through the allow-listing unpickler (and still refuses arbitrary classes)"""
    import collections
    import pickle

    import numpy as np
    import pytest
    import torch

    from ralf_amd.retrieval.retriever import load_cache_table

    t = collections.defaultdict(list)
    t[7] = np.arange(5, dtype=np.float32)
    t["a"] = np.linspace(0, 1, 6).astype(np.float32)
    p = tmp_path / "scores.pt"
    torch.save(t, str(p))
    out = load_cache_table(str(p), 3)
    assert isinstance(out, dict) and np.array_equal(out[7], np.arange(3, dtype=np.float32)) and out["a"].shape == (3,)

    class Evil:
        def __reduce__(self):
            return (print, ("pwned",))
    q = tmp_path / "evil.pt"
    torch.save({1: Evil()}, str(q))
    with pytest.raises(pickle.UnpicklingError):
        load_cache_table(str(q), 3)


def test_random_retrieval_wrapper_draws_like_the_reference():
    """`generator.random_retrieval=true` (train/train.py:136-137 -> helpers/random_retrieval_dataset_wrapper.py:75-77): K uniform draws from
    torch's global generator with upper bound len(dataset), used as database rows, no table file; the flag on the ordinary wrapper must select
    the same behaviour (it was accepted and silently ignored: VERDICT r5)"""
    import torch
    from ralf_amd.retrieval import RandomRetrievalDatasetWrapper, RetrievalDatasetWrapper

    g = np.random.default_rng(3)
    db = []
    for i in range(23):
        m = int(g.integers(1, 8))
        db.append({"id": i, "label": g.integers(0, 3, m).tolist(), "center_x": g.random(m).tolist(), "center_y": g.random(m).tolist(),
                   "width": g.random(m).tolist(), "height": g.random(m).tolist()})
    val = db[:9]
    for cls, kw in ((RandomRetrievalDatasetWrapper, {}), (RetrievalDatasetWrapper, {"random_retrieval": True})):
        w = cls("pku", val, db, "val", 4, 10, "dreamsim", saliency_k=None, cache_dir="/nonexistent", **kw)
        assert len(w) == 9 and w.random_retrieval
        torch.manual_seed(11)
        items = [w[i] for i in (0, 5, 8)]
        torch.manual_seed(11)
        for it in items:
            want = torch.randint(low=0, high=9, size=[4]).tolist()
            r = it["retrieved"][0]
            assert r["index"] == want
            for k, j in enumerate(want):
                m = len(db[j]["label"])
                assert r["mask"][k].tolist() == [True] * m + [False] * (10 - m)
                assert r["label"][k, :m].tolist() == db[j]["label"] and r["label"][k, m:].eq(0).all()
                assert torch.allclose(r["width"][k, :m], torch.tensor(db[j]["width"], dtype=torch.float32)) and r["width"][k, m:].eq(0).all()
    # the table-driven wrapper still needs its table
    with pytest.raises(Exception):
        RetrievalDatasetWrapper("pku", val, db, "val", 4, 10, "dreamsim", cache_dir="/nonexistent")
