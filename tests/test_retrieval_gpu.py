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
    # the embedding cache in faiss' flat-index file format: written above, re-read here instead of embedding again
    from ralf_amd.retrieval.faiss_io import index_cache_path, read_flat_index

    vec, metric = read_flat_index(index_cache_path("pku", "saliency", str(tmp_path)))
    assert metric == 0 and np.array_equal(vec, X.astype(np.float32))

    def no_embedding(_):
        raise AssertionError("the cached index must be used")

    r2 = Retriever(db_dataset=db, max_seq_length=10, top_k=16, dataset_name="pku", retrieval_backbone="saliency", cache_dir=str(tmp_path), feature_fn=no_embedding)
    assert r2.preprocess_retrieval_cache("val", val, top_k=32, queries=Qv) == t_val


def test_layout_features_and_mmr_rerank(golden):
    """8f rank 1-2: batched layout-encoder embeddings of a whole split == the oracle's FIDNet restatement per row, and the
    MMR re-rank driver == a straightforward per-sample computation on those embeddings."""
    from oracle import ralf_oracle as O
    from oracle.detweights import det_state_dict
    from ralf_amd.functional import Runtime
    from ralf_amd.nn import LayoutEncoder
    from ralf_amd.retrieval import layout_features, maximal_marginal_relevance, rerank_tables
    from test_model_cpu import ref_shapes

    shapes = {k[len("layout_encoer."):]: v for k, v in dict(ref_shapes("ralf_state_shapes.json")).items() if k.startswith("layout_encoer.")}
    sd = det_state_dict({"layout_encoer." + k: v for k, v in shapes.items()})
    enc = LayoutEncoder(num_label=3)
    enc.load_state_dict({k[len("layout_encoer."):]: v for k, v in sd.items()}, strict=True)
    enc = enc.cuda()
    rng = np.random.default_rng(5)
    M, N = 700, 10
    n_el = rng.integers(1, N + 1, M)
    mask = torch.from_numpy(np.arange(N)[None, :] < n_el[:, None])
    fields = {"label": torch.from_numpy(rng.integers(0, 3, (M, N))) * mask, "mask": mask}
    for k in ("center_x", "center_y", "width", "height"):
        fields[k] = torch.from_numpy(rng.random((M, N)).astype(np.float32)) * mask
    feats = layout_features(enc, fields, Runtime(torch.float32), batch_rows=256)
    ref = O.fidnet_extract(sd, "layout_encoer", {k: v for k, v in fields.items()})
    torch.testing.assert_close(feats.cpu(), ref, atol=2e-4, rtol=1e-4)

    K, top_k = 32, 16
    ids = list(range(50))
    table = {i: rng.choice(M, K, replace=False).tolist() for i in ids}
    scores = {i: np.sort(rng.random(K).astype(np.float32))[::-1].copy() for i in ids}
    out = rerank_tables(table, scores, feats, top_k, "mmr", lam=0.5, chunk=16)
    f = ref.numpy().astype(np.float64)
    for i in ids:
        p = f[table[i]]
        nrm = np.linalg.norm(p, axis=1)
        cos = (p @ p.T) / np.maximum(nrm[:, None] * nrm[None, :], 1e-8)
        local = maximal_marginal_relevance(scores[i], cos, 0.5, top_k, "similarity")
        assert out[i] == np.asarray(table[i])[local].tolist()
    rnd = rerank_tables(table, None, feats, top_k, "random")
    assert all(len(set(v)) == top_k and set(v) <= set(table[i]) for i, v in rnd.items())


def test_merged_and_cross_dataset_tables(tmp_path):
    """retriever.py:231-343 (merged backbones) and cross_retriever.py:133-207 on the HIP scan vs the CPU oracle."""
    from oracle import knn_oracle
    from ralf_amd.retrieval import cross_dataset_table, merge_retrieval_cache, merged_vectors

    rng = np.random.default_rng(3)
    a, b = rng.standard_normal((500, 64)).astype(np.float32), rng.standard_normal((500, 32)).astype(np.float32)
    for where in ("before_concat", "after_concat"):
        db = merged_vectors([a, b], where)
        assert db.shape == (500, 96)
        qs = np.stack([merged_vectors([a[i], b[i]], where) for i in range(40)])
        t = merge_retrieval_cache("pku", "train", ["clip", "saliency"], [a, b], [a[:40], b[:40]], [str(7000 + i) for i in range(40)], 8, where, cache_dir=str(tmp_path))
        ref, _ = knn_oracle.topk_ip(db, qs, 9)
        assert all(t[7000 + i] == ref[i, 1:].tolist() for i in range(40))
        assert (tmp_path / f"pku_train_merge_clip_saliency_{where}__topk8.pt").exists()
    refv, q = rng.standard_normal((300, 256)).astype(np.float32), rng.standard_normal((25, 256)).astype(np.float32)
    t = cross_dataset_table("cgl", "pku", "test", "saliency", refv, q, [f"id{i}" for i in range(25)], 16, save_scores=True, cache_dir=str(tmp_path))
    ref, _ = knn_oracle.topk_ip(refv, q, 17)
    assert all(t[f"id{i}"] == ref[i].tolist() for i in range(25))
    assert (tmp_path / "source_cgl_reference_pku_test_saliency_cross_dataset_indexes_top_k16.pt").exists()
    assert (tmp_path / "source_cgl_reference_pku_test_saliency_cross_dataset_scores_top_k16.pt").exists()


def test_preprocess_drivers_end_to_end_on_a_saved_dataset(tmp_path):
    """the command-line drivers themselves (image2layout/preprocess/build_retrieval_indexes.py:42-121, rerank_indexes.py:86-148):
    build_retrieval_indexes.main() and rerank_indexes.main() run end to end on a tiny `datasets.DatasetDict.save_to_disk` directory
    (saliency backbone, FIDNetV3 checkpoint in the reference's model_best.pth.tar layout, DDP-prefixed keys) and write the reference's
    table files; the tables equal the CPU oracle's search / a per-sample MMR."""
    import datasets as ds

    from oracle import knn_oracle
    from ralf_amd.nn import LayoutEncoder
    from ralf_amd.preprocess import build_retrieval_indexes, rerank_indexes
    from ralf_amd.retrieval import coarse_saliency, load_cache_table, maximal_marginal_relevance, table_path

    rng = np.random.default_rng(9)
    feats = ds.Features({"id": ds.Value("string"), "saliency": ds.Array3D((1, 32, 24), "float32"),
                         "label": ds.Sequence(ds.ClassLabel(names=["text", "logo", "underlay"])),
                         **{k: ds.Sequence(ds.Value("float32")) for k in ("center_x", "center_y", "width", "height")}})

    def split(n, base):
        rows = {k: [] for k in feats}
        for i in range(n):
            m = int(rng.integers(1, 11))
            rows["id"].append(str(base + i))
            rows["saliency"].append(rng.random((1, 32, 24)).astype(np.float32))
            rows["label"].append(rng.integers(0, 3, m).tolist())
            for k in ("center_x", "center_y", "width", "height"):
                rows[k].append(rng.random(m).astype(np.float32).tolist())
        return ds.Dataset.from_dict(rows, features=feats)
    dd = ds.DatasetDict({"train": split(150, 1000), "val": split(20, 5000), "test": split(20, 7000), "with_no_annotation": split(30, 9000)})
    root = tmp_path / "data" / "pku10"
    dd.save_to_disk(str(root))
    dd_cgl = ds.DatasetDict({"train": split(120, 20000), "with_no_annotation": split(25, 30000)})
    dd_cgl.save_to_disk(str(tmp_path / "data" / "cgl"))
    cache = tmp_path / "cache"
    build_retrieval_indexes.main(["--dataset_name", "pku", "--dataset_path", str(tmp_path / "data"), "--retrieval_backbone", "saliency",
                                  "--top_k", "32", "--save_scores", "--cache_dir", str(cache)])
    X = np.stack([coarse_saliency(torch.as_tensor(np.asarray(e["saliency"]))) for e in dd["train"]])
    tables = {}
    for sp in ("train", "val", "test"):
        tables[sp] = load_cache_table(table_path("pku", sp, "saliency", 32, str(cache)), 32)
        Q = np.stack([coarse_saliency(torch.as_tensor(np.asarray(e["saliency"]))) for e in dd[sp]])
        ref, _ = knn_oracle.topk_ip(X, Q, 33)
        for i, e in enumerate(dd[sp]):
            assert tables[sp][int(e["id"])] == (ref[i, 1:] if sp == "train" else ref[i, :32]).tolist(), (sp, i)
        assert (cache / f"pku_{sp}_saliency_wo_head_table_between_dataset_indexes_top_k32.pt").exists()
        assert (cache / f"pku_{sp}_saliency_wo_head_table_between_dataset_scores_top_k32.pt").exists()
    # the cross-dataset driver (image2layout/preprocess/build_retrieval_indexes_cross_dataset.py:40-105): the unannotated split of each dataset
    # against the other one's train-split index, all top_k + 1 hits kept
    from ralf_amd.preprocess import build_retrieval_indexes_cross_dataset
    build_retrieval_indexes_cross_dataset.main(["--dataset_path", str(tmp_path / "data"), "--retrieval_backbone", "saliency", "--top_k", "16", "--save_scores",
                                                "--cache_dir", str(cache)])
    emb = lambda d: np.stack([coarse_saliency(torch.as_tensor(np.asarray(e["saliency"]))) for e in d])   # noqa: E731
    for src, ref_name, dsrc, dref in (("pku", "cgl", dd, dd_cgl), ("cgl", "pku", dd_cgl, dd)):
        t = load_cache_table(str(cache / f"source_{src}_reference_{ref_name}_with_no_annotation_saliency_cross_dataset_indexes_top_k16.pt"), 17)
        want, _ = knn_oracle.topk_ip(emb(dref["train"]), emb(dsrc["with_no_annotation"]), 17)
        assert len(t) == len(dsrc["with_no_annotation"])
        for i, e in enumerate(dsrc["with_no_annotation"]):
            assert t[int(e["id"]) if src == "pku" else e["id"]] == want[i].tolist(), (src, i)
        assert (cache / f"source_{src}_reference_{ref_name}_with_no_annotation_saliency_cross_dataset_scores_top_k16.pt").exists()
    # FIDNetV3 checkpoint as a DDP run would have written it: keys prefixed with "module." (the loader strips it and refuses a
    # checkpoint that leaves encoder keys unfilled)
    enc = LayoutEncoder(num_label=3)
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for p in enc.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.05)
    wdir = tmp_path / "fidnet"
    wdir.mkdir()
    torch.save({"state_dict": {"module." + k: v for k, v in enc.state_dict().items()}}, str(wdir / "model_best.pth.tar"))
    rerank_indexes.main(rerank_indexes.parse(["--dataset", "pku", "--dataset_path", str(tmp_path / "data"), "--top_k", "16", "--retrieval_backbone", "saliency",
                                              "--rerank_pool_size", "32", "--rerank_type", "mmr", "--rerank_mmr_lam", "0.5", "--fid_weight_dir", str(wdir),
                                              "--cache_dir", str(cache)]))
    from ralf_amd.functional import Runtime
    from ralf_amd.retrieval import RetrievalDatasetWrapper, layout_features
    fields = RetrievalDatasetWrapper._layout_table(dd["train"], 10)
    f = layout_features(enc.cuda().eval(), fields, Runtime(torch.float32).to(torch.device("cuda")), device=torch.device("cuda")).cpu().numpy().astype(np.float64)
    for sp in ("val", "test"):
        out = torch.load(str(cache / f"pku_{sp}_saliency_rerank_mmr_lam_0.5_wo_head_table_between_dataset_indexes_top_k16.pt"), weights_only=False)
        sc = load_cache_table(table_path("pku", sp, "saliency", 32, str(cache)).replace("indexes", "scores"), 32)
        for i, pool in tables[sp].items():
            p = f[pool]
            nrm = np.linalg.norm(p, axis=1)
            cosm = (p @ p.T) / np.maximum(nrm[:, None] * nrm[None, :], 1e-8)
            local = maximal_marginal_relevance(np.asarray(sc[i]), cosm, 0.5, 16, "similarity")
            assert out[i] == np.asarray(pool)[local].tolist(), (sp, i)
    bad = tmp_path / "bad"
    bad.mkdir()
    torch.save({"state_dict": {"encoder." + k: v for k, v in enc.state_dict().items()}}, str(bad / "model_best.pth.tar"))
    with pytest.raises(RuntimeError, match="encoder keys missing"):
        rerank_indexes._load_fidnet(3, 10, str(bad), torch.device("cuda"))
