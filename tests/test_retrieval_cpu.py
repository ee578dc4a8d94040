"""CPU: host side of the retrieval pipeline -- re-rankers vs known answers recorded from the reference,
coarse saliency feature, table file format."""
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


def test_coarse_saliency_batch_equals_per_image():
    from ralf_amd.retrieval import coarse_saliency_batch

    s = torch.rand(5, 1, 350, 240) * 1.2 - 0.1   # values outside [0,1] exercise the clamp
    f = coarse_saliency_batch(s)
    assert f.shape == (5, 256)
    for b in range(5):
        assert np.array_equal(f[b].numpy(), coarse_saliency(s[b]))
