"""GPU parity: HIP exact inner-product top-k (through the C ABI) vs the C oracle, bit-exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def unit_rows(n, d, seed):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, d)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x


def run_hip(X, Q, k):
    from ralf_amd.retrieval import knn_topk_ip

    val, idx = knn_topk_ip(torch.from_numpy(X).cuda(), torch.from_numpy(Q).cuda(), k)
    torch.cuda.synchronize()
    return idx.cpu().numpy(), val.cpu().numpy()


def assert_exact(X, Q, k):
    from oracle import knn_oracle as K

    i_ref, v_ref = K.topk_ip(X, Q, k)
    i_hip, v_hip = run_hip(X, Q, k)
    np.testing.assert_array_equal(i_hip, i_ref)
    np.testing.assert_array_equal(v_hip, v_ref)  # bit-exact fp32 scores (fmaf chain == fp32 MFMA chain)


@pytest.mark.parametrize("nq", [1, 5, 16, 17, 32, 40, 64, 100, 200])
@pytest.mark.parametrize("k", [1, 16, 17, 33])
def test_toy_index_ties_and_self_queries(nq, k):
    X = unit_rows(2048, 64, 0)
    X[100] = X[7]; X[200] = X[7]; X[5] = X[1900]; X[2047] = X[0]   # planted exact ties
    Q = np.concatenate([X[: (nq + 1) // 2], unit_rows(nq // 2, 64, 1)])[:nq]  # self + fresh queries
    assert_exact(X, Q, k)


@pytest.mark.parametrize("n,d,nq,k", [(10, 4, 3, 17), (1, 8, 1, 1), (257, 100, 9, 16), (8193, 36, 33, 33), (20000, 256, 130, 256), (300, 512, 70, 300)])
def test_ragged_shapes(n, d, nq, k):
    assert_exact(unit_rows(n, d, 2), unit_rows(nq, d, 3), k)


@pytest.mark.parametrize("d,nq", [(256, 64), (1792, 32), (512, 1), (1792, 16)])
def test_cgl_sized_index(d, nq):
    assert_exact(unit_rows(61548, d, 4), unit_rows(nq, d, 5), 17)


@pytest.mark.parametrize("nq", [1, 20, 50, 100])
@pytest.mark.parametrize("k", [1, 16, 64])
def test_fused_scan_selection_degenerate_rows(nq, k):
    """the fused scan + per-workgroup selection (k <= 64) on inputs that stress the in-LDS selection: whole row chunks of identical
    rows (ties fill a chunk's list), all-zero rows (score -0.0 / +0.0), chunks shorter than k, negative-only scores"""
    rng = np.random.default_rng(nq * 100 + k)
    X = unit_rows(1000, 32, 11)
    X[128:384] = X[130]                      # two full 128-row chunks (one 256-row chunk) of identical rows
    X[500:520] = 0.0                         # zero rows
    X[900:] = -np.abs(X[900:])               # last (short) chunk
    Q = np.concatenate([X[[130, 0, 505]], np.abs(unit_rows(max(nq - 3, 1), 32, 12))])[:nq]
    assert_exact(X, Q, k)
    assert_exact(X[:37], Q, k)               # fewer rows than k = 64: padded with (-inf, -1)


def test_two_stage_certificate_is_a_bound():
    """the certificate's eps (ralf_knn_rownorms + ralf_knn_select_cand) really bounds |coarse - exact| for every (query, row)
    pair: operand rounding to bf16 plus the fp32 accumulation of both passes"""
    from ralf_amd import ops
    from ralf_amd.retrieval.knn import knn_rownorms, knn_scores

    g = torch.Generator(device="cuda").manual_seed(3)
    for scale in (1.0, 37.0):                 # un-normalised rows too
        X = torch.randn(3000, 1792, device="cuda", generator=g) * scale
        Q = torch.randn(64, 1792, device="cuda", generator=g)
        Xb, Qb = ops.cast(X, torch.bfloat16), ops.cast(Q, torch.bfloat16)
        qn, _ = knn_rownorms(Q, Qb)
        xr, xn = knn_rownorms(X, Xb, want_rows=True, want_max=True)
        torch.testing.assert_close(qn[:, 0], Q.norm(dim=1), rtol=1e-5, atol=0)
        torch.testing.assert_close(xr[:, 2], (X - Xb.float()).norm(dim=1), rtol=1e-4, atol=0)
        assert torch.equal(xn, xr.max(dim=0).values)
        coarse = ops.gemm(Qb, Xb, 64, 3000, 1792, out_dtype=torch.float32)
        exact = knn_scores(X, Q)
        eps = qn[:, 2] * xn[0] + qn[:, 1] * xn[2] + 2 * 1792 * 2.0 ** -24 * qn[:, 0] * xn[0]
        assert bool(((coarse - exact).abs() <= eps[:, None]).all())
        assert float(((coarse - exact).abs() / eps[:, None]).max()) > 0.02      # ... and is not vacuous


def test_select_cand_orders_by_score_then_row():
    from ralf_amd.retrieval.knn import knn_select_cand

    rng = np.random.default_rng(2)
    for pool, k in ((65, 16), (200, 33), (1024, 64), (5, 5)):
        cand = np.stack([rng.permutation(5000)[:pool] for _ in range(7)]).astype(np.int64)
        sc = rng.integers(0, 6, (7, pool)).astype(np.float32)       # massive ties
        v, i, bad = knn_select_cand(torch.from_numpy(sc).cuda(), torch.from_numpy(cand).cuda(), k)
        assert bad is None
        for q in range(7):
            order = np.lexsort((cand[q], -sc[q]))[:k]
            np.testing.assert_array_equal(i[q].cpu().numpy(), cand[q][order])
            np.testing.assert_array_equal(v[q].cpu().numpy(), sc[q][order])


def test_select_degenerate_scores():
    from oracle import knn_oracle as K
    from ralf_amd.retrieval.knn import knn_select

    rng = np.random.default_rng(6)
    s = np.zeros((6, 20000), np.float32)
    s[1] = 1.5                                  # all equal -> indices 0..k-1
    s[2] = rng.integers(0, 3, 20000)            # massive ties straddling the threshold
    s[3] = rng.standard_normal(20000); s[3, 17] = np.inf; s[3, 9000] = -np.inf
    s[4] = -rng.random(20000); s[4, ::2] = -0.0
    s[5] = rng.standard_normal(20000).astype(np.float32) * 1e-30
    for k in (1, 17, 64, 1000):
        i_ref, v_ref = K.select(s, k)
        v, i = knn_select(torch.from_numpy(s).cuda(), k)
        np.testing.assert_array_equal(i.cpu().numpy(), i_ref)
        np.testing.assert_array_equal(v.cpu().numpy(), v_ref)


def test_full_size_properties():
    """BASELINE config 4: 61 548 x 1792 index, nq = 1024, k = 16 -- size-independent properties."""
    from oracle import knn_oracle as K

    X = unit_rows(61548, 1792, 7)
    Q = np.concatenate([X[:512], unit_rows(512, 1792, 8)])
    idx, val = run_hip(X, Q, 16)
    assert (idx[:512, 0] == np.arange(512)).all()             # self-match is rank 0 (retriever.py:211-213)
    assert (np.diff(val, axis=1) <= 0).all()                  # sorted
    assert all(len(set(r)) == 16 for r in idx)                # no duplicates
    sub = np.arange(0, 1024, 37)                              # oracle on a bounded sample of the queries
    i_ref, v_ref = K.topk_ip(X, Q[sub], 16)
    np.testing.assert_array_equal(idx[sub], i_ref)
    np.testing.assert_array_equal(val[sub], v_ref)
    # batch-size independence: the same query gives the same answer alone and inside the batch
    i1, v1 = run_hip(X, Q[700:701], 16)
    np.testing.assert_array_equal(i1[0], idx[700]); np.testing.assert_array_equal(v1[0], val[700])


def test_errors_are_loud():
    from ralf_amd._lib import RalfHipError
    from ralf_amd.retrieval import knn_topk_ip

    X = torch.zeros(16, 6, device="cuda"); Q = torch.zeros(2, 6, device="cuda")
    with pytest.raises(RalfHipError, match="multiple of 4"):
        knn_topk_ip(X, Q, 3)
    with pytest.raises(RalfHipError, match="k=0"):
        knn_topk_ip(torch.zeros(16, 8, device="cuda"), torch.zeros(2, 8, device="cuda"), 0)


@pytest.mark.parametrize("N,D,nq,k", [(5000, 256, 300, 16), (61548, 1792, 256, 16), (3000, 64, 200, 5)])
def test_two_stage_search_equals_exhaustive(N, D, nq, k):
    """bf16 coarse pass + exact fp32 re-score + per-query certificate == the exhaustive fp32 scan, bit for bit
    (scores and indices), including near-duplicate rows that defeat the certificate and force the fallback."""
    from ralf_amd import ops
    from ralf_amd.retrieval.knn import FlatIPIndex, knn_topk_ip, knn_topk_ip_two_stage

    g = torch.Generator(device="cuda").manual_seed(N + nq)
    X = torch.randn(N, D, device="cuda", generator=g)
    X /= X.norm(dim=1, keepdim=True)
    X[N // 2: N // 2 + 100] = X[7] + 1e-4 * torch.randn(100, D, device="cuda", generator=g)   # near duplicates: more than the candidate pool
    Q = torch.randn(nq, D, device="cuda", generator=g)
    Q /= Q.norm(dim=1, keepdim=True)
    Q[3] = X[7]                                                                               # a query inside the cluster
    v_ref, i_ref = knn_topk_ip(X, Q, k)
    v, i, nfb = knn_topk_ip_two_stage(X, ops.cast(X, torch.bfloat16), Q, k)
    assert torch.equal(i, i_ref) and torch.equal(v, v_ref)
    assert nfb >= 1 and nfb < nq // 2          # the cluster query falls back, ordinary queries are certified
    index = FlatIPIndex(X)                      # the index front end picks the two-stage path for large batches
    v2, i2 = index.search(Q, k)
    assert torch.equal(i2, i_ref) and torch.equal(v2, v_ref)
    index.max_queries_per_call = 96             # ... and a query set larger than one call takes travels in blocks (the last one ragged): the same table
    v3, i3 = index.search(Q, k)
    assert torch.equal(i3, i_ref) and torch.equal(v3, v_ref)


@pytest.mark.parametrize("filtered", [True, False])
def test_two_stage_filtered_coarse_pass_equals_the_dense_one(filtered):
    """the coarse pass that keeps only the scores above a per-query bound (RalfGemmDesc.flt_*: no [nq, N] score matrix) returns what the dense
    coarse pass returns = the exhaustive scan; an index whose leading slice holds nothing similar to the queries floods the lists
    (more hits than slots), which must be noticed and redone exhaustively"""
    from ralf_amd import ops
    from ralf_amd.retrieval import knn as K

    N, D, nq, k = 20000, 256, 300, 16
    g = torch.Generator(device="cuda").manual_seed(11)
    X = torch.randn(N, D, device="cuda", generator=g)
    X /= X.norm(dim=1, keepdim=True)
    X[9000] = X[40]                                     # exact duplicates inside the candidate range
    Q = torch.randn(nq, D, device="cuda", generator=g)
    Q /= Q.norm(dim=1, keepdim=True)
    Q[5] = X[40]
    v_ref, i_ref = K.knn_topk_ip(X, Q, k)
    v, i, nfb = K.knn_topk_ip_two_stage(X, ops.cast(X, torch.bfloat16), Q, k, filtered=filtered)
    assert torch.equal(i, i_ref) and torch.equal(v, v_ref) and nfb < nq // 4
    if not filtered:
        return
    # flooded lists: the threshold slice is orthogonal to every query, the rest of the index is not
    u = torch.zeros(D, device="cuda"); u[0] = 1.0
    X2 = X.clone()
    X2[:K.FILTER_SAMPLE_ROWS] = -u                      # scores <= 0 against queries with a positive first coordinate
    Q2 = Q.clone(); Q2[:, 0] = Q2[:, 0].abs() + 0.5; Q2 /= Q2.norm(dim=1, keepdim=True)
    X2[K.FILTER_SAMPLE_ROWS:, 0] = X2[K.FILTER_SAMPLE_ROWS:, 0].abs() + 0.5
    X2[K.FILTER_SAMPLE_ROWS:] /= X2[K.FILTER_SAMPLE_ROWS:].norm(dim=1, keepdim=True)
    v_ref, i_ref = K.knn_topk_ip(X2, Q2, k)
    v, i, nfb = K.knn_topk_ip_two_stage(X2, ops.cast(X2, torch.bfloat16), Q2, k, filtered=True)
    assert torch.equal(i, i_ref) and torch.equal(v, v_ref)
    assert nfb == nq                                    # every query has flooded tiles (all 128 columns of a tile pass, 16 slots): all redone exhaustively


@pytest.mark.parametrize("N,D,nq,k", [(20000, 256, 600, 16), (61548, 1792, 1024, 16), (70000, 64, 520, 5), (3000, 128, 513, 16)])
def test_one_call_filtered_search_equals_exhaustive(N, D, nq, k):
    """ralf_knn_topk_ip_two_stage_filtered (threshold product on the index's first rows, filtered product, selection straight from the slot lists,
    re-score, certificate -- one library call) == the exhaustive scan, bit for bit, on ordinary data (a handful of fallbacks at most: an exact
    duplicate pair and a query on it), with more slots than one selection segment (N = 70000: 547 tiles x 16 slots), and on an index whose leading
    rows hold nothing similar to the queries: every tile floods, every query is flagged and redone, and the index front end stops using the form"""
    from ralf_amd import ops
    from ralf_amd.retrieval import knn as K

    g = torch.Generator(device="cuda").manual_seed(N + nq)
    X = torch.randn(N, D, device="cuda", generator=g)
    X /= X.norm(dim=1, keepdim=True)
    X[N // 2] = X[40]
    Q = torch.randn(nq, D, device="cuda", generator=g)
    Q /= Q.norm(dim=1, keepdim=True)
    Q[5] = X[40]
    Xb = ops.cast(X, torch.bfloat16)
    _, xn = K.knn_rownorms(X, Xb, want_rows=False, want_max=True)
    v_ref, i_ref = K.knn_topk_ip(X, Q, k)
    v, i, nfb, ws = K.knn_topk_ip_two_stage_fused(X, Xb, Q, k, xn, filtered=True)
    assert torch.equal(i, i_ref) and torch.equal(v, v_ref) and nfb < nq // 8
    v, i, nfb2, ws = K.knn_topk_ip_two_stage_fused(X, Xb, Q, k, xn, workspace=ws, filtered=False)   # the same workspace serves the dense form
    assert torch.equal(i, i_ref) and torch.equal(v, v_ref)
    index = K.FlatIPIndex(X, filtered_min_queries=512)   # (default 640: BASELINE config 4's crossover)
    assert nq >= index.filtered_min_queries
    v, i = index.search(Q, k)
    assert torch.equal(i, i_ref) and torch.equal(v, v_ref) and index.filtered_min_queries > 0
    if N < 8192:
        return
    ns = 4096
    u = torch.zeros(D, device="cuda"); u[0] = 1.0
    X2 = X.clone()
    X2[:ns] = -u
    Q2 = Q.clone(); Q2[:, 0] = Q2[:, 0].abs() + 0.5; Q2 /= Q2.norm(dim=1, keepdim=True)
    X2[ns:, 0] = X2[ns:, 0].abs() + 0.5
    X2[ns:] /= X2[ns:].norm(dim=1, keepdim=True)
    v_ref, i_ref = K.knn_topk_ip(X2, Q2, k)
    index = K.FlatIPIndex(X2, filtered_min_queries=512)
    v, i = index.search(Q2, k)
    assert torch.equal(i, i_ref) and torch.equal(v, v_ref)
    assert index.last_fallbacks == nq and index.filtered_min_queries == 0
    v, i = index.search(Q2, k)                       # ... and the next batch takes the dense coarse pass
    assert torch.equal(i, i_ref) and torch.equal(v, v_ref) and index.last_fallbacks < nq // 8


def test_sharded_search_forms_on_the_hip_scan():
    """SURVEY 8e on the real scan (1-rank RCCL group; the 2-rank exchange is covered on CPU with gloo): query-sharded replicas and
    an index shard with a row offset return the table of the plain search"""
    import os

    import torch.distributed as dist
    from ralf_amd.retrieval import FlatIPIndex, search_index_sharded, search_query_sharded

    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29000 + os.getpid() % 2000))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    g = torch.Generator().manual_seed(4)
    X = torch.randn(5000, 64, generator=g)
    Q = torch.randn(37, 64, generator=g).cuda()
    index = FlatIPIndex(X)
    s0, i0 = index.search(Q, 16)
    s1, i1 = search_query_sharded(index.search, Q, 16)
    assert torch.equal(i1, i0) and torch.equal(s1, s0)
    s2, i2 = search_index_sharded(index.search, 1000, Q, 16)
    assert torch.equal(i2, i0 + 1000) and torch.equal(s2, s0)


def test_summation_order_risk_at_full_size():
    """faiss-cpu (absent here) sums the D products of a score in an order of its own; the oracle and the HIP scan use the
    ascending-d fmaf chain.  Any two legal fp32 summation orders of one score differ by at most 2*D*2^-24*|q||x| (each is within
    D*u*sum|q_i x_i| <= D*u*|q||x| of the exact value, u = 2^-24).  On the BASELINE config-4 workload (61 548 x 1792, nq = 1024,
    k = 16) this test COUNTS the queries a different order could change -- as a SET (gap between rank k and rank k+1 below the bound)
    and as an ordered LIST (any adjacent gap inside the top k+1 below the bound) -- and asserts that every other query equals the
    float64 NumPy ranking exactly (reference call site: models/retrieval/retriever.py:193-213)."""
    import json
    import os

    N, D, nq, k = 61548, 1792, 1024, 16
    X, Q = unit_rows(N, D, 7), unit_rows(nq, D, 8)
    idx, val = run_hip(X, Q, k + 1)
    X64 = X.astype(np.float64)
    bound = 2.0 * D * 2.0 ** -24          # unit rows and queries: |q||x| = 1
    typical = 2.0 * np.sqrt(D) * 2.0 ** -24   # random-walk size of the same difference (for the report only)
    set_risk = order_risk = set_typ = order_typ = 0
    checked_set = checked_order = 0
    for q0 in range(0, nq, 128):
        S = Q[q0:q0 + 128].astype(np.float64) @ X64.T                      # exact scores to ~1e-16
        order = np.lexsort((np.broadcast_to(np.arange(N), S.shape), -S), axis=1)[:, :k + 1]
        top = np.take_along_axis(S, order, 1)
        gaps = top[:, :-1] - top[:, 1:]                                    # [128, k] exact adjacent gaps, the last one = rank k vs k+1
        for r in range(gaps.shape[0]):
            q = q0 + r
            s_risky, o_risky = gaps[r, -1] < bound, bool((gaps[r] < bound).any())
            set_risk += s_risky; order_risk += o_risky
            set_typ += gaps[r, -1] < typical; order_typ += bool((gaps[r] < typical).any())
            if not s_risky:     # the top-k SET is determined whatever the summation order
                assert set(idx[q, :k].tolist()) == set(order[r, :k].tolist()), q
                checked_set += 1
            if not o_risky:     # ... and so is the ordered list
                np.testing.assert_array_equal(idx[q, :k], order[r, :k])
                checked_order += 1
    report = {"index": f"{N}x{D} fp32 unit rows", "nq": nq, "k": k, "bound_2_D_u": bound, "queries_set_at_risk": int(set_risk),
              "queries_order_at_risk": int(order_risk), "queries_set_at_risk_typical_error": int(set_typ),
              "queries_order_at_risk_typical_error": int(order_typ), "verified_equal_to_float64_set": checked_set,
              "verified_equal_to_float64_ordered": checked_order}
    print("summation-order risk:", json.dumps(report))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "knn_summation_order_risk.json"), "w") as f:
            json.dump(report, f, indent=1)
    assert checked_set + set_risk == nq and checked_set > nq // 2
