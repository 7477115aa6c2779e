"""SURVEY 8f.4 on the device: mu.pp.neighbors / knn through HipBackend (neighbourhood means on the
row-stream SpMM kernel, searches and the fuzzy simplicial set as device tensor operations) against
oracle/wnn_oracle.py, and downstream of the LSI: X_lsi -> knn -> weighted nearest neighbours."""
import numpy as np
import pytest
import scipy.sparse as sp

from muon_amd import AnnData, MuData
from muon_amd import atac as ac
from muon_amd._core import preproc as pp
from oracle import wnn_oracle
from tests.synth import planted_topics_csr
from tests.test_wnn import two_modalities

pytestmark = pytest.mark.gpu


def test_knn_and_wnn_match_the_oracle_on_the_gpu(hip):
    lab, x1, x2 = two_modalities(260, 4)
    md = MuData({"rna": AnnData(x1.copy()), "atac": AnnData(x2.copy())})
    for name, m in md.mod.items():
        pp.knn(m, n_neighbors=15, use_rep="X", backend=hip)
        D, C, _ = wnn_oracle.knn_graph(m.X, 15)
        assert (abs(m.obsp["distances"] - D) > 1e-9).nnz == 0 and (abs(m.obsp["connectivities"] - C) > 1e-6).nnz == 0
    pp.neighbors(md, n_multineighbors=50, backend=hip)
    ref = wnn_oracle.neighbors({"rna": x1, "atac": x2}, {k: v.obsp["distances"] for k, v in md.mod.items()},
                               n_multineighbors=50)
    D, C, W, _sig, k = ref
    got = md.obsp["distances"]
    # (the neighbourhood means run in f32 on the SpMM kernel: weights to 1e-4, the graph itself identical
    #  up to exact ties)
    np.testing.assert_allclose(md.obs["rna:mod_weight"].values, W[:, 0], rtol=2e-4, atol=2e-5)
    same = np.mean(got.indices == D.indices)
    assert same > 0.995, same
    m = got.indices == D.indices
    np.testing.assert_allclose(got.data[m], D.data[m], rtol=1e-4, atol=1e-6)
    assert abs(md.obsp["connectivities"] - C).max() < 5e-3
    w1 = md.obs["rna:mod_weight"].values
    assert w1[lab <= 1].mean() > w1[lab >= 2].mean()


def test_lsi_embedding_feeds_weighted_neighbours(hip):
    """The consumer of X_lsi: tfidf -> lsi on a planted-topic ATAC matrix, a second modality with the same
    clusters, knn per modality on the device, then mu.pp.neighbors: neighbours share the planted topic."""
    n = 1200
    X = planted_topics_csr(n, 1500, n_topics=6, density=0.06, seed=3, dtype=np.float32)
    atac = AnnData(X)
    ac.pp.tfidf(atac, backend=hip)
    ac.tl.lsi(atac, n_comps=10, backend=hip)
    emb = atac.obsm["X_lsi"][:, 1:]  # (first component = depth)
    from sklearn.cluster import KMeans

    lab = KMeans(6, n_init=4, random_state=0).fit_predict(emb)
    rng = np.random.default_rng(0)
    rna = AnnData(rng.standard_normal((6, 12))[lab] * 2 + rng.standard_normal((n, 12)))
    atac.obsm["X_lsi"] = emb.copy()
    md = MuData({"rna": rna, "atac": atac})
    pp.knn(md.mod["rna"], n_neighbors=20, use_rep="X", backend=hip)
    pp.knn(md.mod["atac"], n_neighbors=20, use_rep="X_lsi", backend=hip)
    pp.l2norm(md, rep=["X", "X_lsi"])
    pp.neighbors(md, n_multineighbors=100, backend=hip)
    g = md.obsp["distances"]
    assert g.shape == (n, n) and np.all(np.diff(g.indptr) == 21)
    agree = np.mean(lab[g.indices] == np.repeat(lab, 21))
    assert agree > 0.9, agree
    assert md.uns["neighbors"]["params"]["use_rep"] == {"rna": "X", "atac": "X_lsi"}
    c = md.obsp["connectivities"]
    assert c.shape == (n, n) and abs(c - c.T).max() < 1e-12 and c.max() <= 1.0 + 1e-12


@pytest.mark.parametrize("n,p,k,metric", [(9000, 7, 12, "euclidean"), (20000, 50, 200, "euclidean"),
                                          (12000, 33, 30, "cosine"), (8300, 130, 15, "sqeuclidean")])
def test_filter_kernel_search_equals_the_tiled_search(hip, n, p, k, metric):
    """csrc/knn.hip behind device_knn: same neighbours and distances as the tiled tensor search (no backend)."""
    import torch

    rng = np.random.default_rng(n + k)
    lab = rng.integers(0, 15, n)
    X = rng.standard_normal((15, p))[lab] * 1.5 + rng.standard_normal((n, p))
    Xd = hip.to_device(X)
    i0, d0 = pp.device_knn(Xd, k, metric)
    i1, d1 = pp.device_knn(Xd, k, metric, backend=hip)
    assert torch.allclose(d0, d1, rtol=0, atol=1e-11)
    assert float((i0 == i1).double().mean()) > 0.9999  # (exact ties aside)


def test_filter_kernel_counts_and_buffers(hip):
    """The kernel alone against a dense evaluation: every candidate of the panel below the threshold, nothing
    else, the query itself left out, counts beyond the capacity reported."""
    import torch

    rng = np.random.default_rng(3)
    n, p, pp_ = 700, 10, 12
    X = torch.zeros((n, pp_), dtype=torch.float64, device=hip.device)
    X[:, :p] = hip.to_device(rng.standard_normal((n, p)))
    sq = (X * X).sum(dim=1)
    thr = hip.to_device(rng.uniform(3.0, 9.0, n))
    self_pos = torch.arange(n, device=hip.device, dtype=torch.int32)
    for c_lo, c_hi, cap in ((0, n, 64), (130, 517, 8)):
        bp = torch.full((n, cap), -1, dtype=torch.int32, device=hip.device)
        bd = torch.zeros((n, cap), dtype=torch.float64, device=hip.device)
        cnt = torch.zeros((n,), dtype=torch.int32, device=hip.device)
        hip.knn_filter(X, X, sq, sq, thr, self_pos, c_lo, c_hi, bp, bd, cnt)
        D = sq[:, None] + sq[None, :] - 2.0 * (X @ X.T)
        ok = D < thr[:, None]
        ok.fill_diagonal_(False)
        ok[:, :c_lo] = False
        ok[:, c_hi:] = False
        margin = (D - thr[:, None]).abs() < 1e-9  # rounding of the two evaluation orders
        assert not bool((margin & ~torch.eye(n, dtype=torch.bool, device=hip.device)).any())
        assert torch.equal(cnt.long(), ok.sum(dim=1))
        for i in range(0, n, 37):
            m = min(int(cnt[i]), cap)
            got = bp[i, :m].long()
            assert bool(ok[i, got].all()) and got.unique().numel() == m
            assert torch.allclose(bd[i, :m], D[i, got], rtol=0, atol=1e-11)
            if int(cnt[i]) <= cap:
                assert set(got.tolist()) == set(torch.nonzero(ok[i])[:, 0].tolist())


def test_indices_only_search_returns_the_same_neighbour_sets(hip):
    import torch

    rng = np.random.default_rng(9)
    lab = rng.integers(0, 10, 15000)
    X = hip.to_device(rng.standard_normal((10, 20))[lab] * 1.5 + rng.standard_normal((15000, 20)))
    i0, d0 = pp.device_knn(X, 150, "euclidean", backend=hip)
    i1, d1 = pp.device_knn(X, 150, "euclidean", backend=hip, indices_only=True)
    assert torch.equal(torch.sort(i0, dim=1).values, torch.sort(i1, dim=1).values)
    assert torch.allclose(d0, d1, rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("n,p,k", [(3000, 12, 15), (20000, 50, 20)])
def test_bandwidth_kernel_equals_the_tensor_formulation(hip, n, p, k):
    """csrc/wnn.hip (a wave per cell) against A A^T + pair distances + three sorts (the backend without the kernel)."""
    import torch

    rng = np.random.default_rng(n)
    lab = rng.integers(0, 12, n)
    X = rng.standard_normal((12, p))[lab] * 1.5 + rng.standard_normal((n, p))
    a = AnnData(X.copy())
    pp.knn(a, n_neighbors=k, use_rep="X", backend=hip)
    G = a.obsp["distances"]
    Xd = hip.to_device(X)

    class _NoKernel:
        def __getattr__(self, name):
            if name == "wnn_bandwidth":
                raise AttributeError(name)
            return getattr(hip, name)

    want = pp._bandwidths(_NoKernel(), Xd, G, 20)
    got = pp._bandwidths(hip, Xd, G, 20)
    assert torch.allclose(got, want, rtol=1e-12, atol=0, equal_nan=True)  # (sums in a different order)


def test_bandwidth_kernel_reports_hub_cells(hip):
    """A cell listed by more cells than the wave's buffer holds: the overflow flag sends the call to the tensor path."""
    import torch

    n = 9000  # (the wave's buffer holds 8192 entries)
    rng = np.random.default_rng(1)
    X = rng.standard_normal((n, 4))
    X[0] = 0.0
    idx = np.zeros((n, 3), dtype=np.int64)  # every cell lists cell 0 (a hub) and two others
    idx[:, 1] = (np.arange(n) + 1) % n
    idx[:, 2] = (np.arange(n) + 2) % n
    idx[0, 0] = 3
    G = sp.csr_matrix((np.ones(idx.size), idx.reshape(-1), np.arange(0, 3 * n + 1, 3)), shape=(n, n))
    G.sort_indices()
    Xd = hip.to_device(X)
    cs, over = hip.wnn_bandwidth(Xd, hip.to_device(G.indptr, np.int64), hip.to_device(G.indices, np.int32),
                                 hip.to_device(G.T.tocsr().indptr, np.int64), hip.to_device(G.T.tocsr().indices, np.int32),
                                 20, 5.0)
    assert over
    out = pp._bandwidths(hip, Xd, G, 20)  # falls back, finite everywhere
    assert bool(torch.isfinite(out).all())


def test_umap_strengths_kernel_equals_the_tensor_bisection(hip):
    """csrc/wnn.hip k_umap_strengths (a thread per row) against the 64-step tensor bisection it replaces, incl.
    rows with duplicates (zero distances) and a row that lists itself."""
    import torch

    rng = np.random.default_rng(4)
    n, k = 5000, 21
    d = np.sort(rng.gamma(2.0, 1.0, (n, k)), axis=1)
    d[:, 0] = 0.0
    d[::50, 1:4] = 0.0  # duplicated points
    idx = rng.integers(0, n, (n, k))
    idx[:, 0] = np.arange(n)
    dd, ii = hip.to_device(d), hip.to_device(idx, np.int64)
    want = pp.fuzzy_simplicial_set(ii, dd, n, k)
    got = pp.fuzzy_simplicial_set(ii, dd, n, k, backend=hip)
    assert got.shape == want.shape and abs(got - want).max() < 1e-6


@pytest.mark.parametrize("kc,cap", [(28, 148), (208, 688), (5, 3), (64, 960)])
def test_merge_kernel_selects_the_kc_smallest_pairs(kc, cap):
    """mu_knn_merge_f64: list ++ buffer -> the kc smallest (distance, position) pairs, ascending, ties by position"""
    import torch

    from muon_amd._backend import get_backend

    be = get_backend()
    g = torch.Generator(device=be.device).manual_seed(kc)
    n = 3000
    cur_d = torch.rand((n, kc), generator=g, device=be.device, dtype=torch.float64)
    cur_d[5] = 0.25  # ties inside the list
    cur_p = torch.stack([torch.randperm(10**6, generator=g, device=be.device)[:kc] for _ in range(8)])[torch.arange(n, device=be.device) % 8].contiguous()
    buf_d = torch.rand((n, cap), generator=g, device=be.device, dtype=torch.float64)
    buf_d[5, :3] = 0.25
    buf_pos = torch.randint(10**6, 2 * 10**6, (n, cap), generator=g, device=be.device, dtype=torch.int32)
    cnt = torch.randint(0, cap + 20, (n,), generator=g, device=be.device, dtype=torch.int32)
    cnt[0], cnt[1] = 0, cap
    got_d, got_p, thr = be.knn_merge(cur_d, cur_p, buf_d, buf_pos, cnt)
    c = torch.clamp(cnt, max=cap).long()
    valid = torch.arange(cap, device=be.device)[None, :] < c[:, None]
    d_all = torch.cat([cur_d, torch.where(valid, buf_d, torch.full_like(buf_d, float("inf")))], dim=1)
    p_all = torch.cat([cur_p, buf_pos.long()], dim=1)
    # reference order: by (distance, position): sort by position first, then stably by distance
    o = torch.argsort(p_all, dim=1)
    d_s, p_s = torch.gather(d_all, 1, o), torch.gather(p_all, 1, o)
    o = torch.argsort(d_s, dim=1, stable=True)[:, :kc]
    want_d, want_p = torch.gather(d_s, 1, o), torch.gather(p_s, 1, o)
    assert torch.equal(got_d, want_d) and torch.equal(got_p, want_p)
    assert torch.equal(thr, want_d[:, -1])


@pytest.mark.parametrize("tag", ["a", "b"])
def test_neighbors_on_the_gpu_against_the_reference_executing(golden_dir, tag):
    """the HIP path on the inputs of tests/golden/wnn_golden.npz against what /root/reference/muon/_core/preproc.py:264-640
    itself produced for them (tests/golden/make_wnn_golden.py: third-party search / connectivities stubbed)"""
    from muon_amd import AnnData, MuData
    from muon_amd._backend import get_backend
    from muon_amd._core import preproc as pp
    from tests.test_wnn import _golden_case, _same_graph

    be = get_backend()
    x1, x2, graphs, kw, want = _golden_case(golden_dir, tag)
    md = MuData({"rna": AnnData(x1.copy()), "atac": AnnData(x2.copy())})
    for m, k in zip(("rna", "atac"), want["k"]):
        md.mod[m].obsp["distances"] = graphs[m]
        md.mod[m].uns["neighbors"] = {"connectivities_key": "connectivities", "distances_key": "distances",
                                      "params": {"n_neighbors": k, "method": "umap", "metric": "euclidean"}}
    pp.neighbors(md, backend=be, **kw)
    assert md.uns["neighbors"]["params"]["n_neighbors"] == want["n_neighbors"]
    np.testing.assert_allclose(np.asarray(md.obs["rna:mod_weight"]), want["w_rna"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(np.asarray(md.obs["atac:mod_weight"]), want["w_atac"], rtol=0, atol=1e-6)
    _same_graph(md.obsp["distances"], want["dist"], 1e-6)
