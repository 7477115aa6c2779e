"""CPU restatement of muon.pp.neighbors - weighted nearest neighbours (TEST INFRASTRUCTURE).

PARITY PINNED (r04) to the reference EXECUTING, with two stated exceptions.  /root/reference/muon/_core/preproc.py
cannot be imported as it is (numba, umap-learn, pynndescent and scanpy are absent) and the reference has no test for
this function (tests/test_muon_preproc.py covers the filters only) - but tests/golden/make_wnn_golden.py loads the file
where it lies with stubs for the THIRD-PARTY pieces only and runs its own `neighbors()` (:264-640) and `l2norm`
(:182-262); tests/test_wnn.py compares this oracle with the fixture it wrote (tests/golden/wnn_golden.npz): modality
weights 4e-16, multimodal distances 1e-16, identical neighbour sets in every row.  The exceptions: (i) the reference
finds neighbours with UMAP's NN-descent (approximate, seeded: `nearest_neighbors(...)`, :453-461 and :525-533) - the
stub and this oracle search exhaustively, i.e. give the exact answer NN-descent approximates; (ii) the UMAP
connectivities come from scanpy (`_compute_connectivities_umap`, :615-622): `fuzzy_simplicial_set` below restates
umap-learn's published algorithm and is NOT pinned.  Partial modality overlap follows the reference's intent, not its
row indexing (see `neighbors`).  This file follows the reference statement by statement - every block cites its
lines.  numpy loops, small inputs only.

Also here: the exact k-nearest-neighbour graph + UMAP connectivities in the slots scanpy's
`sc.pp.neighbors` writes (`.obsp["distances"]`, `.obsp["connectivities"]`, `.uns["neighbors"]`): the
input the reference requires per modality (:366-373).
"""
import numpy as np
from scipy.sparse import coo_matrix, csr_matrix
from scipy.spatial.distance import cdist
from scipy.special import softmax


# ---- umap-learn's fuzzy simplicial set, as scanpy calls it (scanpy/neighbors/_connectivity.py `umap`:
#      set_op_mix_ratio = 1, local_connectivity = 1; umap/umap_.py smooth_knn_dist / compute_membership_strengths)
def smooth_knn_dist(dists, k, n_iter=64, local_connectivity=1.0, bandwidth=1.0):
    target = np.log2(k) * bandwidth
    n = dists.shape[0]
    rho, sigma = np.zeros(n), np.zeros(n)
    mean_all = dists.mean()
    for i in range(n):
        lo, hi, mid = 0.0, np.inf, 1.0
        d = dists[i]
        nz = d[d > 0.0]
        if nz.size >= local_connectivity:
            index = int(np.floor(local_connectivity))
            interp = local_connectivity - index
            if index > 0:
                rho[i] = nz[index - 1]
                if interp > 1e-5:
                    rho[i] += interp * (nz[index] - nz[index - 1])
            else:
                rho[i] = interp * nz[0]
        elif nz.size > 0:
            rho[i] = nz.max()
        for _ in range(n_iter):
            psum = 0.0
            for j in range(1, d.size):
                x = d[j] - rho[i]
                psum += np.exp(-x / mid) if x > 0 else 1.0
            if abs(psum - target) < 1e-5:
                break
            if psum > target:
                hi = mid
                mid = (lo + hi) / 2.0
            else:
                lo = mid
                mid = mid * 2 if hi == np.inf else (lo + hi) / 2.0
        sigma[i] = mid
        if rho[i] > 0.0:
            m = d.mean()
            if sigma[i] < 1e-3 * m:
                sigma[i] = 1e-3 * m
        elif sigma[i] < 1e-3 * mean_all:
            sigma[i] = 1e-3 * mean_all
    return sigma, rho


def fuzzy_simplicial_set(knn_indices, knn_dists, n_obs, n_neighbors):
    knn_dists = knn_dists.astype(np.float32).astype(np.float64)  # (umap works on float32 distances)
    sigma, rho = smooth_knn_dist(knn_dists, float(n_neighbors))
    rows = np.repeat(np.arange(n_obs), knn_indices.shape[1])
    cols = knn_indices.reshape(-1)
    d = knn_dists.reshape(-1)
    val = np.where(cols == rows, 0.0,
                   np.where((d - rho[rows] <= 0.0) | (sigma[rows] == 0.0), 1.0,
                            np.exp(-(d - rho[rows]) / np.where(sigma[rows] == 0, 1.0, sigma[rows]))))
    res = coo_matrix((val, (rows, cols)), shape=(n_obs, n_obs)).tocsr()
    res.eliminate_zeros()
    t = res.T.tocsr()
    prod = res.multiply(t)
    out = (res + t - prod).tocsr()
    out.eliminate_zeros()
    return out


def knn_graph(X, n_neighbors=15, metric="euclidean"):
    """What sc.pp.neighbors(adata, n_neighbors, use_rep=...) leaves behind for `muon.pp.neighbors`, with
    exhaustive search: distances (n_neighbors - 1 stored per row: scanpy counts the cell itself),
    connectivities, the `.uns["neighbors"]` record."""
    X = np.asarray(X, dtype=np.float64)
    n = X.shape[0]
    D = cdist(X, X, metric=metric)
    np.fill_diagonal(D, -1.0)  # the cell itself comes first, like in scanpy's knn arrays
    idx = np.argsort(D, axis=1, kind="stable")[:, :n_neighbors]
    dist = np.take_along_axis(D, idx, axis=1)
    dist[:, 0] = 0.0
    conn = fuzzy_simplicial_set(idx, dist, n, n_neighbors)
    rows = np.repeat(np.arange(n), n_neighbors - 1)
    distances = csr_matrix((dist[:, 1:].reshape(-1), (rows, idx[:, 1:].reshape(-1))), shape=(n, n))
    uns = {"connectivities_key": "connectivities", "distances_key": "distances",
           "params": {"n_neighbors": n_neighbors, "method": "umap", "metric": metric}}
    return distances, conn, uns


def neighbors(reps, graphs, graph_metrics=None, n_neighbors=None, n_bandwidth_neighbors=20, n_multineighbors=200,
              metric="euclidean", eps=1e-4, present=None):
    """``reps``: {modality: X (n x p) dense}; ``graphs``: {modality: CSR distances (n x n) of its own kNN graph};
    everything is laid out over the n observations of the MuData object.  ``present``: {modality: bool mask} of the
    cells a modality has (default: all) - the rows of ``reps`` / ``graphs`` of the others are ignored.  The
    reference's bookkeeping for partial overlap (:381-384, :451, :546-575) is followed in INTENT: a modality
    contributes to a cell's weights, to the neighbourhood means and to the affinity of a pair only where it has the
    cells involved, and its ratio for a cell it lacks stays at the initial -inf (weight 0); the reference's own loops
    index the joint graph with modality-local row numbers there (:586-593).
    Returns (distances CSR n x n with n_neighbors + 1 per row, connectivities, weights n x M, sigmas)."""
    mods = list(reps)
    n = reps[mods[0]].shape[0]
    M = len(mods)
    present = present or {}
    pres = {m: np.asarray(present.get(m, np.ones(n, dtype=bool)), dtype=bool) for m in mods}
    graph_metrics = graph_metrics or {m: "euclidean" for m in mods}
    if n_neighbors is None:  # :375-377
        ks = np.array([int(np.diff(graphs[m].indptr).max()) + 1 for m in mods])
        n_neighbors = int(round(np.mean(ks[ks > 0]), 0))
    ratios = np.full((n, M), -np.inf)
    sigmas = {}
    for i1, m1 in enumerate(mods):
        X = np.asarray(reps[m1], dtype=np.float64)
        G1 = graphs[m1].tocsr()
        S1 = np.nonzero(pres[m1])[0]
        nnd = np.zeros(n)
        nnd[S1] = [G1[i].data.min() for i in S1]  # :389-398
        # :400-461  the n_bandwidth_neighbors cells with the lowest non-zero Jaccard index of the kNN
        # sets, ties broken towards the larger Euclidean distance (metric :53-77: N (1 - jaccard
        # distance) + (bbox - euclid) / bbox, N + 1 without overlap); N = the modality's own cells
        n1 = len(S1)
        bbox = np.linalg.norm(np.ptp(X[S1], axis=0))
        sets = {i: set(G1.indices[G1.indptr[i]:G1.indptr[i + 1]]) for i in S1}
        csig = np.ones(n)
        for i in S1:
            cand = []
            for j in S1:
                if j == i:
                    continue
                inter = len(sets[i] & sets[j])
                if inter == 0:
                    continue
                jac_dist = 1.0 - inter / len(sets[i] | sets[j])
                if jac_dist < 1.0:
                    e = np.linalg.norm(X[i] - X[j])
                    cand.append(((n1 - jac_dist * n1) + (bbox - e) / bbox, j, e))
            cand.sort(key=lambda t: (t[0], t[1]))
            picked = cand[:n_bandwidth_neighbors]
            csig[i] = np.mean([c[2] for c in picked])  # :463-472
        thetas, cur = [], None
        for i2, m2 in enumerate(mods):  # :484-506
            G2 = graphs[m2].tocsr()
            both = pres[m1] & pres[m2]
            th = np.full(n, -np.inf)
            for i in np.nonzero(both)[0]:
                cols = G2.indices[G2.indptr[i]:G2.indptr[i + 1]]
                cols = cols[(G2.data[G2.indptr[i]:G2.indptr[i + 1]] != 0) & pres[m1][cols]]  # (`.nonzero()`, :495)
                r = X[cols].mean(axis=0)
                th[i] = np.exp(-max(np.linalg.norm(X[i] - r) - nnd[i], 0) / (csig[i] - nnd[i]))
            if i1 == i2:
                cur = th
            else:
                thetas.append(th)
        with np.errstate(invalid="ignore"):
            ratio = cur / (np.max(np.stack(thetas, axis=1), axis=1) + eps)  # :507
        ratios[pres[m1], i1] = ratio[pres[m1]]
        sigmas[m1] = csig
    weights = softmax(ratios, axis=1)  # :510
    # :517-575  candidates: the union of every modality's n_multineighbors nearest neighbours (among its own cells)
    pattern = np.zeros((n, n), dtype=bool)
    for m in mods:
        S = np.nonzero(pres[m])[0]
        X = np.asarray(reps[m], dtype=np.float64)[S]
        D = cdist(X, X, metric=graph_metrics[m])
        np.fill_diagonal(D, np.inf)
        k = min(n_multineighbors, len(S) - 1)
        idx = np.argsort(D, axis=1, kind="stable")[:, :k]
        pattern[np.repeat(S, k), S[idx.reshape(-1)]] = True
    aff = np.zeros((n, n))
    for i, m in enumerate(mods):  # :579-609
        X = np.asarray(reps[m], dtype=np.float64)
        if metric in ("seuclidean", "mahalanobis"):
            # scipy takes V / VI from the rows of EACH call, and the reference calls once per cell over that cell's
            # candidates (:596-606): the same loop here (candidates absent from the modality left out)
            for cell in np.nonzero(pres[m])[0]:
                cols = np.nonzero(pattern[cell] & pres[m])[0]
                if len(cols):
                    d = cdist(X[None, cell, :], X[cols, :], metric=metric)[0]
                    aff[cell, cols] += np.exp(-d / sigmas[m][cell]) * weights[cell, i]
            continue
        D = cdist(X, X, metric=metric)
        term = np.exp(-D / sigmas[m][:, None]) * weights[:, i][:, None]
        aff += np.where(pres[m][:, None] & pres[m][None, :], term, 0.0)
    dist = np.sqrt(0.5 * (1.0 - aff))  # :610
    # :612  the n_neighbors + 1 smallest per row among the candidates
    k1 = n_neighbors + 1
    knn_idx = np.empty((n, k1), dtype=np.int64)
    knn_d = np.empty((n, k1))
    for i in range(n):
        cols = np.nonzero(pattern[i])[0]
        o = np.argsort(dist[i, cols], kind="stable")[:k1]
        knn_idx[i], knn_d[i] = cols[o], dist[i, cols][o]
    distances = csr_matrix((knn_d.reshape(-1), knn_idx.reshape(-1), np.arange(0, n * k1 + 1, k1)), shape=(n, n))
    conn = fuzzy_simplicial_set(knn_idx, knn_d, n, k1)  # :615-622
    return distances, conn, weights, sigmas, n_neighbors
