"""muon_amd.atac.tl.lsi on the GPU against the reference executed here (golden fixture)
and the f64 ARPACK oracle on seeded planted-topic matrices.
Bar (BASELINE.json north_star): largest principal angle between the spans of the top-k
right singular vectors < 1e-4 rad; singular values within 1e-5 relative."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from muon_amd import AnnData
from muon_amd import atac as ac
from muon_amd._atac.tools import lsi_device
from oracle import lsi_oracle
from tests.synth import planted_topics_csr

pytestmark = pytest.mark.gpu
ANGLE = 1e-4


def test_against_reference_fixture(golden_dir):
    g = np.load(f"{golden_dir}/lsi_golden.npz")
    X = sp.csr_matrix((g["tfidf_data"], g["tfidf_indices"], g["tfidf_indptr"]), shape=tuple(g["tfidf_shape"]))
    for tag, scale in (("scaled", True), ("raw", False)):
        ad = AnnData(X.copy())
        ac.tl.lsi(ad, scale_embeddings=scale, n_comps=12)
        assert lsi_oracle.max_subspace_angle(ad.varm["LSI"], g[f"LSI_{tag}"]) < ANGLE
        assert lsi_oracle.max_subspace_angle(ad.obsm["X_lsi"], g[f"X_lsi_{tag}"]) < 5 * ANGLE
        np.testing.assert_allclose(ad.uns["lsi"]["stdev"], g[f"stdev_{tag}"], rtol=1e-5)
        assert ad.obsm["X_lsi"].dtype == X.dtype


def test_pipeline_tfidf_then_lsi_against_oracle():
    X = planted_topics_csr(6000, 9000, n_topics=50, density=0.03, seed=2, dtype=np.float32)
    ad = AnnData(X.copy())
    ac.pp.tfidf(ad)
    ref = lsi_oracle.lsi(ad.X, n_comps=50)
    ac.tl.lsi(ad, n_comps=50)  # re-uses the device copy left by tfidf
    assert lsi_oracle.max_subspace_angle(ad.varm["LSI"], ref["LSI"]) < ANGLE
    np.testing.assert_allclose(ad.uns["lsi"]["stdev"], ref["stdev"], rtol=1e-5)
    # the leading component is isolated: compare it vector-wise (sign aligned)
    u = lsi_oracle.sign_align(ad.obsm["X_lsi"][:, :1], ref["X_lsi"][:, :1])
    np.testing.assert_allclose(u, ref["X_lsi"][:, :1], atol=2e-3)
    np.testing.assert_allclose(ad.obsm["X_lsi"].mean(axis=0), 0, atol=1e-3)
    np.testing.assert_allclose(ad.obsm["X_lsi"].std(axis=0), 1, rtol=1e-3)


def test_small_widths_and_errors():
    X = planted_topics_csr(300, 200, n_topics=5, density=0.1, seed=4, dtype=np.float64)
    ad = AnnData(X.copy())
    ac.pp.tfidf(ad)
    ref = lsi_oracle.lsi(ad.X, n_comps=5)
    ac.tl.lsi(ad, n_comps=5)
    assert ad.obsm["X_lsi"].dtype == np.float64
    assert lsi_oracle.max_subspace_angle(ad.varm["LSI"], ref["LSI"]) < ANGLE
    with pytest.raises(ValueError):
        ac.tl.lsi(AnnData(sp.random(10, 60, density=0.5, format="csr")), n_comps=50)
    with pytest.raises(TypeError):
        ac.tl.lsi(np.ones((3, 3)))


def test_run_to_run_bit_reproducible():
    X = planted_topics_csr(1500, 2500, n_topics=10, density=0.05, seed=8, dtype=np.float32)
    a, b = AnnData(X.copy()), AnnData(X.copy())
    for ad in (a, b):
        ac.pp.tfidf(ad)
        ac.tl.lsi(ad, n_comps=10)
    assert np.array_equal(a.obsm["X_lsi"], b.obsm["X_lsi"])
    assert np.array_equal(a.varm["LSI"], b.varm["LSI"])


def test_binarize_tfidf_lsi_resident_pipeline(hip, monkeypatch):
    # one upload for the three calls; results equal the oracle's on the binarised counts
    from oracle import tfidf_oracle

    X = planted_topics_csr(5000, 7000, n_topics=30, density=0.03, seed=4, dtype=np.float32)
    uploads = []
    real = hip.upload_csr
    monkeypatch.setattr(hip, "upload_csr", lambda *a, **k: (uploads.append(1), real(*a, **k))[1])
    ad = AnnData(X.copy())
    ac.pp.binarize(ad, backend=hip)
    assert set(np.unique(ad.X.data)) == {1.0}
    ac.pp.tfidf(ad, backend=hip)
    ac.tl.lsi(ad, n_comps=30, backend=hip)
    assert len(uploads) == 1
    B = X.copy()
    B.data[:] = 1
    T = tfidf_oracle.canonical(tfidf_oracle.tfidf(B))
    np.testing.assert_array_equal(ad.X.indices, T.indices)
    np.testing.assert_allclose(ad.X.data, T.data, rtol=1e-5)
    ref = lsi_oracle.lsi(T, n_comps=30)
    assert lsi_oracle.max_subspace_angle(ad.varm["LSI"], ref["LSI"]) < ANGLE


def test_more_components_than_the_block_width():
    X = planted_topics_csr(5000, 6000, n_topics=100, density=0.04, seed=8, dtype=np.float32)
    ad = AnnData(X.copy())
    ac.pp.tfidf(ad)
    ref = lsi_oracle.lsi(ad.X, n_comps=100)
    ac.tl.lsi(ad, n_comps=100)
    assert ad.obsm["X_lsi"].shape == (5000, 100) and ad.varm["LSI"].shape == (6000, 100)
    assert lsi_oracle.max_subspace_angle(ad.varm["LSI"], ref["LSI"]) < ANGLE
    np.testing.assert_allclose(ad.uns["lsi"]["stdev"], ref["stdev"], rtol=1e-5)


def test_baseline_shape_10k_by_30k_against_the_f64_oracle(hip):
    # BASELINE.json configs[0]/[1]: 10 000 cells x 30 000 peaks, 3 % nnz, tfidf + lsi(n_comps=50).
    # The reference path (/root/reference/muon/_atac/tools.py:53-69) restated in f64 is the oracle.
    from muon_amd._atac.tools import lsi_device

    X = planted_topics_csr(10_000, 30_000, n_topics=50, density=0.03, seed=0, dtype=np.float32)
    ad = AnnData(X.copy())
    ac.pp.tfidf(ad, backend=hip)
    ref = lsi_oracle.lsi(ad.X, n_comps=50)
    ac.tl.lsi(ad, n_comps=50, backend=hip)
    angle = lsi_oracle.max_subspace_angle(ad.varm["LSI"], ref["LSI"])
    print(f"10k x 30k: angle {angle:.2e}, stdev rel {np.max(np.abs(ad.uns['lsi']['stdev'] / ref['stdev'] - 1)):.1e}")
    assert angle < ANGLE
    np.testing.assert_allclose(ad.uns["lsi"]["stdev"], ref["stdev"], rtol=1e-5)
    assert lsi_oracle.max_subspace_angle(ad.obsm["X_lsi"], ref["X_lsi"]) < 5 * ANGLE
    from muon_amd._atac.preproc import resident
    _, _, _, info = lsi_device(hip, resident(ad.X, hip), n_comps=50, return_info=True)
    # (the bound is a statement at the 1e-5 ... 1e-4 level; at the f32 floor, ~1e-6, it is only indicative)
    assert info["converged"] and info["angle_bound"] < ANGLE and angle <= max(info["angle_bound"], 1e-5)


@pytest.mark.parametrize("case", ["k_inside_cluster", "k_past_the_planted_rank", "unstructured"])
def test_k_not_at_a_spectral_gap_is_never_silently_wrong(hip, case):
    # n_topics=80 / n_comps=50 (k inside the planted cluster), n_topics=30 /
    # n_comps=50 (k past the planted rank, inside the bulk) and the unstructured generator.  Either
    # `converged` is True and the subspace is within 1e-4 of f64 ARPACK, or `converged` is False;
    # `angle_bound` covers the true angle in every case.  The gap is printed with the angle.
    from muon_amd._atac.tools import lsi_device
    from oracle import tfidf_oracle
    from tests.synth import unstructured_csr

    if case == "k_inside_cluster":
        X = planted_topics_csr(3000, 2500, n_topics=80, density=0.03, seed=3, dtype=np.float32)
    elif case == "k_past_the_planted_rank":
        X = planted_topics_csr(2000, 1500, n_topics=30, density=0.03, seed=4, dtype=np.float32)
    else:
        X = unstructured_csr(2000, 1500, density=0.03, seed=5)
    T = tfidf_oracle.canonical(tfidf_oracle.tfidf(X)).astype(np.float32)
    ref = lsi_oracle.lsi(T, n_comps=50)
    Xd = hip.upload_csr(T.indptr, T.indices, T.data, T.shape)
    _, sd, V, info = lsi_device(hip, Xd, n_comps=50, return_info=True)
    angle = lsi_oracle.max_subspace_angle(hip.to_host(V), ref["LSI"])
    print(f"{case}: gap_rel={info['gap_rel']:.2e} angle={angle:.2e} bound={info['angle_bound']:.2e} "
          f"converged={info['converged']} spmm={info['spmm']}")
    assert angle <= max(info["angle_bound"], 1e-5)
    if info["converged"]:
        assert angle < ANGLE
    np.testing.assert_allclose(sd, ref["stdev"], rtol=1e-5)
    if case == "k_inside_cluster":
        assert info["converged"]  # gap 1.3 %: reachable in f32, and reached


def test_two_ranks_on_one_gpu_match_the_single_process_result():
    """The row-sharded HIP path with real collectives: two processes share cuda:0 (gloo), hold
    uneven shards, all-reduce the per-peak sums, the Grams and Z = X^T Y, and must reproduce the
    single-process TF-IDF values and LSI subspace (scripts/dist_gpu_check.py does the comparison)."""
    import os
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(root, "scripts", "dist_gpu_check.py")],
                       capture_output=True, text=True, timeout=600, env=env, cwd=root)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0 and "dist gpu check ok" in r.stdout


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` outside torch.distributed.run (the form the driver uses): bench.py
    starts its own two ranks (here both on cuda:0 over gloo - the one-GPU test hook) and rank 0 prints
    one JSON line for the row-sharded workload."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(MUON_AMD_BENCH_SHARED_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "c3shard",
                        "--cells", "6000", "--peaks", "9000", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=root)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["config"]["lsi"]["converged"]
    assert out["roofline"]["frac"] > 0 and out["roofline"]["lds_frac"] > 0


def test_bench_rccl_branch_runs_on_one_gpu():
    """MUON_AMD_BENCH_FORCE_DIST=1: bench.py builds the `nccl` (RCCL) process group with device_id and sends the
    exchange steps of tfidf + lsi (column sums, Z, Grams, agree) through it at world size 1 - the multi-GPU branch
    itself (bench.py: init_process_group("nccl", device_id=...)) executes on hardware before an 8-GPU lease does."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(MUON_AMD_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--workload", "c3shard",
                        "--cells", "6000", "--peaks", "9000", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=root)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["config"]["lsi"]["converged"]


def test_rank_deficient_input_takes_the_device_flag_and_the_host_redo(hip):
    """min(n, d) >= 8192 selects the device-side Cholesky of CholeskyQR; a matrix of rank 20 asked for 30
    components exhausts the Krylov space, the device flags the pivot that is not safely positive, the call
    is redone on the host path (which truncates the dependent directions), and the 20 singular values
    that exist come out right (this path had no test)."""
    rng = np.random.default_rng(0)
    n = d = 9000
    base = sp.random(20, d, density=0.02, random_state=rng, format="csr", dtype=np.float32)
    idx = rng.integers(0, 20, n)
    X = (sp.diags((1 + rng.random(n)).astype(np.float32)) @ base[idx]).tocsr()
    X.sort_indices()
    Xd = hip.upload_csr(X.indptr, X.indices, X.data.astype(np.float32), X.shape)
    calls = []
    orig = hip.chol_rinv
    hip.chol_rinv = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        U, stdev, V, info = lsi_device(hip, Xd, n_comps=30, n_obs=n, return_info=True)
    finally:
        del hip.chol_rinv
    assert calls, "the device Cholesky path was not taken first"
    s = stdev * np.sqrt(n - 1)
    ref = np.linalg.svd((base.toarray().astype(np.float64).T * 1.0), compute_uv=False)  # (only for the count)
    assert np.sum(ref > 1e-6) == 20
    from scipy.sparse.linalg import svds

    want = np.sort(svds(X.astype(np.float64), k=20, return_singular_vectors=False))[::-1]
    np.testing.assert_allclose(s[:20], want, rtol=1e-5)
    assert np.all(s[20:] < 1e-3 * s[0]) and np.all(np.isfinite(hip.to_host(U))) and np.all(np.isfinite(hip.to_host(V)))


def test_widened_bench_records_run_small_and_hold_parity(hip):
    """scripts/bench_widened.py (the `secondary` records of SURVEY 8f.2 - 8f.4 in the bench line) at toy
    sizes: every record carries a value and its parity against the oracle / the scipy route."""
    import importlib.util
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_widened", os.path.join(root, "scripts", "bench_widened.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    r = mod.run_ingest(hip, n_cells=3000, n_feat=5000, density=0.03)
    assert r["parity"]["identical"] and r["value"] > 0 and r["config"]["kept_columns"] == 4000
    r = mod.run_mofa_ng(hip, n=600, d_dense=50, d_sparse=300, iters=3, sample=200)
    assert r["parity"]["elbo_max_rel"] < 1e-10 and r["parity"]["Z_max_abs"] < 1e-8 and r["elbo_monotone"]
    r = mod.run_wnn(hip, n=3000, sample=300)
    assert r["parity"]["graph_identical_fraction"] > 0.99 and r["parity"]["modality_weight_max_abs"] < 1e-4


def test_transposition_takes_the_slab_pointers_tfidf_searched(hip):
    """tfidf_device leaves the 8192-column slab pointers with its result; the transposition of that matrix reads them
    instead of searching again (csrc/tpack4.hip mu_tpack4_count d_slab_ptr) - same output, and a wrapper of the
    backend (bench.py's TimedBackend) must not lose them"""
    import torch

    from muon_amd._atac.preproc import tfidf_device
    from muon_amd._backend import DeviceCSR

    class Wrapped:  # forwards like bench.TimedBackend
        def __init__(self, be):
            self._be = be

        def __getattr__(self, name):
            return getattr(self._be, name)

    X = hip.synth_counts(0, 3000, 20000, 20, 0.03, 0)
    for be in (hip, Wrapped(hip)):
        T = tfidf_device(be, X, 3000, 3, 1e4)
        sp_, key = T.slab_ptr
        assert sp_.numel() == 3000 * (3 + 1) and key == (T.indptr.data_ptr(), T.indices.data_ptr(), 3000, 20000)
        assert hip._slab_ptr_of(T) is sp_
        plain = DeviceCSR(T.indptr, T.indices, T.values, T.shape)
        assert hip._slab_ptr_of(plain) is None
        a, b = hip.transpose_csr(T), hip.transpose_csr(plain)
        assert torch.equal(a.indptr, b.indptr) and torch.equal(a.indices, b.indices) and torch.equal(a.values, b.values)
        sa, sb = hip.transpose_stream(T), hip.transpose_stream(plain)
        assert torch.equal(sa.sptr, sb.sptr) and torch.equal(sa.ent, sb.ent)
    # pointers of other index arrays are not taken
    other = DeviceCSR(T.indptr.clone(), T.indices.clone(), T.values, T.shape)
    other.slab_ptr = T.slab_ptr
    assert hip._slab_ptr_of(other) is None


def test_warm_start_saves_an_expansion_and_gives_the_same_subspace(hip, monkeypatch):
    """r05: the start block takes power steps on a slice of the cells before the first full product
    (`_lsi_device`, MUON_AMD_LSI_WARM).  Forced on a matrix below its size threshold: fewer expansions than the cold
    run, both within the parity bar of the f64 oracle, singular values alike."""
    from muon_amd._atac.preproc import tfidf_device
    from muon_amd._atac.tools import lsi_device
    from oracle import lsi_oracle, tfidf_oracle
    from tests.synth import planted_topics_csr

    X = planted_topics_csr(40000, 12000, n_topics=20, density=0.03, seed=4, dtype=np.float32)
    Xd = hip.upload_csr(X.indptr, X.indices, X.data, X.shape)
    T = tfidf_device(hip, Xd, X.shape[0], 3, 1e4)
    monkeypatch.setenv("MUON_AMD_LSI_WARM", "0")
    Uc, sdc, Vc, ic = lsi_device(hip, T, n_comps=20, n_obs=X.shape[0], return_info=True)
    monkeypatch.setenv("MUON_AMD_LSI_WARM", "8:2")
    Uw, sdw, Vw, iw = lsi_device(hip, T, n_comps=20, n_obs=X.shape[0], return_info=True)
    assert ic["warm_start"] is None and iw["warm_start"]["power_steps"] == 2 and 4000 < iw["warm_start"]["cells"] < 12000
    assert iw["converged"] and ic["converged"] and iw["iterations"] < ic["iterations"], (ic["bounds"], iw["bounds"])
    ref = lsi_oracle.lsi(tfidf_oracle.canonical(tfidf_oracle.tfidf(X)), n_comps=20)
    for V, sd in ((Vc, sdc), (Vw, sdw)):
        assert lsi_oracle.max_subspace_angle(hip.to_host(V), ref["LSI"]) < 1e-4
        assert np.max(np.abs(sd - ref["stdev"]) / ref["stdev"]) < 1e-5


@pytest.mark.parametrize("n,d,n_s,dens", [(40000, 12000, 5000, 0.03), (6000, 30000, 1500, 0.02), (3000, 700, 640, 0.1),
                                          (70000, 9000, 2200, 0.01)])
def test_slice_products_without_operands_of_their_own(hip, n, d, n_s, dens):
    """r06: the warm start's products on a cell slice.  X_S Q on the compact slice stream with the column super-slabs
    split over blockIdx.y (mu_csr_slice_stream, mu_spmm_stream_ranges_f32), X_S^T Y_S on the row stream of X^T as it
    is, the slice's cells found through the transposition's count prefixes - against scipy on the same rows."""
    import scipy.sparse as sp

    rng = np.random.default_rng(n + d)
    m = sp.random(n, d, density=dens, format="csr", random_state=rng, dtype=np.float32)
    m.data = (0.5 + rng.random(m.nnz)).astype(np.float32)
    m.sort_indices()
    if n == 3000:  # rows with hundreds of entries in one 256-column slab, and empty rows
        m = m.tolil(); m[5, :300] = 1.5; m[7, :] = 0; m = m.tocsr(); m.sort_indices(); m.eliminate_zeros()
    X = hip.upload_csr(m.indptr, m.indices, m.data, m.shape)
    Xs, Xt = hip.stream_both(X)
    assert Xt.t4 is not None
    plan = hip.slice_plan(X, n_s)
    assert plan is not None and hip.slice_plan(X, n_s) is plan  # (cached with the index arrays)
    rows = np.concatenate([np.arange(r["row0"], r["row1"]) for r in plan["ranges"]])
    assert len(rows) == plan["n_s"] and len(set(rows)) == len(rows) and abs(len(rows) - n_s) <= max(plan["rpb"], n_s // 2)
    ms = m[rows].astype(np.float64)
    assert plan["nnz_s"] == ms.nnz
    S = hip.slice_stream(X, plan)
    Q = rng.standard_normal((d, 64)).astype(np.float32)
    Ys = hip.spmm_slice(S, hip.to_device(Q))
    want = ms @ Q.astype(np.float64)
    got = hip.to_host(Ys)
    assert np.max(np.abs(got - want)) <= 2e-5 * np.max(np.abs(want))
    Z = hip.to_host(hip.spmm_slice_t(Xt, plan, Ys))
    wantz = ms.T @ got.astype(np.float64)
    assert np.max(np.abs(Z - wantz)) <= 2e-5 * np.max(np.abs(wantz))
    # the same bytes run to run (fixed-order sums)
    assert np.array_equal(hip.to_host(hip.spmm_slice(S, hip.to_device(Q))), got)


def test_warm_start_on_ranges_matches_the_slice_operands(hip, monkeypatch):
    """The r06 slice (ranges of row blocks, no operands) and the r05 slice (copy + transposition of the slice) start
    the same iteration: same number of expansions, both inside the parity bar of the f64 oracle."""
    from muon_amd._atac.preproc import tfidf_device
    from muon_amd._atac.tools import lsi_device
    from oracle import lsi_oracle, tfidf_oracle
    from tests.synth import planted_topics_csr

    X = planted_topics_csr(40000, 12000, n_topics=20, density=0.03, seed=4, dtype=np.float32)
    Xd = hip.upload_csr(X.indptr, X.indices, X.data, X.shape)
    T = tfidf_device(hip, Xd, X.shape[0], 3, 1e4)
    ref = lsi_oracle.lsi(tfidf_oracle.canonical(tfidf_oracle.tfidf(X)), n_comps=20)
    monkeypatch.setenv("MUON_AMD_LSI_WARM", "8:2")
    out = {}
    for how in ("ranges", "operands"):
        monkeypatch.setenv("MUON_AMD_LSI_WARM_SLICE", how)
        U, sd, V, info = lsi_device(hip, T, n_comps=20, n_obs=X.shape[0], return_info=True)
        assert info["warm_start"]["slice"] == how and info["converged"]
        assert lsi_oracle.max_subspace_angle(hip.to_host(V), ref["LSI"]) < 1e-4
        assert np.max(np.abs(sd - ref["stdev"]) / ref["stdev"]) < 1e-5
        out[how] = info
    assert out["ranges"]["iterations"] == out["operands"]["iterations"]


def test_f64_input_is_answered_in_f64_arithmetic(hip):
    """the reference runs f64 ARPACK when X is f64 (/root/reference/muon/_atac/tools.py:53).  r05
    documented the f32 floor (~1e-5 rad on this case); r06 continues the f32 Krylov process in f64 for f64 input
    (tools._refine_f64: the row-stream SpMM's f64 blocks - f32 streams for the gathers, products accumulated in f64 - f64
    bases, exact residuals, stopped by a Davis-Kahan bound of 1e-6).  The hardest gapped case of the suite - 80 planted
    topics, f64 TF-IDF values - at n_comps = 50 (1.3 % gap) and at n_comps = 26 (0.16 % gap): below 1e-6 rad of f64
    ARPACK, singular values to 1e-8 (the stream's VALUES stay f32: 8e-10), outputs f64 like the reference's."""
    from muon_amd import AnnData
    from muon_amd import atac as ac
    from muon_amd._atac.tools import lsi_device
    from oracle import tfidf_oracle

    X = planted_topics_csr(3000, 2500, n_topics=80, density=0.03, seed=3, dtype=np.float64)
    T = tfidf_oracle.canonical(tfidf_oracle.tfidf(X))
    assert T.dtype == np.float64
    for k in (50, 26):
        ref = lsi_oracle.lsi(T, n_comps=k)
        ad = AnnData(T.copy())
        ac.tl.lsi(ad, n_comps=k)
        assert ad.varm["LSI"].dtype == np.float64 and ad.obsm["X_lsi"].dtype == np.float64
        ang = lsi_oracle.max_subspace_angle(ad.varm["LSI"], ref["LSI"])
        print(f"f64 input, n_comps = {k}: angle to f64 ARPACK {ang:.2e}")
        assert ang < 1e-6
        np.testing.assert_allclose(ad.uns["lsi"]["stdev"], ref["stdev"], rtol=1e-8)
        np.testing.assert_allclose(ad.obsm["X_lsi"].mean(axis=0), 0, atol=1e-11)
        np.testing.assert_allclose(ad.obsm["X_lsi"].std(axis=0), 1, rtol=1e-9)
    # more components than the block width (the continuation starts from every Ritz vector the f32 run kept) and a
    # Krylov space that reaches the matrix dimension (deflation): tests/test_host_logic.py has the anatomy
    X2 = planted_topics_csr(1500, 900, n_topics=30, density=0.05, seed=3, dtype=np.float64)
    T2 = tfidf_oracle.canonical(tfidf_oracle.tfidf(X2))
    for k in (63, 70):
        ref2 = lsi_oracle.lsi(T2, n_comps=k)
        ad = AnnData(T2.copy())
        ac.tl.lsi(ad, n_comps=k)
        assert ad.varm["LSI"].shape == (900, k) and ad.varm["LSI"].dtype == np.float64
        assert lsi_oracle.max_subspace_angle(ad.varm["LSI"], ref2["LSI"]) < 1e-6
        np.testing.assert_allclose(ad.uns["lsi"]["stdev"], ref2["stdev"], rtol=1e-8)
    # the same operand with and without the continuation: what it costs and what it buys
    Xd = hip.upload_csr(T.indptr, T.indices, T.data, T.shape, values_dtype=np.float32)
    ref = lsi_oracle.lsi(T, n_comps=50)
    _, _, V32, i32 = lsi_device(hip, Xd, n_comps=50, return_info=True)
    _, _, V64, i64 = lsi_device(hip, Xd, n_comps=50, return_info=True, refine_f64=True)
    r = i64["refine_f64"]
    a32 = lsi_oracle.max_subspace_angle(hip.to_host(V32), ref["LSI"])
    a64 = lsi_oracle.max_subspace_angle(hip.to_host(V64), ref["LSI"])
    print(f"f32 process {a32:.2e} rad after {i32['spmm']} products; f64 continuation {a64:.2e} rad, {r['blocks']} blocks "
          f"({r['products']} wide products), bound {r['angle_bound']:.1e}")
    assert V64.dtype == torch.float64 and a64 < 1e-7 < a32 < 1e-4 and r["angle_bound"] <= 1e-6 and r["blocks"] <= 8
