"""First-contact probe of the matrix-core SpMM (r04): hardware semantics, cutter, product, timings.
Prints what it finds instead of stopping at the first surprise."""
import sys, os, time, functools
print = functools.partial(print, flush=True)
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from muon_amd._backend import get_backend
from muon_amd._ffi import check
from tests import cells_ref as cr
from tests.synth import planted_topics_csr

be = get_backend()
p = lambda t: t.data_ptr()

# 1. transpose read: every lane reads a distinct (row, piece); report where each output element came from
rows = 256
img = (np.arange(rows, dtype=np.uint16)[:, None] * 256 + np.arange(16, dtype=np.uint16)[None, :])
addr = np.zeros(64, dtype=np.uint32)
for l in range(64):
    addr[l] = (l * 3 + 1) * 32 + (l & 3) * 8  # row 3 l + 1, piece l & 3
d_out = be.zeros((128,), torch.int32)
d_img = be.to_device(img.reshape(-1).view(np.uint32)); d_addr = be.to_device(addr)
check(be.lib.mu_probe_tr16(p(d_img), d_img.numel(), p(d_addr), p(d_out), None)); torch.cuda.synchronize()
got = be.to_host(d_out).view(np.uint16).reshape(64, 4)
print("tr16: out[lane][e] = (source lane, element of its 4) :")
for l in list(range(0, 20)) + [32, 33, 48]:
    src = []
    for e in range(4):
        r, c = int(got[l, e]) >> 8, int(got[l, e]) & 255
        sl = (r - 1) // 3
        src.append((sl, c - 4 * (sl & 3)) if (r - 1) % 3 == 0 and 0 <= c - 4 * (sl & 3) < 4 else ("?", r, c))
    print("  lane", l, src)
ok0 = all(got[l, e] == ((3 * (16 * (l >> 4) + 4 * e + ((l & 15) >> 2)) + 1) * 256 + (l & 15) % 4 + 4 * ((l & 15) >> 2) * 0 + 0) or True for l in range(64) for e in range(4))

# 2..: the tests' checks, verbosely
def run(name, fn):
    t0 = time.time()
    try:
        fn()
        print(f"[ok] {name} ({time.time()-t0:.1f}s)", flush=True)
    except Exception as ex:  # noqa
        import traceback
        print(f"[FAIL] {name}: {type(ex).__name__}: {str(ex)[:600]}", flush=True)
        traceback.print_exc(limit=2)

import tests.test_gpu_mfma as T
run("tr16 semantics", lambda: T.test_transpose_read_gathers_four_rows_per_lane_group(be))
run("mfma layout", lambda: T.test_mfma_16x16x32_f16_fragment_layout(be))
for ns, sh in [(1, (300, 1700)), (1, (37, 513)), (2, (130, 900)), (1, (1000, 2048))]:
    run(f"cut nset={ns} {sh}", lambda: T.test_cells_cut_holds_every_entry_once(be, ns, sh))

# product on HOST-built cells first (separates the kernel from the cutter)
def host_cells_product(shape=(300, 1700), nset=1):
    from muon_amd._backend import DeviceCells
    m = planted_topics_csr(shape[0], shape[1], n_topics=7, density=0.04, seed=shape[1])
    rng = np.random.default_rng(2)
    Q = (rng.standard_normal((shape[1], 64)) * rng.uniform(1e-3, 1e2, 64)).astype(np.float32)
    hdr, base, cells, vs = cr.encode(m, nset)
    g = cr.geometry(nset)
    Xc = DeviceCells(be.to_device(hdr), be.to_device(base), be.to_device(cells), m.shape, m.nnz, nset,
                     be.to_device(np.array([vs], dtype=np.float32)), g["slab_rows"], g["stride"], {})
    for trmap in (0,):
        Qd = be.to_device(Q)
        Y = be.to_host(be.spmm(Xc, Qd))
        Yref, Qr = cr.product(m, Q, nset=nset, vscale=vs)
        scale = np.abs(m).astype(np.float64) @ np.abs(Qr).astype(np.float64) + 1e-30
        err = np.max(np.abs(Y - Yref) / scale)
        print(f"   host cells {shape} nset={nset} trmap={trmap}: max scaled err {err:.3e}  rounded block ok {np.array_equal(be.to_host(Qd), Qr if nset == 1 else Q)}  nan {np.isnan(Y).sum()}")
        if err > 1e-5 and trmap == 0:
            bad = np.argwhere(np.abs(Y - Yref) / scale > 1e-5)
            print("   first bad (row, col):", bad[:8].tolist(), "rows hit:", np.unique(bad[:, 0])[:16].tolist(), "cols hit:", np.unique(bad[:, 1])[:16].tolist())
            print("   Y[0,:6]", Y[0, :6], "ref", Yref[0, :6])
run("product on host-built cells", host_cells_product)
run("product on host-built cells (3 slabs, 2 wgs)", lambda: host_cells_product((700, 1300)))
run("product on host-built cells nset 2", lambda: host_cells_product((130, 900), 2))
for sh in [(300, 1700), (37, 513), (1000, 2048), (5000, 3000)]:
    run(f"product {sh}", lambda: T.test_product_equals_f64_arithmetic_on_the_rounded_operands(be, sh))
run("product nset 2", lambda: T.test_product_with_a_two_term_operand(be))

# timings at one shard of configs[2]
def timing(n=125_000, d=200_000):
    X = be.synth_counts(0, n, d, 50, 0.03, 7)
    X = X.with_values(X.values.to(torch.float32))
    torch.cuda.synchronize()
    ev = lambda: torch.cuda.Event(enable_timing=True)
    def tm(fn, reps=3):
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            a, b = ev(), ev(); a.record(); r = fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
        return min(ts), r
    t_cut, Xc = tm(lambda: be.cells(X))
    be._cells_check(Xc)
    steps = int(Xc.hdr.sum().item())
    print(f"   {n} x {d}: nnz {X.nnz}  cut {t_cut:.2f} ms  steps {steps}  slot use {X.nnz / (32 * steps):.3f}  stream {steps * 224 / 1e9:.2f} GB (alloc {Xc.cells.numel() / 1e9:.2f})")
    Q = be.randn(d, 64, 3)
    t, Y = tm(lambda: be.spmm(Xc, Q), 5)
    print(f"   X Q  matrix cores: {t:.3f} ms  = {X.nnz / t / 1e6:.1f} G entries/s")
    for mode, what in ((1, "no MFMA"), (4, "no masks"), (16, "no slab copies"), (32, "no barrier"), (53, "no copies, no barrier, no MFMA, no masks"), (64, "conflict-free gathers"), (117, "53 + conflict-free gathers"), (128, "no stream"), (181, "53 + no stream"), (245, "53 + conflict-free + no stream")):
        be.tune("mfma_mode", mode)
        tt, _ = tm(lambda: be.spmm(Xc, Q), 3)
        print(f"      mode {mode} ({what}): {tt:.3f} ms")
    for mode in (8,):
        be.tune("mfma_mode", mode)
        Yt = be.spmm(Xc, Q)
        torch.cuda.synchronize()
        acc = Yt[::32, :5].double()  # per band: total, take, multiply, slab, steps
        tot, take, mul, slab, steps = [acc[:, i] for i in range(5)]
        print(f"      accounting mode {mode} (cycles per step and wave; {int(steps.sum())} steps): total {float(tot.sum() / steps.sum()):.0f}"
              f"  steps {float(mul.sum() / steps.sum()):.0f}"
              f"  slab transitions {float(slab.sum() / steps.sum()):.0f}   (wave total: mean {float(tot.mean()):.0f} max {float(tot.max()):.0f})")
    be.tune("mfma_mode", 0)
    # the two-term operand (nset 2) on the same matrix
    Xc2 = be.cells(X, nset=2)
    be._cells_check(Xc2)
    st2 = int(Xc2.hdr.sum().item())
    t22, Y22 = tm(lambda: be.spmm(Xc2, Q), 5)
    print(f"   X Q  matrix cores, two-term operand: {t22:.3f} ms  steps {st2} slot use {X.nnz / (32 * st2):.3f}")
    be.tune("mfma_mode", 8)
    Yt = be.spmm(Xc2, Q); torch.cuda.synchronize()
    acc = Yt[::32, :5].double()
    tot, take, mul, slab, steps = [acc[:, i] for i in range(5)]
    print(f"      accounting nset 2: total {float(tot.sum() / steps.sum()):.0f}  steps {float(mul.sum() / steps.sum()):.0f}  slab {float(slab.sum() / steps.sum()):.0f}")
    be.tune("mfma_mode", 0)
    del Xc2
    for cfg in ():
        be.tune("mfma_cfg", cfg)
        try:
            tt, Yc = tm(lambda: be.spmm(Xc, Q), 3)
            print(f"      cfg {cfg} (waves {cfg // 100}, ring {cfg % 100}): {tt:.3f} ms  same result {bool(torch.equal(Yc, Y))}")
        except Exception as ex:  # noqa
            print(f"      cfg {cfg}: {str(ex)[:200]}")
    Xs = be.stream(X)
    t2, Yw = tm(lambda: be.spmm(Xs, Q), 5)
    print(f"   X Q  row stream f32: {t2:.3f} ms;  max |diff| / max |Y| = {(Y - Yw).abs().max().item() / Yw.abs().max().item():.2e}")
    t3, _ = tm(lambda: be.dense16(Xc, Q, True), 5)
    print(f"   dense16: {t3:.3f} ms")
    return X
run("timing 125k x 200k", timing)
if len(sys.argv) > 1 and sys.argv[1] == "full":
    run("timing 1M x 200k", lambda: timing(1_000_000, 200_000))
