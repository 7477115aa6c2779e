"""VERDICT r03 item 1 b: the LSI product X Q (B = 64, f32) on a PRE-TILED operand (csrc/spmm_ell64.hip) against the
row-stream kernel, 125 000 x 200 000.  The layout is built here with tensor operations ("offline by any means").
usage: python scripts/probes/ell64_probe.py [cells] [peaks]
ARCHIVED with its kernel (scripts/probes/spmm_ell64.hip is not compiled into the library: the experiment was a no-go,
DESIGN.md 4.4, profiles/r04_ell64_probe.txt).  To run it again: copy the .hip into muon_amd/csrc/, add it to
csrc/build.py SOURCES with EXTRA ["-Wno-inline-asm", "-std=c++20"], declare
    "mu_spmm_ell64_f32": (C.c_int, [_i32, _i32, _i64, _i64] + [_vp] * 8)
in muon_amd/_ffi.py SIGNATURES (and the prototype at the end of the .hip in include/muon_amd.h), rebuild."""
import sys
import time

import torch

import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from muon_amd._backend import _p, get_backend  # noqa: E402
from muon_amd._ffi import check  # noqa: E402


def build(X, K, n_waves=None):
    n, d = X.shape
    dev = X.indices.device
    S = -(-d // 256)
    lens = X.indptr[1:] - X.indptr[:-1]
    order = torch.argsort(lens, descending=True, stable=True)
    n_sets = -(-n // 4)
    n_waves = max(-(-n_sets // K), int(n_waves or 0))  # (more waves than needed: the last row-set slots stay empty)
    n_pos = n_waves * K * 4
    i = torch.arange(n, device=dev)
    q, g = i // 4, i % 4
    pos_sorted = ((q % n_waves) * K + q // n_waves) * 4 + g
    perm = torch.full((n_pos,), -1, dtype=torch.int32, device=dev)
    perm[pos_sorted] = order.to(torch.int32)
    pos_of_row = torch.empty((n,), dtype=torch.int64, device=dev)
    pos_of_row[order] = pos_sorted
    rows = torch.repeat_interleave(torch.arange(n, device=dev), lens)
    pos = pos_of_row[rows]
    del rows
    sl = (X.indices >> 8).to(torch.int64)
    cnt = torch.bincount(pos * S + sl, minlength=n_pos * S).view(n_pos, S)
    steps = cnt.view(n_waves, K, 4, S).amax(dim=2).permute(0, 2, 1).contiguous()  # [wave, slab, k]
    assert int(steps.max()) <= 255
    hdr = torch.zeros((n_waves, S, 16), dtype=torch.uint8, device=dev)  # 16 count bytes per (wave, slab)
    hdr[:, :, :K] = steps.to(torch.uint8)
    n_ovf_w = torch.clamp((steps + 15) // 16 - 1, min=0)
    ovf_start = torch.zeros(n_ovf_w.numel() + 1, dtype=torch.int64, device=dev)
    torch.cumsum(n_ovf_w.reshape(-1), 0, out=ovf_start[1:])
    n_ovf = int(ovf_start[-1])
    ovf_start = ovf_start[:-1].view(n_waves, S, K)
    ovf_base = ovf_start[:, 0, 0].contiguous()
    # rank of an entry inside its (row, slab)
    start = torch.cumsum(cnt, dim=1) - cnt
    e = torch.arange(X.nnz, device=dev)
    row_start = torch.repeat_interleave(X.indptr[:-1], lens)
    t = e - row_start - start[pos, sl]
    del e, row_start, start, cnt
    gw, kk, gg = pos // (4 * K), (pos // 4) % K, pos % 4
    n_reg = n_waves * S * K
    win = torch.where(t < 16, (gw * S + sl) * K + kk, n_reg + K + ovf_start[gw, sl, kk] + torch.clamp(t // 16 - 1, min=0))
    dest = win * 64 + 16 * gg + t % 16
    total = n_reg + K + n_ovf + 1
    vals = torch.zeros((total * 64,), dtype=torch.float32, device=dev)
    offs = torch.zeros((total * 64,), dtype=torch.int16, device=dev)
    vals[dest] = X.values
    off = (X.indices.to(torch.int32) & 255) << 8
    offs[dest] = torch.where(off >= 32768, off - 65536, off).to(torch.int16)
    del dest, win, t, off
    ent = torch.cat([vals.view(torch.uint8).view(-1, 256), offs.view(torch.uint8).view(-1, 128)], dim=1).contiguous()
    used = float((steps.sum() * 4).item())
    return dict(hdr=hdr, ent=ent, ovf_off=(n_reg + K) * 384, ovf_base=ovf_base, perm=perm, n_pos=n_pos, K=K, n_waves=n_waves,
                n_ovf=n_ovf, slot_use=X.nnz / max(used, 1.0), step_mean=float(steps.float().mean()))


def product(be, L, Q, n, d, waves):
    out = torch.empty((n, 64), dtype=torch.float32, device=Q.device)
    ent_ptr = L["ent"].data_ptr()
    with be._dev_ctx():
        check(be.lib.mu_spmm_ell64_f32(L["K"], waves, L["n_pos"], d, _p(L["hdr"]), ent_ptr, _p(L["ovf_base"]),
                                       ent_ptr + L["ovf_off"], _p(L["perm"]), _p(Q), _p(out), be._stream()))
    return out


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
    be = get_backend()
    X = be.synth_counts(0, n, d, 50, 0.03, 0)
    X = type(X)(X.indptr, X.indices, torch.log1p(X.values.to(torch.float32)) + 0.25, X.shape)
    print(f"X {X.shape} nnz {X.nnz}", flush=True)
    Q = torch.randn(d, 64, device=be.device, dtype=torch.float32)
    S = be.stream(X)
    ref = be.spmm(S, Q)
    t_ref = timed(lambda: be.spmm(S, Q))
    alg = 8 * X.nnz + 8 * (n + 1) + 4 * 64 * (n + d)
    print(f"row stream k_spmm_win: {t_ref:.3f} ms ({alg / t_ref / 1e6:.0f} GB/s algorithmic)", flush=True)
    del S
    n_sets = -(-n // 4)
    cus = torch.cuda.get_device_properties(be.device).multi_processor_count
    configs = []
    for K in (10, 9, 8, 6):
        for waves in (15, 14, 12, 10, 8):
            need = -(-n_sets // K)
            wgs = -(-need // waves)
            rounds = -(-wgs // cus)
            fill = n_sets / (rounds * cus * waves * K)
            configs.append((fill, K, waves, rounds * cus * waves))
    configs.sort(reverse=True)
    for fill, K, waves_c, nw in configs[:6]:
        t0 = time.perf_counter()
        L = build(X, K, nw)
        torch.cuda.synchronize()
        tb = time.perf_counter() - t0
        line = (f"K={K}: build {tb:.1f}s  windows {L['ent'].shape[0]} ({L['ent'].numel() / 1e9:.2f} GB, {L['ent'].numel() / alg:.2f}x algorithmic) "
                f"overflow windows {L['n_ovf']}  mean steps {L['step_mean']:.2f}  slot use {L['slot_use']:.3f}")
        line += f"  [fill {fill:.2f}]"
        for waves in (waves_c,):
            out = product(be, L, Q, n, d, waves)
            err = float((out - ref).abs().max() / ref.abs().max())
            same = bool(torch.equal(out, ref))
            ts = []
            for mode in (0, 1, 2, 3, 4, 7):
                be.tune("ell_mode", mode)
                ts.append(timed(lambda: product(be, L, Q, n, d, waves)))
            be.tune("ell_mode", 0)
            wgs = -(-L["n_waves"] // waves)
            line += (f"\n   waves {waves} ({wgs} wgs): {ts[0]:.3f} ms ({alg / ts[0] / 1e6:.0f} GB/s algorithmic; err {err:.1e}, bit-identical to "
                     f"k_spmm_win {same}) | no gathers {ts[1]:.3f} | no slab copies {ts[2]:.3f} | neither {ts[3]:.3f} | overflow ignored {ts[4]:.3f} | "
                     f"all three {ts[5]:.3f}")
        print(line, flush=True)
        del L


if __name__ == "__main__":
    main()
