#!/usr/bin/env python
"""The HOST side of the C-ABI under AddressSanitizer + UndefinedBehaviorSanitizer (sanitizers; VERDICT r05 weak
15).  GPU AddressSanitizer is not available on this pool, so the device code is compiled as usual (-fno-gpu-sanitize) and
the entry points are driven WITHOUT a GPU: what runs is everything a call does before its first launch - argument
validation, geometry / work-size arithmetic, the by-value range descriptors of the ranged SpMM, the tune table, the error
string - plus the entry points that are pure host code (mu_host_hash64 and its threads).

    python scripts/sanitize_host.py            # builds muon_amd/csrc/build/san/libmuon_amd_san.so (cached) and drives it
    python scripts/sanitize_host.py --drive SO # (internal) the driver, run under LD_PRELOAD=libclang_rt.asan

Exit status 0 = no sanitizer report.  tests/test_sanitizers.py runs it.
"""
import ctypes as C
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "muon_amd", "csrc")
SAN = os.path.join(CSRC, "build", "san")
FLAGS = ["-O1", "-g", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
         "-fsanitize=address,undefined", "-fno-gpu-sanitize", "-fno-sanitize-recover=undefined", "-w"]


def build():
    sys.path.insert(0, ROOT)
    from muon_amd.csrc.build import EXTRA, SOURCES, _hipcc

    hipcc = _hipcc()
    os.makedirs(SAN, exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    hdrs = [os.path.join(CSRC, "common.hpp"), os.path.join(CSRC, "sweep.hpp"), os.path.join(ROOT, "include", "muon_amd.h")]
    h = hashlib.sha256()
    for p in srcs + hdrs:
        h.update(open(p, "rb").read())
    h.update(" ".join(FLAGS[:4] + FLAGS[7:]).encode())
    lib, stamp = os.path.join(SAN, "libmuon_amd_san.so"), os.path.join(SAN, ".stamp")
    if os.path.exists(lib) and os.path.exists(stamp) and open(stamp).read() == h.hexdigest():
        return lib, hipcc
    procs, objs = [], []
    for s in srcs:
        o = os.path.join(SAN, os.path.basename(s) + ".o")
        objs.append(o)
        procs.append((s, subprocess.Popen([hipcc] + FLAGS + EXTRA.get(os.path.basename(s), []) + ["-c", s, "-o", o])))
    for s, p in procs:
        if p.wait() != 0:
            raise SystemExit(f"hipcc (sanitizer build) failed on {s}")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fsanitize=address,undefined",
                           "-fno-gpu-sanitize", "-o", lib] + objs + ["-Wl,-rpath,/opt/rocm/lib"])
    open(stamp, "w").write(h.hexdigest())
    return lib, hipcc


def drive(path):
    """Host-only calls.  No GPU is needed (or used): a call that would launch is stopped by its own argument checks."""
    lib = C.CDLL(path)
    lib.mu_last_error.restype = C.c_char_p
    i64, sz = C.c_int64, C.c_size_t
    assert lib.mu_version() >= 600
    # the tune table: known key, unknown key, NULL key, negative value
    assert lib.mu_tune_set(b"tpack4_c", 64) == 0 and lib.mu_tune_get(b"tpack4_c") == 64 and lib.mu_tune_set(b"tpack4_c", 0) == 0
    assert lib.mu_tune_set(b"no_such_key_with_a_rather_long_name_" * 8, 1) != 0 and b"unknown key" in lib.mu_last_error()
    assert lib.mu_tune_set(None, 1) != 0 and lib.mu_tune_set(b"spmm_k", -1) != 0 and lib.mu_tune_get(None) == -1
    # host hash: 0 bytes, 1 byte, unaligned start, many chunks on many threads, NULL
    out = C.c_uint64(0)
    buf = (C.c_ubyte * ((48 << 20) + 7))(*([0] * 16))
    for off, n, thr in ((0, 0, 1), (0, 1, 4), (3, 5, 2), (1, (48 << 20) + 5, 16), (0, 16 << 20, 64), (0, (16 << 20) + 1, 3)):
        assert lib.mu_host_hash64(C.byref(buf, off), sz(n), thr, C.c_uint64(7), C.byref(out)) == 0
    a = C.c_uint64(0)
    lib.mu_host_hash64(C.byref(buf, 1), sz(1 << 20), 1, C.c_uint64(7), C.byref(a))
    lib.mu_host_hash64(C.byref(buf, 1), sz(1 << 20), 9, C.c_uint64(7), C.byref(out))
    assert a.value == out.value  # the digest does not depend on the thread count
    assert lib.mu_host_hash64(None, sz(8), 2, C.c_uint64(1), C.byref(out)) != 0 and lib.mu_last_error()
    # geometry / work sizes / offsets of the transposition over a spread of shapes (pure arithmetic)
    for f in ("mu_tpack4_worksize", "mu_tpack4_err_offset", "mu_tpack4_cnt_offset", "mu_csr_row_col_sums_worksize",
              "mu_csr_transpose_worksize", "mu_gram_worksize"):
        getattr(lib, f).restype = sz
    for n, d, nnz in ((1, 1, 1), (7, 5, 17), (125000, 200000, 783805300), (1000000, 200000, 6276529450),
                      ((1 << 31) + 5, 10, 1 << 20), (600, (1 << 31) - 1, 1 << 30), (512, 1100000, 1 << 30)):
        ok = lib.mu_tpack4_supported(i64(n), i64(d), i64(nnz))
        rpb, G, Ct = i64(0), C.c_int(0), C.c_int(0)
        assert lib.mu_tpack4_geometry(i64(n), i64(d), i64(nnz), C.byref(rpb), C.byref(G), C.byref(Ct)) == 0
        assert lib.mu_tpack4_geometry(i64(n), i64(d), i64(nnz), None, None, None) == 0
        if ok:
            w = lib.mu_tpack4_worksize(i64(n), i64(d), i64(nnz))
            e, c = lib.mu_tpack4_err_offset(i64(n), i64(d), i64(nnz)), lib.mu_tpack4_cnt_offset(i64(n), i64(d), i64(nnz))
            assert c < e < w and e % 4 == 0 and e + 4 <= w and c + 4 * (G.value + 1) * d <= e
        lib.mu_csr_row_col_sums_worksize(i64(n), i64(d))
        lib.mu_csr_transpose_worksize(i64(n), i64(d), i64(nnz))
    for n in (0, 1, 100, 16384, 125000, 200000, 1000000, 1 << 33):
        assert 1 <= lib.mu_spmm_stream_k(i64(n)) <= 8
    # argument checks of the entry points that launch: every one returns an error BEFORE touching a device
    one = (C.c_byte * 64)()
    P = C.byref(one)
    bad = [
        lib.mu_spmm_stream_f32(i64(10), i64(10), None, None, None, 1, None, 64, None, None),           # null operands
        lib.mu_spmm_stream_f32(i64(10), i64(10), P, P, None, 1, P, 48, P, None),                        # B not 16 / 32 / 64
        lib.mu_spmm_stream_f32(i64(10), i64(1 << 31), P, P, None, 1, P, 64, P, None),                   # too many columns
        lib.mu_spmm_stream_f64(i64(10), i64(10), P, P, None, 1, P, 64, P, 0, None),                     # f64: B = 64
        lib.mu_spmm_stream_ranges_f32(i64(10), P, P, None, 4, P, i64(10), P, i64(0), P, i64(10), 33, P, 1, None),   # > 32 ranges
        lib.mu_spmm_stream_ranges_f32(i64(10), P, P, None, 0, P, i64(10), P, i64(0), P, i64(10), 1, P, 1, None),    # no layout K
        lib.mu_spmm_stream_ranges_f32(i64(10), P, P, None, 4, P, i64(10), P, i64(0), P, i64(10), 1, None, 1, None), # no ranges
        lib.mu_csr_slice_stream(0, P, P, P, P, i64(10), P, P, P, P, P, P, P, None),                      # 0 ranges
        lib.mu_csr_slice_stream(40, P, P, P, P, i64(10), P, P, P, P, P, P, P, None),                     # 40 ranges
        lib.mu_tpack4_count(i64(0), i64(10), i64(5), P, P, P, P, sz(0), None, None),                     # unsupported shape
        lib.mu_tpack4_fill_stream(i64(100), i64(10), i64(50), P, None, None, None, None, P, None, P, P, sz(1), None),  # no source
        lib.mu_mofa_update_w(0, i64(10), 40, 1, P, P, P, P, P, P, P, 1, P, P, P, P, P, None),           # K > 32
        lib.mu_mofa_update_w(7, i64(10), 4, 1, P, P, P, P, P, P, P, 1, P, P, P, P, P, None),            # bad dtype
    ]
    assert all(rc != 0 for rc in bad), bad
    # one descending range: h[1] < h[0]
    h = (C.c_int32 * 5)(512, 256, 0, 0, 1)
    assert lib.mu_spmm_stream_ranges_f32(i64(10), P, P, None, 4, P, i64(10), P, i64(0), P, i64(10), 1, h, 1, None) != 0
    # slice ranges with a negative extent
    r0, rows = (C.c_int64 * 1)(0), (C.c_int64 * 1)(4)
    lo, hi = (C.c_int64 * 1)(10), (C.c_int64 * 1)(3)
    assert lib.mu_csr_slice_stream(1, r0, rows, lo, hi, i64(10), P, P, P, P, P, P, P, None) != 0
    msg = lib.mu_last_error()
    assert msg and len(msg) < 512
    print("sanitizer drive ok")


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--drive":
        drive(sys.argv[2])
        sys.exit(0)
    lib, hipcc = build()
    clang = os.path.join(os.path.dirname(os.path.realpath(hipcc)), "..", "lib", "llvm", "bin", "clang")
    if not os.path.exists(clang):
        clang = "/opt/rocm/lib/llvm/bin/clang"
    rt = subprocess.run([clang, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=23",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1:exitcode=24")
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--drive", lib], env=env, capture_output=True, text=True)
    sys.stdout.write(r.stdout)
    sys.stderr.write(r.stderr[-4000:])
    sys.exit(r.returncode)
