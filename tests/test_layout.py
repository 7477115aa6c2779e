"""Repository contracts that need no GPU: the C-ABI library loads and exports every symbol
include/muon_amd.h declares, the ctypes table covers them, and the product package never
touches the oracle (which is test infrastructure)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "muon_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mu_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from muon_amd import _ffi

    names = _declared()
    assert len(names) >= 30
    lib = ctypes.CDLL(_ffi.LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in muon_amd.h but not exported: {missing}"
    assert lib.mu_version() >= 100


def test_ffi_table_matches_header():
    from muon_amd import _ffi

    assert sorted(_ffi.SIGNATURES) == _declared()
    _ffi.lib()  # loads and binds every entry


def test_error_reporting_without_gpu_calls():
    from muon_amd import _ffi

    lib = _ffi.lib()
    # argument validation happens before any HIP call: usable without a device
    rc = lib.mu_spmm_f32(1, 1, None, None, None, None, 7, None, 0, None)
    assert rc == -1 and b"B must be" in lib.mu_last_error()
    rc = lib.mu_tfidf_scale(0, 0, None, None, None, None, None, 1.0, 4 | 1, None, None, None)
    assert rc == -1 and b"log_tfidf" in lib.mu_last_error()


def test_product_never_imports_oracle():
    pat = re.compile(r"^\s*(from|import)\s+oracle\b", re.M)
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "muon_amd")):
        for f in files:
            if f.endswith(".py"):
                p = os.path.join(base, f)
                if pat.search(open(p).read()):
                    bad.append(p)
    assert not bad, bad


def test_no_reference_sources_copied():
    # the reference is read where it lies; only fixtures produced by executing it are committed
    assert not os.path.exists(os.path.join(ROOT, "muon"))
    for f in os.listdir(os.path.join(ROOT, "tests", "golden")):
        assert f.endswith((".npz", ".py")), f
