"""Repository contracts that need no GPU: the C-ABI library loads and exports every symbol
include/muon_amd.h declares, the ctypes table covers them, and the product package never
touches the oracle (which is test infrastructure)."""
import ctypes
import os
import re

import pytest

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


def test_stream_spmm_isa_keeps_out_of_the_asm_owned_registers(tmp_path):
    """The row-stream SpMM issues its window requests from inline asm into v[110..125], registers hipcc
    must never touch (a compiler copy of a register whose load is in flight reads stale data), and
    hipcc must not spill (its scratch loads would share vmcnt with the hand-counted requests).
    Audit the generated gfx950 ISA of every instance: no scratch, no compiler-issued instruction
    naming v110+, 126 allocated VGPRs."""
    import re
    import shutil
    import subprocess

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, "muon_amd", "csrc", "spmm_win.hip")
    out = tmp_path / "spmm_win.s"
    subprocess.check_call([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "muon_amd", "csrc"), "-S", "--cuda-device-only", "-w",
                           "-o", str(out), src])
    text = out.read_text()
    kernels = re.findall(r"^(_ZN[^\n:]*k_spmm_win[^\n:]*):[^\n]*\n(.*?)\.end_amdhsa_kernel", text,
                         flags=re.S | re.M)
    # production instances only (MODE = 0, the second template argument); the timing ablations never
    # feed results to anybody
    kernels = [(n, b) for n, b in kernels if re.search(r"k_spmm_winILi\d+ELi0ELi|k_spmm_win_rngILi\d+E", n)]
    # K = 1..8 x (f32: B = 64 / 32 / 16; f64 blocks: B = 32 / 16) + the ranged B = 64 instances of r06 (K = 1..8), whose
    # per-range set-up must not keep anything alive across the slab loop (K = 7 / 8 spilled before it was made opaque)
    assert len(kernels) == 48
    reg = re.compile(r"\bv(\d+)\b|v\[(\d+):(\d+)\]")
    for name, body in kernels:
        assert "scratch_" not in body, name
        assert re.search(r"\.amdhsa_private_segment_fixed_size 0\b", body), name
        m = re.search(r"\.amdhsa_next_free_vgpr (\d+)", body)
        assert m and int(m.group(1)) == 126, (name, m and m.group(1))
        inasm = False
        for line in body.splitlines():
            if "#ASMSTART" in line:
                inasm = True
            elif "#ASMEND" in line:
                inasm = False
            elif not inasm and not line.lstrip().startswith((".", ";")):
                for a, b, c in reg.findall(line.split(";")[0]):
                    hi = int(a) if a else int(c)
                    assert hi < 110, (name, line)




def test_poisson_matrix_core_sweep_isa(tmp_path):
    """k_pois_mfma (csrc/mofa_poisson.hip, r06): the sweep costs what its instruction count says - the f32 matrix
    instructions and the vector ALU do not overlap there - so the generated gfx950 ISA is pinned: accumulators in VGPRs
    (no v_accvgpr copies around every product: the file is built with -amdgpu-mfma-vgpr-form, csrc/build.py), no spills,
    no IEEE division sequence and no libm log expansion in the transform (one v_exp_f32 per prediction, one v_rcp_f32
    where the mode needs the sigmoid, one v_log_f32 where it needs the likelihood), KP / 4 + 4 matrix instructions per
    16 x 16 tile (KP / 4 in the likelihood-only mode).  f64 models: the same kernel on v_mfma_f64_16x16x4_f64 (two own
    tiles per wave, libm in the transform)."""
    import re
    import shutil
    import subprocess

    from muon_amd.csrc import build as csrc_build

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, "muon_amd", "csrc", "mofa_poisson.hip")
    assert "-amdgpu-mfma-vgpr-form" in csrc_build.EXTRA["mofa_poisson.hip"]
    out = tmp_path / "mofa_poisson.s"
    subprocess.check_call([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "muon_amd", "csrc"), *csrc_build.EXTRA["mofa_poisson.hip"], "-S",
                           "--cuda-device-only", "-w", "-o", str(out), src])
    text = out.read_text()
    kernels = re.findall(r"^(_ZN[^\n:]*k_pois_mfmaI([fd])Li(\d+)ELi(\d)E[^\n:]*):[^\n]*\n(.*?)\.end_amdhsa_kernel", text,
                         flags=re.S | re.M)
    assert len(kernels) == 32  # (f32, f64) x KP = 4, 8, 12, 16 x modes 0..3
    for name, t, kp, mode, body in kernels:
        kp, mode = int(kp), int(mode)
        tiles = 4 if t == "f" else 2  # PmMap<T>::OWN: 16-row own tiles per wave, all in one unrolled step
        assert "scratch_" not in body and re.search(r"\.amdhsa_private_segment_fixed_size 0\b", body), name
        assert "accvgpr" not in body, name
        count = lambda op: len(re.findall(r"^\s*" + op + r"\b", body, flags=re.M))
        op = "v_mfma_f32_16x16x4_f32" if t == "f" else "v_mfma_f64_16x16x4_f64"
        assert count(op) == tiles * (kp // 4 + (0 if mode == 2 else 4)), name
        m = re.search(r"\.amdhsa_next_free_vgpr (\d+)", body)
        assert m and int(m.group(1)) <= 160, (name, m and m.group(1))
        if t == "d":
            continue  # (f64: libm's exp / log1p in the transform)
        assert "v_div_" not in body and "v_ldexp" not in body, name
        assert count("v_exp_f32_e32") + count("v_exp_f32_e64") == 4 * tiles, name
        assert count("v_rcp_f32_e32") + count("v_rcp_f32_e64") == (0 if mode == 2 else 4 * tiles), name
        assert count("v_log_f32_e32") + count("v_log_f32_e64") == (4 * tiles if mode in (2, 3) else 0), name


def test_jaakkola_sweep_isa(tmp_path):
    """k_jaakkola_sweep (csrc/mofa_bernoulli.hip, r06): 14 instances - (padded width, column tiles of the packed moments) x
    f32 / f64 - each without spills and with its accumulators in VGPRs; the matrix-instruction count of a 16-row step is
    OWN x (3 KP / 4 + 4 CT): the prediction and the two variance products, then one product per column tile and
    accumulator register."""
    import re
    import shutil
    import subprocess

    from muon_amd.csrc import build as csrc_build

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, "muon_amd", "csrc", "mofa_bernoulli.hip")
    assert "-amdgpu-mfma-vgpr-form" in csrc_build.EXTRA["mofa_bernoulli.hip"]
    out = tmp_path / "mofa_bernoulli.s"
    subprocess.check_call([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "muon_amd", "csrc"), *csrc_build.EXTRA["mofa_bernoulli.hip"], "-S",
                           "--cuda-device-only", "-w", "-o", str(out), src])
    text = out.read_text()
    kernels = re.findall(r"^(_ZN[^\n:]*k_jaakkola_sweepI([fd])Li(\d+)ELi(\d+)E[^\n:]*):[^\n]*\n(.*?)\.end_amdhsa_kernel", text,
                         flags=re.S | re.M)
    assert sorted((t, int(kp), int(ct)) for _, t, kp, ct, _ in kernels) == sorted(
        (t, kp, ct) for t in "df" for kp, ct in ((4, 1), (8, 2), (8, 3), (12, 4), (12, 5), (16, 7), (16, 9)))
    for name, t, kp, ct, body in kernels:
        kp, ct = int(kp), int(ct)
        own = 2 if (t == "f" and ct <= 5) else 1
        assert "scratch_" not in body and re.search(r"\.amdhsa_private_segment_fixed_size 0\b", body), name
        assert "accvgpr" not in body, name
        op = "v_mfma_f32_16x16x4_f32" if t == "f" else "v_mfma_f64_16x16x4_f64"
        assert len(re.findall(r"^\s*" + op + r"\b", body, flags=re.M)) == own * (3 * kp // 4 + 4 * ct), name


def test_tpack4_isa_keeps_out_of_the_asm_owned_registers(tmp_path):
    """The fourth-generation transposition (csrc/tpack4.hip) keeps the next tile's header in v[88..91] and its 18 window
    slots in v[92..127], written by loads issued from inline asm one tile ahead: hipcc must stay below v88 (a copy or a
    spill of a register whose load is in flight reads stale data) and must not spill at all in the production instances
    (scratch traffic shares vmcnt with the hand-placed waits)."""
    import re
    import shutil
    import subprocess

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, "muon_amd", "csrc", "tpack4.hip")
    out = tmp_path / "tpack4.s"
    subprocess.check_call([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "muon_amd", "csrc"), "-S", "--cuda-device-only", "-w",
                           "-o", str(out), src])
    text = out.read_text()
    kernels = re.findall(r"^(_ZN[^\n:]*k_t4_fill[^\n:]*):[^\n]*\n(.*?)\.end_amdhsa_kernel", text, flags=re.S | re.M)
    assert len(kernels) == 7  # (row stream | CSR arrays) x (stream target, CSR target, stream target with phase accounting) + the row stream through circular windows
    reg = re.compile(r"\bv(\d+)\b|v\[(\d+):(\d+)\]")
    for name, body in kernels:
        production = re.search(r"k_t4_fillILb[01]ELb0ELb[01]E", name) is not None
        m = re.search(r"\.amdhsa_next_free_vgpr (\d+)", body)
        assert m and int(m.group(1)) == 128, (name, m and m.group(1))
        if production:
            assert "scratch_" not in body, name
            assert re.search(r"\.amdhsa_private_segment_fixed_size 0\b", body), name
        inasm = False
        for line in body.splitlines():
            if "#ASMSTART" in line:
                inasm = True
            elif "#ASMEND" in line:
                inasm = False
            elif not inasm and not line.lstrip().startswith((".", ";")):
                for a, b, c in reg.findall(line.split(";")[0]):
                    hi = int(a) if a else int(c)
                    assert hi < 88, (name, line)


def test_every_environment_switch_is_in_the_table_of_integration_md():
    """ONE table of the switches - every MUON_AMD_* variable the package or
    bench.py reads must be a row of it, and every row must still be read somewhere."""
    import re

    read = set()
    for base, _dirs, files in os.walk(os.path.join(ROOT, "muon_amd")):
        for f in files:
            if f.endswith(".py"):
                read |= set(re.findall(r"MUON_AMD_[A-Z0-9_]+", open(os.path.join(base, f)).read()))
    read |= set(re.findall(r"MUON_AMD_[A-Z0-9_]+", open(os.path.join(ROOT, "bench.py")).read()))
    read |= set(re.findall(r"MUON_AMD_BENCH_[A-Z0-9_]+", open(os.path.join(ROOT, "scripts", "bench_mofa.py")).read()))
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    table = text[text.index("**Switches** (environment"):text.index("Folded in r06 (no longer read)")]
    rows = set(re.findall(r"MUON_AMD_[A-Z0-9_]+", table))
    assert read <= rows, sorted(read - rows)
    assert rows <= read, sorted(rows - read)
    folded = set(re.findall(r"MUON_AMD_[A-Z0-9_]+", text[text.index("Folded in r06 (no longer read)"):]))
    assert not (folded & read), sorted(folded & read)
