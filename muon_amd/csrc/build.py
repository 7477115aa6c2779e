#!/usr/bin/env python
"""Build libmuon_amd.so for gfx950 with hipcc (in-tree, next to the sources).

hipcc cross-compiles without a GPU.  The shared object is git-ignored but travels to
the GPU box with the repo snapshot.  Usage: python muon_amd/csrc/build.py [--force]
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SOURCES = ["runtime.hip", "tfidf.hip", "transpose.hip", "spmm.hip", "spmm_win.hip", "spmm_narrow.hip", "spmm_ell.hip", "tpack4.hip", "dense.hip", "skinny.hip", "synth.hip", "mofa.hip", "mofa_elbo.hip", "mofa_stats.hip", "mofa_poisson.hip", "mofa_bernoulli.hip", "knn.hip", "wnn.hip"]
HEADERS = ["common.hpp", "sweep.hpp", os.path.join(ROOT, "include", "muon_amd.h")]
# -amdgpu-mfma-vgpr-form: MFMA accumulators in VGPRs.  The default (AGPR form) kept the accumulators of these files'
# loops in VGPRs BETWEEN the steps and copied them to AGPRs and back around the MFMAs of every step (k_skinny_tn: 64
# v_accvgpr_write + 64 v_accvgpr_read per 8 MFMAs); gfx950's MFMAs take either register file.
_VGPR_FORM = ["-mllvm", "-amdgpu-mfma-vgpr-form"]
EXTRA = {"skinny.hip": _VGPR_FORM, "dense.hip": _VGPR_FORM, "knn.hip": _VGPR_FORM, "mofa_poisson.hip": _VGPR_FORM, "mofa_bernoulli.hip": _VGPR_FORM}
LIB = os.path.join(HERE, "libmuon_amd.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-I" + os.path.join(ROOT, "include"),
         "-I" + HERE, "-Wall", "-Wno-unused-function"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _digest(paths):
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(f for f in FLAGS if not f.startswith("-I")).encode())  # paths differ per box
    h.update(repr(sorted(EXTRA.items())).encode())
    h.update(_compiler_id().encode())  # (kernels with asm-issued loads rest on what THIS hipcc does with their registers)
    return h.hexdigest()


def _compiler_id():
    """First line of `hipcc --version` that names the HIP / clang build ("" if the compiler cannot be asked)."""
    try:
        out = subprocess.run([_hipcc(), "--version"], capture_output=True, text=True, timeout=30).stdout
        return " | ".join(l.strip() for l in out.splitlines() if "version" in l.lower())[:300]
    except Exception:  # noqa: BLE001
        return ""


def build(force=False, verbose=True):
    srcs = [os.path.join(HERE, s) for s in SOURCES if os.path.exists(os.path.join(HERE, s))]
    hdrs = [h if os.path.isabs(h) else os.path.join(HERE, h) for h in HEADERS]
    stamp = os.path.join(HERE, ".build_stamp")
    dig = _digest(srcs + hdrs)
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dig:
        return LIB
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    procs = []
    objs = []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s) + ".o")
        objs.append(o)
        cmd = [hipcc] + FLAGS + EXTRA.get(os.path.basename(s), []) + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {s}")
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs + ["-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(stamp, "w") as f:
        f.write(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
