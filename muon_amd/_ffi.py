"""ctypes binding of libmuon_amd.so (the C-ABI declared in include/muon_amd.h).

The shared object is built in-tree by ``muon_amd/csrc/build.py`` (hipcc, gfx950).  There is
no fallback: if the library is missing, or no MI355X-class device is visible when a kernel
is requested, the product path raises ``MuonAmdError``.

torch is imported *before* the library is loaded on purpose: the PyTorch-ROCm wheel ships its
own ``libamdhip64.so`` (SONAME ``libamdhip64.so.7``); loading ours afterwards makes the dynamic
linker reuse that already-loaded runtime, so device pointers of torch tensors are valid inside
our kernels and there is exactly one HIP runtime in the process.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libmuon_amd.so")

F32, F64 = 0, 1
TFIDF_LOG_TF, TFIDF_LOG_IDF, TFIDF_LOG_TFIDF = 1, 2, 4


class MuonAmdError(RuntimeError):
    pass


_lib = None
_lock = threading.Lock()

_i64, _i32, _dbl, _vp, _sz, _u64 = C.c_int64, C.c_int, C.c_double, C.c_void_p, C.c_size_t, C.c_uint64

# name -> (restype, argtypes); mirrors include/muon_amd.h one to one
SIGNATURES = {
    "mu_version": (C.c_int, []),
    "mu_last_error": (C.c_char_p, []),
    "mu_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "mu_set_device": (C.c_int, [_i32]),
    "mu_device_info": (C.c_int, [_i32, C.c_char_p, _i32, C.POINTER(C.c_int), C.POINTER(_sz)]),
    "mu_malloc": (C.c_int, [C.POINTER(_vp), _sz]),
    "mu_free": (C.c_int, [_vp]),
    "mu_memcpy_h2d": (C.c_int, [_vp, _vp, _sz, _vp]),
    "mu_memcpy_d2h": (C.c_int, [_vp, _vp, _sz, _vp]),
    "mu_memset": (C.c_int, [_vp, _i32, _sz, _vp]),
    "mu_stream_sync": (C.c_int, [_vp]),
    "mu_host_hash64": (C.c_int, [_vp, _sz, _i32, _u64, C.POINTER(_u64)]),
    "mu_csr_row_col_sums_worksize": (_sz, [_i64, _i64]),
    "mu_csr_row_col_sums": (C.c_int, [_i32, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "mu_tfidf_idf": (C.c_int, [_i32, _i64, _dbl, _vp, _i32, _vp, _vp]),
    "mu_tfidf_scale": (C.c_int, [_i32, _i64, _vp, _vp, _vp, _vp, _vp, _dbl, _i32, _vp, _vp, _vp]),
    "mu_tfidf_scale_sweep": (C.c_int, [_i32, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _dbl, _i32, _vp, _vp, _vp,
                                       _sz, _i32, _vp]),
    "mu_csr_count_nonzero": (C.c_int, [_i32, _i64, _vp, _vp, _vp, _vp]),
    "mu_csr_compact_nonzero": (C.c_int, [_i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mu_exclusive_scan_i64": (C.c_int, [_i64, _vp, _vp, _vp]),
    "mu_binarize_values": (C.c_int, [_i32, _i64, _vp, _vp]),
    "mu_csr_transpose_worksize": (_sz, [_i64, _i64, _i64]),
    "mu_csr_transpose": (C.c_int, [_i32, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "mu_spmm_f32": (C.c_int, [_i64, _i64, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _vp]),
    "mu_spmm_stream_k": (C.c_int, [_i64]),
    "mu_tfidf_scale_sweep_stream": (C.c_int, [_i64, _i64, _vp, _vp, _vp, _vp, _vp, _dbl, _i32, _vp, _vp, _vp, _sz, _i32,
                                              _vp, _vp, _vp, _vp]),
    "mu_csr_slab_ptr": (C.c_int, [_i64, _i64, _vp, _vp, _vp, _vp]),
    "mu_csr_slab_ptr_width": (C.c_int, [_i64, _i64, _i64, _vp, _vp, _vp, _vp]),
    "mu_csr_row_col_sums_sp": (C.c_int, [_i32, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    "mu_tfidf_scale_sweep_sp": (C.c_int, [_i32, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _dbl, _i32, _vp, _vp, _vp, _vp]),
    "mu_tpack4_supported": (C.c_int, [_i64, _i64, _i64]),
    "mu_tpack4_geometry": (C.c_int, [_i64, _i64, _i64, C.POINTER(_i64), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mu_tpack4_worksize": (_sz, [_i64, _i64, _i64]),
    "mu_tpack4_count": (C.c_int, [_i64, _i64, _i64, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    "mu_tpack4_fill_stream": (C.c_int, [_i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "mu_tpack4_fill_csr": (C.c_int, [_i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "mu_tpack4_status": (C.c_int, [_vp, _i64, _i64, _i64, C.POINTER(C.c_int)]),
    "mu_tpack4_err_offset": (_sz, [_i64, _i64, _i64]),
    "mu_tpack4_phase_cycles": (C.c_int, [_vp, _i32]),
    "mu_csr_stream_len": (C.c_int, [_i64, _vp, _vp, _vp, _vp]),
    "mu_csr_stream_fill": (C.c_int, [_i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mu_spmm_stream_f32": (C.c_int, [_i64, _i64, _vp, _vp, _vp, _i32, _vp, _i32, _vp, _vp]),
    "mu_spmm_stream_f64": (C.c_int, [_i64, _i64, _vp, _vp, _vp, _i32, _vp, _i32, _vp, _i32, _vp]),
    "mu_csr_slice_stream": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mu_spmm_stream_ranges_f32": (C.c_int, [_i64, _vp, _vp, _vp, _i32, _vp, _i64, _vp, _i64, _vp, _i64, _i32, _vp, _i32, _vp]),
    "mu_tpack4_cnt_offset": (_sz, [_i64, _i64, _i64]),
    "mu_spmm_ell16_waves": (C.c_int, [_i64]),
    "mu_dense_col_moments_chunks": (C.c_int, [_i64, _i64]),
    "mu_dense_col_moments": (C.c_int, [_i32, _i64, _i64, _i64, _vp, _i32, _vp, _vp]),
    "mu_ell16_fill": (C.c_int, [_i64, _i64, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mu_spmm_ell16_f32": (C.c_int, [_i32, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mu_spmm_ell16_f64": (C.c_int, [_i32, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "mu_spmm_ell16_parts": (C.c_int, [_i64, _i64, _i32, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mu_spmm_ell16_parts_f32": (C.c_int, [_i32, _i32, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "mu_spmm_ell16_parts_f64": (C.c_int, [_i32, _i32, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "mu_tune_set": (C.c_int, [C.c_char_p, _i32]),
    "mu_tune_get": (C.c_int, [C.c_char_p]),
    "mu_spmm_f64": (C.c_int, [_i64, _i64, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _vp]),
    "mu_gram_worksize": (_sz, [_i64, _i32]),
    "mu_gram_f32": (C.c_int, [_i64, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "mu_gram_cross_f32": (C.c_int, [_i64, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "mu_dense_apply_f32": (C.c_int, [_i64, _i32, _vp, _vp, _vp, _vp, _vp]),
    "mu_dense_project_out_f32": (C.c_int, [_i64, _i32, _vp, _vp, _vp, _vp]),
    "mu_chol_rinv_f64": (C.c_int, [_i32, _i32, _vp, _vp, _vp, _vp]),
    "mu_randn_f32": (C.c_int, [_i64, _u64, _vp, _vp]),
    "mu_skinny_tn_worksize": (_sz, [_i32, _i64, _i64]),
    "mu_skinny_nn": (C.c_int, [_i32, _i64, _i64, _i64, _vp, _vp, _vp, _vp]),
    "mu_skinny_tn": (C.c_int, [_i32, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "mu_skinny_nn_f64_f32": (C.c_int, [_i64, _i64, _i64, _vp, _vp, _vp, _vp]),
    "mu_skinny_tn_f64_f32": (C.c_int, [_i64, _i64, _i64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "mu_mofa_update_w": (C.c_int, [_i32, _i64, _i32, _i32] + [_vp] * 7 + [_i32] + [_vp] * 6),
    "mu_mofa_update_z": (C.c_int, [_i32, _i64, _i32, _i32, _i32] + [_vp] * 11),
    "mu_mofa_rowstats_work_doubles": (_sz, [_i32]),
    "mu_mofa_rowstats": (C.c_int, [_i32, _i64, _i64, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _i32, _vp, _i64,
                                   _vp, _vp, _vp, _vp, _vp]),
    "mu_umap_strengths_f64": (C.c_int, [_i64, _i32, _vp, _vp, _dbl, _dbl, _vp, _vp]),
    "mu_wnn_bandwidth_f64": (C.c_int, [_i64, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _dbl, _vp, _vp, _vp]),
    "mu_knn_filter_f64": (C.c_int, [_i64, _i64, _i64, _i32] + [_vp] * 6 + [_i32] + [_vp] * 4),
    "mu_knn_merge_f64": (C.c_int, [_i64, _i32, _i32] + [_vp] * 9),
    "mu_csr_densify_rows": (C.c_int, [_i32, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "mu_mofa_jaakkola": (C.c_int, [_i32, _i64, _vp, _vp, _vp, _vp, _vp]),
    "mu_mofa_poisson_pseudo": (C.c_int, [_i32, _i64, _i64, _i32, _vp, _vp, _vp, _vp, _vp]),
    "mu_mofa_poisson_blocks": (_i64, [_i64, _i64]),
    "mu_mofa_poisson_blocks_for": (_i64, [_i32, _i32, _i32, _i64, _i64]),
    "mu_mofa_poisson_dense": (C.c_int, [_i32, _i32, _i64, _i64, _i32, _i64, _vp, _vp, _vp, _vp, _vp]),
    "mu_mofa_poisson_sparse": (C.c_int, [_i32, _i32, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mu_mofa_poisson_dense_ld": (C.c_int, [_i32, _i32, _i64, _i64, _i32, _i32, _i64, _vp, _vp, _vp, _vp, _vp]),
    "mu_mofa_poisson_sparse_ld": (C.c_int, [_i32, _i32, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mu_mofa_jaakkola_cols": (C.c_int, [_i32]),
    "mu_mofa_pack_moments": (C.c_int, [_i32, _i64, _i32, _i32, _vp, _vp, _vp, _vp]),
    "mu_mofa_jaakkola_blocks": (_i64, [_i32, _i32, _i64, _i64]),
    "mu_mofa_jaakkola_sweep": (C.c_int, [_i32, _i64, _i64, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp]),
    "mu_mofa_gs_update": (C.c_int, [_i32, _i64, _i32] + [_vp] * 5 + [_i32] + [_vp] * 6),
    "mu_mofa_elbo_work_doubles": (_sz, [_i32]),
    "mu_mofa_tau_elbo": (C.c_int, [_i32, _i64, _i32, _i32] + [_vp] * 7 + [_dbl, _dbl] + [_vp] * 5),
    "mu_mofa_stats_resid": (C.c_int, [_i32, _i64, _i32, _i64] + [_vp] * 7),
    "mu_mofa_tau_finish": (C.c_int, [_i32, _i64, _vp, _vp, _dbl, _dbl] + [_vp] * 5),
    "mu_mofa_w_elbo": (C.c_int, [_i32, _i64, _i32, _i32, _i32] + [_vp] * 3 + [_dbl] * 5 + [_vp] * 7),
    "mu_mofa_z_sums": (C.c_int, [_i32, _i64, _i64, _i32] + [_vp] * 5),
    "mu_mofa_z_elbo": (C.c_int, [_i32, _i32, _i32, _i32, _vp, _vp, _dbl, _dbl] + [_vp] * 4),
    "mu_synth_row_nnz": (C.c_int, [_i64, _i64, _i64, _i32, _dbl, _u64, _vp, _vp]),
    "mu_synth_fill": (C.c_int, [_i64, _i64, _i64, _i32, _dbl, _u64, _vp, _vp, _vp, _vp]),
}


def lib():
    """Load (once) and return the ctypes handle of libmuon_amd.so."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise MuonAmdError(
                f"{LIB_PATH} is missing: build it with `python muon_amd/csrc/build.py` "
                "(or __graft_entry__.build()). muon_amd has no CPU fallback."
            )
        try:
            import torch  # noqa: F401  (loads the HIP runtime we must share; see module docstring)
        except Exception:  # pragma: no cover - torch is part of the image
            pass
        handle = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        msg = lib().mu_last_error()
        raise MuonAmdError(f"libmuon_amd error {rc}: {msg.decode() if msg else '?'}")


def device_count() -> int:
    n = C.c_int(0)
    rc = lib().mu_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def require_gpu() -> None:
    """Fail loudly when the HIP path cannot run (no silent CPU fallback)."""
    import torch

    if not torch.cuda.is_available() or device_count() <= 0:
        raise MuonAmdError(
            "muon_amd needs an AMD Instinct (gfx950) GPU: no HIP device is visible and there "
            "is deliberately no CPU fallback. Use the reference (scverse/muon) on CPU instead."
        )
