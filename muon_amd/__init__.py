"""muon_amd - MI355X-native TF-IDF / LSI / MOFA hot path of scverse/muon.

Drop-in namespaces for the three hot-path entry points of the reference
(/root/reference/muon/__init__.py:6-14, muon/atac.py:1):

    muon_amd.atac.pp.tfidf   <->  muon.atac.pp.tfidf
    muon_amd.atac.tl.lsi     <->  muon.atac.tl.lsi
    muon_amd.tl.mofa         <->  muon.tl.mofa
    muon_amd.pp.neighbors    <->  muon.pp.neighbors   (the consumer of X_lsi / X_mofa)

Everything else of muon (I/O, plotting, clustering, ...) is out of scope; see DESIGN.md.
"""
from ._containers import AnnData, MuData  # duck-typed stand-ins when anndata/mudata are absent
from . import atac  # noqa: F401
from ._core import tools as tl  # noqa: F401
from ._core import preproc as pp  # noqa: F401  (mu.pp.neighbors - weighted nearest neighbours -, mu.pp.l2norm)
from ._core import io  # noqa: F401  (arrays of 10x / mtx / snap files -> row-sharded device CSR)

__version__ = "0.1.0"
