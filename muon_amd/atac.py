"""``muon_amd.atac`` mirrors ``muon.atac`` (/root/reference/muon/atac.py:1) for the hot path:
``atac.pp.tfidf``, ``atac.pp.binarize``, ``atac.tl.lsi``."""
from ._atac import pp, tl  # noqa: F401
