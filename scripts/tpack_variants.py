import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from muon_amd._atac.preproc import tfidf_device
from muon_amd._backend import HipBackend
be = HipBackend(0)
X = be.synth_counts(0, 125000, 200000, 50, 0.03, 0)
T = tfidf_device(be, X, 125000, 3, 1e4)
for sort in (False, True):
    for c in (0, 512):
        for abl in (0, 8):
            be.tune("tpack_c", c); be.tune("tpack_abl", abl)
            be.transpose_stream(T, sort_rows=sort); torch.cuda.synchronize()
