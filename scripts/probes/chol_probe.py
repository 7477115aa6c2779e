#!/usr/bin/env python
"""mu_chol_rinv_f64 (the B x B step of CholeskyQR on the device): time per call and error against numpy."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from muon_amd._backend import HipBackend
be = HipBackend(0)
A = torch.randn(200000, 64, device="cuda")
G, _ = be.gram(A)
flag = be.zeros((1,), torch.int32)
M = be.chol_rinv(G, 64, flag)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(100):
    M = be.chol_rinv(G, 64, flag)
e.record(); torch.cuda.synchronize()
print(f"chol_rinv: {s.elapsed_time(e)*10:.1f} us per call, flag {int(flag.item())}")
R = np.linalg.cholesky(G.cpu().numpy()).T
ref = np.linalg.inv(R)
print("max rel err vs numpy:", float(np.abs(M.cpu().numpy() - ref).max() / np.abs(ref).max()))
