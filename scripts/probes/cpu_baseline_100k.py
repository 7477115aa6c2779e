"""One-off (r04, VERDICT r03 item 4 iv): the reference's CPU path - scipy tfidf + f32 ARPACK svds(k=50), the call
sequence of /root/reference/muon/_atac/preproc.py:92-119 and tools.py:53-69 restated in oracle/ - on 100 000 x
200 000 at 3 % (SURVEY 8d's planned sample), on this container's host cores.  No GPU.  Prints one JSON line."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from tests.synth import planted_topics_csr
from oracle import lsi_oracle, tfidf_oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
d = 200_000
t0 = time.perf_counter()
X = planted_topics_csr(n, d, n_topics=50, density=0.03, seed=0, dtype=np.float32, chunk=1024)
t1 = time.perf_counter()
c0 = time.process_time()
tf = tfidf_oracle.tfidf(X)
t2 = time.perf_counter()
ref = lsi_oracle.lsi(tf, n_comps=50, dtype=np.float32)
t3 = time.perf_counter()
busy = (time.process_time() - c0) / (t3 - t1)
print(json.dumps({"cells": n, "peaks": d, "nnz": int(X.nnz), "generate_s": t1 - t0, "tfidf_s": t2 - t1, "lsi_s": t3 - t2,
                  "cells_per_s": n / (t3 - t1), "cores_busy": busy, "host": f"builder container, {os.cpu_count()} cores",
                  "extrapolated_1e6_cells_s": (t3 - t1) * 10}), flush=True)
