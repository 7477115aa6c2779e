"""Host profile of the c2 step (10 000 x 30 000: tfidf + lsi on resident data): cProfile over 20 steps plus the
lsi's own wait / Ritz accounting.  usage: python scripts/probes/c2_host_profile.py [cells] [peaks]"""
import cProfile
import pstats
import sys
import time

import torch

sys.path.insert(0, ".")
from muon_amd._atac.preproc import tfidf_device  # noqa: E402
from muon_amd._atac.tools import lsi_device  # noqa: E402
from muon_amd._backend import get_backend  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 30000
    be = get_backend()
    X = be.synth_counts(0, n, d, 50, 0.0315, 0)
    info = {}

    def step():
        T = tfidf_device(be, X, n, 3, 1e4)
        out = lsi_device(be, T, 50, True, n_obs=n, return_info=True)
        info.update(out[3])
        return out

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    print("ms/step", ms, {k: info.get(k) for k in ("host", "iterations", "restarts", "n_products")}, flush=True)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(40)


if __name__ == "__main__":
    main()
