#!/bin/bash
# r03 profiles: kernel stats of c3 (default line without the secondary records), c3shard, c4; host profile of c2
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=${1:-r03p}
OUT=$PWD/gpurun_out/$R
mkdir -p "$OUT"
export TMPDIR=/tmp
for wl in c3 c3shard c4; do
  extra=""; [ $wl = c3 ] && extra="--no-secondary"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_$wl" -o bench -- python "$OLDPWD/bench.py" --workload $wl --steps 2 --warmup 1 --no-cpu-baseline $extra > "$OUT/prof_bench_$wl.json" 2> "$OUT/prof_$wl.err")
  db=$(find "$OUT/prof_$wl" -name "*.db" | head -1)
  [ -n "$db" ] && python scripts/kstats.py "$db" "rocprofv3 --kernel-trace --stats -- python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline $extra (1 GPU; incl. warm-up; k_synth = input generation outside the timed region)" > "$OUT/r03_${wl}_kernel_stats.md"
  rm -rf "$OUT/prof_$wl"
  head -14 "$OUT/r03_${wl}_kernel_stats.md" | cut -c1-160
done
timeout 300 python - > "$OUT/c2_host_profile.txt" 2>&1 <<'PY'
import cProfile, pstats, io, sys, os
sys.path.insert(0, os.getcwd())
import torch
from muon_amd._atac.preproc import tfidf_device
from muon_amd._atac.tools import lsi_device
from muon_amd._backend import HipBackend
be = HipBackend(0)
X = be.synth_counts(0, 10000, 30000, 50, 0.03, 0)
out = torch.empty_like(X.values)
def step():
    T = tfidf_device(be, X, 10000, 3, 1e4, out=out)
    return lsi_device(be, T, n_comps=50, n_obs=10000, return_info=True)
for _ in range(3): step()
torch.cuda.synchronize()
import time
t0=time.perf_counter()
for _ in range(20): r = step()
torch.cuda.synchronize()
print("ms/step", (time.perf_counter()-t0)/20*1e3, r[3]["host"], r[3]["iterations"], r[3]["restarts"])
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:9000])
PY
tail -70 "$OUT/c2_host_profile.txt"
