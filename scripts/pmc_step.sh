#!/bin/bash
# PMC passes over one TF-IDF + LSI step at 1e6 x 200k for the kernels besides the SpMM: the two TF-IDF
# sweeps, the transposition's count and fill, the streaming copy (counters only, separate passes).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out/${1:-pmc_step}
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
run() {  # name, counters...
  local name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" -d "$OUT/$name" -o pmc --output-format csv -- python "$OLDPWD/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-secondary > "$OUT/$name.log" 2>&1
  echo "$name rc=$?"
}
run fetch FETCH_SIZE
run write WRITE_SIZE
[ "${PMC_STEP_SHORT:-0}" = "1" ] || run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY
[ "${PMC_STEP_SHORT:-0}" = "1" ] || run sq2 SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
want = ("k_tfidf_scale_sweep", "k_row_col_sums", "k_t4_fill", "k_t4_count", "k_t_fill3", "k_t_count", "k_stream_fill", "k_slab_ptr", "k_t_slab_ptr")
agg = collections.defaultdict(float)
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        name = next((w for w in want if w in k), None)
        if name is None:
            continue
        agg[(name, r["Counter_Name"], r.get("Dispatch_Id"))] += float(r["Counter_Value"])
per = collections.defaultdict(list)
for (k, c, d), v in agg.items():
    per[(k, c)].append(v)
for (k, c), vs in sorted(per.items()):
    print(f"{k:24s} {c:24s} n={len(vs):3d} mean={sum(vs)/len(vs):.4e}")
PY
find "$OUT" -name "*.csv" -size +2M -delete
