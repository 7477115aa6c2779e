#!/bin/bash
# MFMA utilisation of the dense steps (LSI Gram / apply, MOFA tall-skinny products): PMC passes only.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out/${1:-mfma}
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L 2>/dev/null | grep -i -E "mfma" | head -20 > "$OUT/counters.txt"
timeout 250 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_MOPS_F32 -d "$OUT/mofa64" -o pmc --output-format csv -- python "$OLDPWD/scripts/bench_mofa.py" --f64 --iters 3 --warmup 1 > "$OUT/mofa64.log" 2>&1
echo "mofa64 rc=$?"
timeout 250 rocprofv3 --pmc GRBM_GUI_ACTIVE -d "$OUT/mofa64g" -o pmc --output-format csv -- python "$OLDPWD/scripts/bench_mofa.py" --f64 --iters 3 --warmup 1 > "$OUT/mofa64g.log" 2>&1
timeout 250 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_MOPS_F32 -d "$OUT/lsi" -o pmc --output-format csv -- python "$OLDPWD/bench.py" --workload c3shard --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/lsi.log" 2>&1
echo "lsi rc=$?"
timeout 250 rocprofv3 --pmc GRBM_GUI_ACTIVE -d "$OUT/lsig" -o pmc --output-format csv -- python "$OLDPWD/bench.py" --workload c3shard --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/lsig.log" 2>&1
cat "$OUT/counters.txt"; du -sh "$OUT"
