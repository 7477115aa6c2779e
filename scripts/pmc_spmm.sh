#!/bin/bash
# PMC passes over the packed SpMM probe (counters only: no tracing domains alongside --pmc).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out/${1:-pmc}
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
run() {  # name, counters...
  local name=$1; shift
  timeout 170 rocprofv3 --pmc "$@" -d "$OUT/$name" -o pmc --output-format csv -- python "$OLDPWD/scripts/spmm_probe.py" --reps 1 --modes ${MODES:-0} > "$OUT/$name.log" 2>&1
  echo "$name rc=$?"
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY
run sq2 SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
find "$OUT" -name "*.csv" | head; du -sh "$OUT"
