#!/bin/bash
# HBM traffic of the row-stream SpMM: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes
# (TCC slot limit; MI355X_MICROARCH.md §HBM), plus a calibration copy kernel (torch clone of a
# known byte count) in the same passes to fix the gfx950 unit/undercount of FETCH_SIZE.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out/${1:-traffic}
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c -d "$OUT/$c" -o pmc --output-format csv -- python "$OLDPWD/scripts/spmm_probe.py" --reps 1 --modes "" --no-packed --calibrate --cells ${CELLS:-125000} > "$OUT/$c.log" 2>&1
  echo "$c rc=$?"
done
du -sh "$OUT"
