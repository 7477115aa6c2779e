#!/bin/bash
# A/B on one box: c4 with the narrow-block SpMM and with the NB = 1 instance of the B = 64 kernel, alternating
mkdir -p gpurun_out/r03ad
for i in 1 2 3; do
  for off in 0 1; do
    MUON_AMD_BENCH_TUNE="spmm_narrow_off=$off" timeout 300 python bench.py --workload c4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('narrow_off=$off', round(d['value'],4))"
  done
done | tee gpurun_out/r03ad/ab.txt
