#!/bin/bash
# r03: narrow-block SpMM: parity tests, the probe at the c4 shape, PMC counters
mkdir -p gpurun_out/r03u
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "spmm" > gpurun_out/r03u/tests.log 2>&1
tail -3 gpurun_out/r03u/tests.log
timeout 300 python scripts/probes/spmm_narrow_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r03u/probe.log
cat gpurun_out/r03u/probe.log
bash scripts/pmc_narrow.sh r03u/pmc > gpurun_out/r03u/pmc_summary.txt 2>&1
grep -E "narrow<10, false>|spmm_win" gpurun_out/r03u/pmc_summary.txt | grep -E "FETCH|ACTIVE_INST_VALU|INSTS_VALU|INSTS_LDS|BANK|IDX_ACTIVE|INSTS_SALU|WAVE_CYCLES"
