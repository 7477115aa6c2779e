#!/bin/bash
mkdir -p gpurun_out/r03aa
timeout 600 python -c "
import cProfile, pstats, sys, runpy
sys.argv=['wnn_probe.py','100000']
cProfile.run('runpy.run_path(\"scripts/wnn_probe.py\", run_name=\"__main__\")', 'gpurun_out/r03aa/prof.out')
p=pstats.Stats('gpurun_out/r03aa/prof.out'); p.sort_stats('cumulative').print_stats(45)
" 2>&1 | grep -v amdgpu.ids > gpurun_out/r03aa/cprof.txt
head -90 gpurun_out/r03aa/cprof.txt | cut -c1-150
