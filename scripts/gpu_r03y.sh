#!/bin/bash
mkdir -p gpurun_out/r03y
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r03y/prof -o wnn --output-format csv -- python $GRAFT_REPO_ROOT/scripts/wnn_probe.py 100000 > $GRAFT_REPO_ROOT/gpurun_out/r03y/wnn.log 2>&1
cd $GRAFT_REPO_ROOT
grep -v amdgpu.ids gpurun_out/r03y/wnn.log | tail -8
f=$(find gpurun_out/r03y/prof -name "*kernel_trace.csv" | head -1); python scripts/kstats.py "$f" wnn | cut -c1-150 | head -34
find gpurun_out/r03y -name "*.csv" -size +3M -delete
