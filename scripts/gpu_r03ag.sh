#!/bin/bash
mkdir -p gpurun_out/r03ag
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r03ag/prof -o ng --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --workload mofa_ng > $GRAFT_REPO_ROOT/gpurun_out/r03ag/ng.json 2>/dev/null
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/r03ag/prof -name "*kernel_trace.csv" | head -1); python scripts/kstats.py "$f" mofa_ng | cut -c1-165 | head -36
find gpurun_out/r03ag -name "*.csv" -size +3M -delete
