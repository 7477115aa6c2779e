#!/bin/bash
# r04 round end: kernel statistics of c3 / c3shard / c4 / c4_f64 / wnn, the transposition and copy probes, then the
# default bench line and the whole GPU suite
mkdir -p gpurun_out/r04z
SKIP_PMC=1 bash scripts/gpu_profiles.sh r04z > /dev/null 2>&1
bash scripts/gpu_c4_stats.sh r04z > /dev/null 2>&1
bash scripts/gpu_wnn_stats.sh r04z > /dev/null 2>&1
timeout 300 python scripts/probes/tpack_asm_probe.py 1000000 > gpurun_out/r04z/tpack_asm_1m.txt 2>&1
timeout 200 python scripts/probes/tpack_asm_probe.py 125000 > gpurun_out/r04z/tpack_asm_125k.txt 2>&1
timeout 300 python scripts/probes/stream_pipe_probe.py 1000000 > gpurun_out/r04z/stream_pipe_1m.txt 2>&1
tail -3 gpurun_out/r04z/tpack_asm_1m.txt | cut -c1-130
bash scripts/gpu_final_check.sh
