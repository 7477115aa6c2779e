#!/bin/bash
mkdir -p gpurun_out/f32store
rm -f gpurun_out/f32store/mofa.txt
timeout 900 python -m pytest tests/test_gpu_mofa.py tests/test_gpu_kernels.py -m gpu -x -q 2>&1 | tail -4 > gpurun_out/f32store/tests.txt
for cfg in "1 tn_pipe=1" "0 tn_pipe=1" "0 tn_pipe=1,nn_fast_off=1"; do
  set -- $cfg
  MUON_AMD_MOFA_F32_STORAGE=$1 MUON_AMD_BENCH_TUNE=$2 timeout 300 python scripts/bench_mofa.py --iters 100 --f64 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('f64 storage=$1 $2', d['value'])" >> gpurun_out/f32store/mofa.txt
done
for t in tn_pipe=1 tn_pipe=1,nn_fast_off=1; do
  MUON_AMD_BENCH_TUNE=$t timeout 300 python scripts/bench_mofa.py --iters 100 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('f32 $t', d['value'])" >> gpurun_out/f32store/mofa.txt
done
cat gpurun_out/f32store/tests.txt gpurun_out/f32store/mofa.txt
MUON_AMD_BENCH_TUNE=tn_pipe=1 bash scripts/gpu_c4_stats.sh r04d >/dev/null 2>&1; grep -n "skinny\|ell16\|matmul\|Cijk" gpurun_out/r04d/r04d_c4_f64_kernel_stats.md gpurun_out/r04d/r04d_c4_kernel_stats.md | cut -c1-200
