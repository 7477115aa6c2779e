#!/bin/bash
# r04: parity of the pipelined TF-IDF sweeps / interleaved nn product, then A/B timings
mkdir -p gpurun_out/pipe
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_tfidf.py tests/test_gpu_mofa.py tests/test_gpu_lsi.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/pipe/tests.txt
timeout 400 python scripts/tfidf_probe.py 1000000 > gpurun_out/pipe/tfidf_1m.txt 2>&1
timeout 200 python scripts/tfidf_probe.py 125000 > gpurun_out/pipe/tfidf_125k.txt 2>&1
for t in 1 0 1 0; do
  MUON_AMD_BENCH_TUNE=nn_interleave=$t timeout 300 python scripts/bench_mofa.py --iters 100 --f64 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400 >> gpurun_out/pipe/mofa_f64.txt
done
for t in 1 0; do
  MUON_AMD_BENCH_TUNE=nn_interleave=$t timeout 300 python scripts/bench_mofa.py --iters 100 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400 >> gpurun_out/pipe/mofa_f32.txt
done
cat gpurun_out/pipe/tests.txt; tail -6 gpurun_out/pipe/tfidf_1m.txt; tail -5 gpurun_out/pipe/tfidf_125k.txt; cat gpurun_out/pipe/mofa_f64.txt gpurun_out/pipe/mofa_f32.txt
