#!/bin/bash
mkdir -p gpurun_out/r03x
for w in ingest mofa_ng wnn; do
  timeout 600 python bench.py --workload $w > gpurun_out/r03x/$w.json 2> gpurun_out/r03x/$w.err
  echo "== $w rc=$?"; tail -c 1800 gpurun_out/r03x/$w.json; tail -3 gpurun_out/r03x/$w.err | grep -v amdgpu.ids
done
