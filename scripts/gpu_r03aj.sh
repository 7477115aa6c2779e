#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "step_stream or spmm_stream_narrow" 2>&1 | tail -12
