#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/splitk
export PYTHONWARNINGS=ignore
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "conv2d" 2>&1 | tail -5
timeout 900 python tools/bench_splitk.py 2>&1 | tee gpurun_out/splitk/bench.txt
