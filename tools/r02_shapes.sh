#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r02_shapes
python tools/conv_shapes.py > gpurun_out/r02_shapes/conv_shapes.txt 2>&1
cat gpurun_out/r02_shapes/conv_shapes.txt | head -120
