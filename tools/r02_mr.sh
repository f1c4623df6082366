#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
OUT=gpurun_out/r02_mr
mkdir -p $OUT
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_models_gpu.py -q -k "mr or grapher or graphconv or reproducib" > $OUT/pytest.txt 2>&1
grep -n "^E  \|passed\|failed\|FAILED" $OUT/pytest.txt | cut -c1-250 | head -20
echo "--- quad gather"; python tools/bench_graph_path.py 2>/dev/null | grep "mr_gather\|knn_graph" | cut -c1-150
echo "--- scalar gather"; GE_MR_QUAD=0 python tools/bench_graph_path.py 2>/dev/null | grep "mr_gather_fwd" | cut -c1-150
