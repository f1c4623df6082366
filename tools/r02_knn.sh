#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
OUT=gpurun_out/r02_knn
mkdir -p $OUT
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_models_gpu.py -q -k "knn" > $OUT/pytest.txt 2>&1
grep -n "^E  \|passed\|failed\|FAILED" $OUT/pytest.txt | cut -c1-250 | head -20
echo "--- 3 waves/SIMD (default build)"; python tools/bench_knn.py 2>/dev/null
echo "--- 4 waves/SIMD"; GE_LIB_PATH=$PWD/graphecho_amd/csrc/libge_knn4.so python tools/bench_knn.py 2>/dev/null
