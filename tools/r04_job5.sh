#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp PYTHONWARNINGS=ignore
OUT=gpurun_out/${1:-r04_j5}
mkdir -p $OUT
python -m pytest tests/test_ops_gpu.py -x -q -k "mr_ or reproducibility or grapher or graphconv" 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -15 | tee $OUT/tests.txt
python tools/bench_graph_path.py > $OUT/graph_path.txt 2>&1
grep -i "mr_gather" $OUT/graph_path.txt
rm -rf $OUT/trace
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python tools/bench_graph_path.py > /dev/null 2>&1
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
grep -E "mr_|Name" $f | cut -c1-200
rm -rf $OUT/trace
