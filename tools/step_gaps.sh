#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp PYTHONWARNINGS=ignore
OUT=gpurun_out/step_gaps
mkdir -p $OUT
for g in ${GRAPHS:-0 1}; do
rm -rf $OUT/trace
GE_GRAPHS=$g GE_MERGE_PASSES=1 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python bench.py --workload full --batch ${B:-8} --steps 8 --warmup 6 --no-cpu-baseline --no-scaling-base --no-kernel-timing > $OUT/bench_g$g.json 2>/dev/null
f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
echo "== graphs=$g"; python tools/trace_gaps.py $f 14
done
rm -rf $OUT/trace
