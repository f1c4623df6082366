#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp PYTHONWARNINGS=ignore
OUT=gpurun_out/step_gaps
mkdir -p $OUT; rm -rf $OUT/trace
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python bench.py --workload temporal --batch 16 --steps 6 --warmup 5 --no-cpu-baseline --no-scaling-base --no-kernel-timing > $OUT/bench_t.json 2>/dev/null
f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python tools/trace_gaps.py $f 16
rm -rf $OUT/trace
