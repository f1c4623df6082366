#!/bin/bash
# What runs right before / after a given kernel in a bench.py step (to locate where small runtime copies come from).
# usage: trace_neighbours.sh <kernel name substring> [bench args]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/neigh
rm -rf $OUT && mkdir -p $OUT
PAT=$1; shift
GE_WGRAD_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing "$@" > $OUT/run.log 2>&1
python - "$PAT" <<PY
import csv, glob, sys, collections
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
ev = sorted((int(r["Start_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]) for r in csv.DictReader(open(f)))
ev = ev[len(ev) * 2 // 5:]
agg = collections.Counter()
for i, (t, n) in enumerate(ev):
    if sys.argv[1] in n and 0 < i < len(ev) - 1:
        agg[(ev[i - 1][1], ev[i + 1][1])] += 1
for (a, b), c in agg.most_common(25):
    print(f"{c/3:6.1f}/step  after {a:60s} before {b}")
PY
rm -f $OUT/*kernel_trace.csv
