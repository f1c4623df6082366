#!/bin/bash
# Per-(kernel, grid) durations of selected kernels in one bench.py run without the wgrad side stream (so no two
# kernels overlap and every duration is the kernel's own).  usage: prof_kernel_shapes.sh "<name regex>" [bench args]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_shapes
rm -rf $OUT && mkdir -p $OUT
PAT=${1:-bn_bwd}
shift
# PROF_CMD overrides the profiled command (default: 6 steps of bench.py); the ms/step column then reads "ms per 6 runs".
CMD=${PROF_CMD:-"python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing $*"}
GE_WGRAD_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- $CMD > $OUT/run.log 2>&1
python - "$PAT" <<PY
import csv, glob, re, sys, collections
pat = re.compile(sys.argv[1])
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if not pat.search(n):
        continue
    key = (n.split("(")[0][:60], r.get("Grid_Size") or r.get("Grid_Size_X"), r.get("Workgroup_Size") or r.get("Workgroup_Size_X"))
    agg[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in agg.values())
print(f"total {tot/6e3:.3f} ms/step over 6 steps")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{sum(v)/6e3:8.3f} ms/step  n/step={len(v)/6:6.1f}  avg={sum(v)/len(v):8.1f}us  min={min(v):8.1f}us  grid={k[1]} wg={k[2]}  {k[0]}")
PY
rm -f $OUT/*kernel_trace.csv $OUT/*/*kernel_trace.csv
