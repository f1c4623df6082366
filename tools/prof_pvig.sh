#!/bin/bash
# rocprofv3 kernel-trace of a pyramid-ViG training step (run on the GPU box): top kernels by total time.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_pvig
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT -o pvig --output-format csv -- python $R/tools/bench_pvig.py --model ${1:-ti} --steps 5 --warmup 2 > $OUT/run.log 2>&1
tail -1 $OUT/run.log
python - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot / 1e6)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    print(f'{float(r["TotalDurationNs"])/1e6:9.2f} ms {100*float(r["TotalDurationNs"])/tot:5.1f}% n={r["Calls"]:>6} avg={float(r["AverageNs"])/1e3:8.1f}us  {r["Name"][:110]}')
PY
