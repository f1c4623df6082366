#!/bin/bash
# Kernel trace of the full workload at 4+4 frames per GPU (config 4 on 8 GPUs): busy time vs wall time per step.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp PYTHONWARNINGS=ignore
OUT=gpurun_out/small_batch
mkdir -p $OUT
for g in ${GRAPHS:-0 1}; do
rm -rf $OUT/trace
GE_GRAPHS=$g GE_MERGE_PASSES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py --workload full --batch ${B:-8} --steps 10 --warmup 6 --no-cpu-baseline --no-scaling-base --no-kernel-timing > $OUT/bench_g$g.json 2>/dev/null
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_g$g.csv
python - $OUT/kernel_stats_g$g.csv $OUT/bench_g$g.json <<'PY'
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
steps = 16
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6 / steps
calls = sum(int(r["Calls"]) for r in rows) / steps
print(f"ms/step wall {d['ms_per_step']}  kernel busy {tot:.2f} ms/step  launches/step {calls:.0f}")
for r in rows[:14]:
    print(f"  {r['Name'][:80]:80s} {int(r['Calls'])/steps:6.1f}/step {float(r['TotalDurationNs'])/1e6/steps:6.2f} ms/step avg {float(r['AverageNs'])/1e3:7.1f} us")
PY
done
rm -rf $OUT/trace
