#!/bin/bash
# Per-rank step of config 4 at small per-GPU batches (8 / 16 frames): bench lines, rocprofv3 kernel stats and the kernel
# trace of ONE step (name, queue, start, end) for offline critical-path analysis.  usage: small_step_profile.sh <tag> [batches]
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp PYTHONWARNINGS=ignore
TAG=${1:-small}
OUT=gpurun_out/$TAG
mkdir -p $OUT
for b in ${2:-8 16}; do
  python bench.py --no-cpu-baseline --no-kernel-timing --workload full --batch $b --steps 20 --warmup 6 $EXTRA 2>/dev/null | tail -1 > $OUT/bench_b$b.json
  python -c "import json,sys; d=json.loads(open('$OUT/bench_b$b.json').read()); print('full b=$b', d['ms_per_step'], 'ms/step')"
  rm -rf $OUT/trace
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py --workload full --batch $b --steps 8 --warmup 6 $EXTRA --no-cpu-baseline --no-scaling-base --no-kernel-timing > $OUT/bench_prof_b$b.json 2>/dev/null
  cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_b$b.csv
  f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
  python tools/trace_gaps.py $f 12
  python - $f $OUT/one_step_b$b.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(rows, key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(ks) if r["Kernel_Name"].startswith("adam_kernel")]
a, b = adam[-3], adam[-2]
t0 = int(ks[a + 1]["Start_Timestamp"])
with open(sys.argv[2], "w") as f:
    w = csv.writer(f)
    w.writerow(["name", "queue", "start_us", "end_us", "grid", "wg", "vgpr", "lds"])
    for r in ks[a + 1:b + 1]:
        w.writerow([r["Kernel_Name"][:140], r.get("Queue_Id", ""), (int(r["Start_Timestamp"]) - t0) / 1e3,
                    (int(r["End_Timestamp"]) - t0) / 1e3, r.get("Grid_Size", ""), r.get("Workgroup_Size", ""),
                    r.get("VGPR_Count", ""), r.get("LDS_Block_Size", "")])
PY
  rm -rf $OUT/trace
done
