#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
OUT=gpurun_out/r02_merge
rm -rf $OUT; mkdir -p $OUT
for m in 1 0; do
GE_WGRAD_STREAM=0 GE_MERGE_PASSES=$m rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t$m -- python bench.py --no-cpu-baseline --no-kernel-timing --workload temporal --batch 16 --steps 6 --warmup 2 > $OUT/b$m.json 2>/dev/null
cp $(find $OUT/t$m -name "*kernel_stats.csv" | head -1) $OUT/stats_merge$m.csv; rm -rf $OUT/t$m
tail -1 $OUT/b$m.json | cut -c1-150
done
python - <<'PY'
import csv
def load(p):
    return {r['Name']: (float(r['TotalDurationNs'])/1e6/8, int(r['Calls'])/8) for r in csv.DictReader(open(p))}
a, b = load('gpurun_out/r02_merge/stats_merge1.csv'), load('gpurun_out/r02_merge/stats_merge0.csv')
print("total merged", sum(v[0] for v in a.values()), "separate", sum(v[0] for v in b.values()))
rows = sorted(set(a) | set(b), key=lambda k: -abs(a.get(k, (0, 0))[0] - b.get(k, (0, 0))[0]))
for k in rows[:25]:
    x, y = a.get(k, (0, 0)), b.get(k, (0, 0))
    print(f"{x[0]-y[0]:+7.3f} ms/step  merged {x[0]:7.3f} ({x[1]:6.1f}x)  separate {y[0]:7.3f} ({y[1]:6.1f}x)  {k[:90]}")
PY
