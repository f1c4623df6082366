#!/bin/bash
# GPU busy/idle accounting of a bench.py workload from a rocprofv3 kernel trace: union of kernel intervals over the
# timed steps, and the largest idle gaps with the kernels on either side (host syncs / launch starvation show up here).
# usage: gpu_idle.sh [bench args, e.g. --workload full]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/gpu_idle
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timing "$@" > $OUT/run.log 2>&1
tail -1 $OUT/run.log | cut -c1-220
python - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:70]) for r in csv.DictReader(open(f)))
# keep the last ~55 % of the trace (the timed steps; warm-up and allocator growth are in front)
t_lo = ev[0][0] + int(0.45 * (ev[-1][1] - ev[0][0]))
ev = [e for e in ev if e[0] >= t_lo]
span = ev[-1][1] - ev[0][0]
busy, cur_s, cur_e, gaps = 0, ev[0][0], ev[0][1], []
last_name = ev[0][2]
for s, e, n in ev[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, last_name, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    if e >= cur_e:
        last_name = n
busy += cur_e - cur_s
print(f"span {span/1e6:.1f} ms, busy {busy/1e6:.1f} ms ({100*busy/span:.1f} %), idle {100-100*busy/span:.1f} %, {len(ev)} kernels")
big = sorted(gaps, reverse=True)
print(f"gaps > 20 us: {sum(1 for g in gaps if g[0] > 20000)} totalling {sum(g[0] for g in gaps if g[0] > 20000)/1e6:.2f} ms; "
      f"gaps <= 20 us: {sum(1 for g in gaps if g[0] <= 20000)} totalling {sum(g[0] for g in gaps if g[0] <= 20000)/1e6:.2f} ms")
agg = {}
for g, a, b in gaps:
    if g > 20000:
        k = (a, b)
        agg[k] = (agg.get(k, (0, 0))[0] + g, agg.get(k, (0, 0))[1] + 1)
for (a, b), (g, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f"{g/1e6:8.2f} ms  x{c:3d}  after {a:55s} before {b}")
PY
rm -f $OUT/*kernel_trace.csv
