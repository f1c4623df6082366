#!/bin/bash
# GPU-busy estimate of a bench.py workload: sum of kernel durations per step (rocprofv3 kernel trace, side stream off so
# kernels never overlap) against the unprofiled step time of the same configuration.  usage: gpu_busy.sh [bench args]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/gpu_busy
rm -rf $OUT && mkdir -p $OUT
export GE_WGRAD_STREAM=0
python $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-kernel-timing "$@" 2>/dev/null | tail -1 > $OUT/plain.json
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-kernel-timing "$@" > $OUT/run.log 2>&1
python - <<PY
import csv, glob, json
plain = json.load(open("$OUT/plain.json"))
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:80]) for r in csv.DictReader(open(f)))
# the 12 timed steps are the tail of the trace: take kernels of the last 12/16 by count (steps launch the same kernels)
n = len(ev)
tail = ev[int(n * 0.30):]          # skip model construction + warm-up (>= 4/16 of the launches)
steps = 12 * len(tail) / (n - 0)   # rough; refine by adam_kernel count below
adam = [e for e in tail if e[2].startswith("adam_kernel")]
k = len(adam)
first, last = adam[0][0], adam[-1][0]
mid = [e for e in tail if first <= e[0] < last]
busy = sum(e[1] - e[0] for e in mid) / 1e6 / (k - 1)
print(f"plain: {plain['ms_per_step']:.2f} ms/step ({plain['value']:.1f} frames/s); kernel time {busy:.2f} ms/step over {k-1} steps "
      f"-> GPU busy {100*busy/plain['ms_per_step']:.1f} % of the unprofiled step")
agg = {}
for s, e, nme in mid:
    a = agg.setdefault(nme, [0, 0]); a[0] += e - s; a[1] += 1
for nme, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:int("${TOP:-14}")]:
    print(f"{t/1e6/(k-1):8.3f} ms/step  n/step={c/(k-1):7.1f}  {nme}")
PY
rm -f $OUT/*kernel_trace.csv
