cd /root/repo
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_now.json
cat gpurun_out/bench_now.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline'])"
export TMPDIR=/tmp
rm -rf gpurun_out/prof_now
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_now -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
f=$(find gpurun_out/prof_now -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:40]:
    print(f"{float(r['TotalDurationNs'])/7e6:8.3f} ms/step {int(r['Calls'])/7:7.1f} calls {float(r['Percentage']):6.2f}%  {r['Name'][:110]}")
print('total ms/step', tot/7e6)
PY
find gpurun_out/prof_now -name "*kernel_trace.csv" -delete
