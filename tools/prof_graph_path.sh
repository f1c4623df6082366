export TMPDIR=/tmp
rm -rf gpurun_out/prof_g
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_g -- python tools/bench_graph_path.py > /dev/null 2>&1
f=$(find gpurun_out/prof_g -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:30]:
    print(f"{float(r['AverageNs'])/1e3:9.1f} us avg {int(r['Calls']):5d} calls min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f}  {r['Name'][:90]}")
PY
find gpurun_out/prof_g -name "*kernel_trace.csv" -delete
