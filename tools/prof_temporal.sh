cd /root/repo
export TMPDIR=/tmp
rm -rf gpurun_out/prof_t
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_t -- python bench.py --workload temporal --batch 16 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1
f=$(find gpurun_out/prof_t -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
n=0
for r in rows:
    if 'conv_gemm' in r['Name'] or 'conv_wgrad' in r['Name']: continue
    n+=1
    if n>28: break
    print(f"{float(r['TotalDurationNs'])/7e6:8.3f} ms/step {int(r['Calls'])/7:7.1f} calls {float(r['Percentage']):6.2f}%  {r['Name'][:100]}")
print('total ms/step', tot/7e6, 'conv share', sum(float(r['TotalDurationNs']) for r in rows if 'conv_' in r['Name'])/tot)
PY
find gpurun_out/prof_t -name "*kernel_trace.csv" -delete
