cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp PYTHONWARNINGS=ignore
OUT=gpurun_out/r04_c5
mkdir -p $OUT; rm -rf $OUT/trace
GE_WGRAD_STREAM=0 GE_GM_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py --workload temporal --backbone VGG16 --in-channel 1 --seg-loss cardiac --batch 16 --steps 6 --warmup 4 --precision f16 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_c5_f16_one_stream.csv
rm -rf $OUT/trace
python - <<'PY'
import csv,re
rows=list(csv.DictReader(open('gpurun_out/r04_c5/kernel_stats_c5_f16_one_stream.csv')))
steps=10
fam={}
for r in rows:
    n=re.sub(r'<.*','',r['Name']).replace('void ','').replace('at::native::','aten:')[:44]
    f=fam.setdefault(n,[0,0.0]); f[0]+=int(r['Calls'])/steps; f[1]+=float(r['TotalDurationNs'])/steps/1e6
tot=sum(v[1] for v in fam.values()); print('kernel ms/step', round(tot,2), 'launches', round(sum(v[0] for v in fam.values())))
for k,v in sorted(fam.items(), key=lambda kv:-kv[1][1])[:28]:
    print(f"  {k:46s} {v[0]:7.1f} {v[1]:7.3f} ms  {1e3*v[1]/v[0]:7.1f} us")
PY
