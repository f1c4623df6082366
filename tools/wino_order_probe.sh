# Round 6: block order of wino3x3_kernel (GE_WN_ORDER 0 / 1 / 2) -- time per layer and fabric traffic (FETCH_SIZE, x2-calibrated)
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp PYTHONWARNINGS=ignore
OUT=gpurun_out/r06_wnorder; mkdir -p $OUT
for o in ${ORDERS_TIMED:-0 1 2 0 1 2}; do
  for shape in "32 256 256 64" "32 256 128 64" "32 64 64 64" "32 128 128 32" "32 256 256 32" "64 256 256 64"; do
    GE_WN_ORDER=$o python tools/bench_wino_one.py $shape 40 2>/dev/null | sed "s/^/order $o: /" >> $OUT/times.txt
  done
done
for o in 0 1 2; do
  GE_WN_ORDER=$o timeout 90 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/p$o -- python tools/bench_wino_one.py 32 256 256 64 6 > $OUT/p$o.log 2>&1
  f=$(find $OUT/p$o -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" $o <<'PY' >> $OUT/traffic.txt
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "wino3x3_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
f = sum(acc["FETCH_SIZE"]) / max(1, len(acc["FETCH_SIZE"]))
print(f"order {sys.argv[2]}: FETCH_SIZE {f:.0f} KiB-units -> read {f * 2048 / 1e6:.1f} MB (x2 calibration; one pass per counter: FETCH_SIZE + WRITE_SIZE together exceed the hardware's counter slots); written 134.2 MB; algorithmic 272.6 MB")
PY
  rm -rf $OUT/p$o
done
cat $OUT/times.txt $OUT/traffic.txt
