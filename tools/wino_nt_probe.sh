# Round 6: cache policy of the Winograd kernel's patch loads / output stores (variants nt00 / nt10 / nt01 / nt11 built by
# tools/build_wino_variants.sh) x block order 0 / 2: time on three layers and fabric reads (FETCH_SIZE, x2-calibrated) on 256->256@64x64x32
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp PYTHONWARNINGS=ignore
OUT=gpurun_out/r06_wnnt; mkdir -p $OUT; rm -f $OUT/*.txt
for rep in 1 2; do
for v in nt00 nt10 nt01 nt11; do
  for o in 0 2; do
    for shape in "32 256 256 64" "32 256 128 64" "32 256 256 32" "8 256 256 64"; do
      GE_LIB_PATH=graphecho_amd/csrc/variants/lib_$v.so GE_WN_ORDER=$o python tools/bench_wino_one.py $shape 40 2>/dev/null | sed "s/^/order $o /" >> $OUT/times.txt
    done
  done
done
done
for v in nt00 nt10 nt01 nt11; do
  for o in 0 2; do
    for c in FETCH_SIZE WRITE_SIZE; do
    GE_LIB_PATH=graphecho_amd/csrc/variants/lib_$v.so GE_WN_ORDER=$o timeout 90 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/p -- python tools/bench_wino_one.py 32 256 256 64 6 > $OUT/p.log 2>&1
    f=$(find $OUT/p -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python - "$f" $v $o $c <<'PY' >> $OUT/traffic.txt
import csv, sys, collections
acc = []
for r in csv.DictReader(open(sys.argv[1])):
    if "wino3x3_kernel" in r["Kernel_Name"] and r["Counter_Name"] == sys.argv[4]:
        acc.append(float(r["Counter_Value"]))
f = sum(acc) / max(1, len(acc))
mb = f * (2048 if sys.argv[4] == "FETCH_SIZE" else 1024) / 1e6
print(f"{sys.argv[2]} order {sys.argv[3]}: {sys.argv[4]} {f:.0f} -> {mb:.1f} MB")
PY
    rm -rf $OUT/p
    done
  done
done
cat $OUT/times.txt $OUT/traffic.txt
