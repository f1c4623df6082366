#!/bin/bash
# Evidence for the fp16-ACTIVATION-STORAGE path (config 5's stated dtype) -> gpurun_out/<tag>/ (copied to profiles/<tag>_*):
# kernel microbench, config 5 bench lines (f16s / f16 / f32), rocprofv3 kernel stats of the f16s step (weight gradients on the
# main stream: exclusive kernel times), PMC traffic / MFMA-busy of the dominant layer shape.   usage: bash tools/collect_half.sh r04
set -u
TAG=${1:-r04}
OUT=gpurun_out/${TAG}_half
mkdir -p $OUT
export TMPDIR=/tmp
python tools/bench_half.py 48 --json $OUT/half_microbench.json > /dev/null 2>&1
C5="--no-cpu-baseline --workload temporal --backbone VGG16 --in-channel 1 --seg-loss cardiac --batch 16 --steps 10 --warmup 8"
for prec in f16s f16 f32; do
  python bench.py $C5 --precision $prec 2>/dev/null | tail -1 > $OUT/c5_$prec.json
  python -c "import json; d=json.load(open('$OUT/c5_$prec.json')); r=d['roofline']; print('$prec', d['value'], d['ms_per_step'], r['kernel'], r['achieved'], r['frac'])"
done
GE_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py $C5 --precision f16s --no-kernel-timing > $OUT/c5_f16s_under_rocprof.json 2>/dev/null
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/c5_f16s_kernel_stats_one_stream.csv
rm -rf $OUT/trace
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/pmc_$name -- python tools/pmc_half.py > $OUT/pmc_half_target.txt 2>/dev/null
  python tools/pmc_summarize.py $(find $OUT/pmc_$name -name "*counter_collection.csv" | head -1) --json > $OUT/pmc_$name.json
  rm -rf $OUT/pmc_$name
done
python - "$OUT" <<'PY'
import json, sys, ast
out = sys.argv[1]
tgt = ast.literal_eval(open(f"{out}/pmc_half_target.txt").read().strip().splitlines()[-1])
f, w = json.load(open(f"{out}/pmc_FETCH_SIZE.json")), json.load(open(f"{out}/pmc_WRITE_SIZE.json"))
m = json.load(open(f"{out}/pmc_SQ_VALU_MFMA_BUSY_CYCLES.json"))
def find(d, key):
    return next((v for k, v in d.items() if key in k), None)
cal_f, cal_w = find(f, "bnh_apply"), find(w, "bnh_apply")
# FETCH_SIZE is in KiB on this rocprofv3; the calibration kernel gives the factor for 16-byte-per-lane accesses
kf = tgt["cal_bytes_read"] / (cal_f["FETCH_SIZE"] * 1024.0)
kw = tgt["cal_bytes_written"] / (cal_w["WRITE_SIZE"] * 1024.0)
rep = {"calibration": {"fetch_factor": round(kf, 3), "write_factor": round(kw, 3), "kernel": "bnh_apply_kernel, 256 MiB read + 256 MiB written"},
       "target": tgt, "kernels": {}}
for key in ("h_conv3x3_kernel", "h_wgrad3x3_kernel"):
    for name in [k for k in f if key in k]:
        row = {"n": f[name]["n"], "hbm_read_mb": round(f[name]["FETCH_SIZE"] * 1024 * kf / 1e6, 1),
               "hbm_written_mb": round(w[name]["WRITE_SIZE"] * 1024 * kw / 1e6, 1)}
        mm = m.get(name)
        if mm:
            # SQ_VALU_MFMA_BUSY_CYCLES is summed over 256 CUs x 4 SIMDs (32 cycles per 32x32x16 fp16 MFMA: the count equals
            # FLOPs / 32768 * 32 exactly); GRBM_GUI_ACTIVE is summed over the 8 XCDs, so / 8 = wall cycles of the dispatch
            wall = mm["GRBM_GUI_ACTIVE"] / 8.0
            row["mfma_busy_cycles"] = mm["SQ_VALU_MFMA_BUSY_CYCLES"]
            row["mfma_busy_cycles_expected"] = tgt["gflop"] * 1e9 / 32768.0 * 32.0
            row["wall_cycles"] = round(wall)
            row["mfma_busy_frac"] = round(mm["SQ_VALU_MFMA_BUSY_CYCLES"] / (wall * 1024.0), 3)
        rep["kernels"][name] = row
json.dump(rep, open(f"{out}/pmc_half_summary.json", "w"), indent=1)
print(json.dumps(rep, indent=1))
PY
