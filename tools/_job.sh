mkdir -p gpurun_out/h8
python bench.py --no-cpu-baseline > gpurun_out/h8/bench2.json 2> gpurun_out/h8/bench2.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/h8/bench2.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"])
for r in d.get("config5", []) if isinstance(d.get("config5"), list) else [d.get("config5")]:
    print(r.get("dtype","")[:20], r.get("value"), r.get("ms_per_step"), r if "error" in r else "")
PY
C5="--no-cpu-baseline --no-kernel-timing --workload temporal --backbone VGG16 --in-channel 1 --seg-loss cardiac --batch 16 --steps 10 --warmup 4"
python bench.py $C5 --precision f16s 2>/dev/null | tail -1 | cut -c1-200
python bench.py $C5 --precision f16s --steps 8 --warmup 8 2>/dev/null | tail -1 | cut -c1-200
