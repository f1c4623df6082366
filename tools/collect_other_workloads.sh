#!/bin/bash
# Bench lines of the non-headline workloads + kernel microbenches -> gpurun_out/<tag>/ (see profiles/README.md).
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
python - "$OUT" <<'PY'
import json, subprocess, sys
out = sys.argv[1]
runs = [
    "--workload full --steps 10 --warmup 3",
    "--workload full --batch 64 --steps 10 --warmup 3 --no-scaling-base",
    # per-rank steps of config 4 on 4 / 8 GPUs: eager (--graphs off), replayed (--graphs on), the default (auto)
    "--workload full --batch 16 --steps 20 --warmup 6 --graphs off",
    "--workload full --batch 8 --steps 20 --warmup 6 --graphs off",
    "--workload full --batch 16 --steps 20 --warmup 6",
    "--workload full --batch 8 --steps 20 --warmup 6",
    "--workload temporal --batch 16 --steps 10 --warmup 3",
    "--workload temporal --batch 16 --steps 10 --warmup 3 --precision f16",
    # config 5 as train_cardiac_uda.py runs it: FPN(in_channel=1, back_bone="VGG16"), Dice + BCE over all channels
    "--workload temporal --backbone VGG16 --in-channel 1 --seg-loss cardiac --batch 16 --steps 10 --warmup 3",
    "--workload temporal --backbone VGG16 --in-channel 1 --seg-loss cardiac --batch 16 --steps 10 --warmup 3 --precision f16",
    # ... in its stated dtype: fp16 MFMA + fp16 ACTIVATION STORAGE (csrc/ge_half.hip)
    "--workload temporal --backbone VGG16 --in-channel 1 --seg-loss cardiac --batch 16 --steps 10 --warmup 8 --precision f16s",
    "--backbone VGG16 --precision f16 --steps 10 --warmup 4",
    "--backbone VGG16 --precision f16s --steps 10 --warmup 4",
    "--precision f16s --steps 20 --warmup 5",
    "--workload full --batch 8 --steps 20 --warmup 6 --graphs on",
    "--workload full --batch 16 --steps 20 --warmup 6 --graphs on",
    "--backbone VGG16 --steps 10 --warmup 3",
    "--precision f16 --steps 20 --warmup 5",
    "--batch 64 --steps 10 --warmup 3",
    "--batch 16 --steps 20 --warmup 5",
]
rows = []
for r in runs:
    p = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline"] + r.split(), capture_output=True, text=True)
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    if not line:
        rows.append({"cmd": "bench.py " + r, "error": p.stderr[-400:]})
        continue
    d = json.loads(line[-1])
    d["cmd"] = "python bench.py --no-cpu-baseline " + r
    if d.get("roofline"):
        d["roofline"].pop("per_instance", None)
    rows.append(d)
    print(d["cmd"], d["value"], d["ms_per_step"], flush=True)
json.dump(rows, open(f"{out}/other_workloads.json", "w"), indent=1)
PY
python tools/bench_graph_path.py --json $OUT/graph_path_microbench.json > /dev/null 2>&1
for m in ti s b; do python tools/bench_pvig.py --model $m --steps 10 --warmup 3 | tail -1; done > $OUT/pvig_training.jsonl
cat $OUT/pvig_training.jsonl | cut -c1-160
# the per-rank step of config 4 under the DISTRIBUTED trainer (one-rank RCCL group: communicator initialised, side streams
# probed, graphs="auto" = everything static replayed), and the launches per 8-frame step by the profiler's count
for f in 8 16; do for m in off auto on; do python tools/per_rank_step.py $f $m dist 2>&1 | grep "per-rank"; done; done > $OUT/per_rank_steps_distributed.txt
cat $OUT/per_rank_steps_distributed.txt
# ... and with every collective of the step ISSUED on that one-rank group (a one-rank group skips them by itself): default graph
# mode, the round-5 form (GE_GRAPHS_DP=partial), eager; the per-segment SyncBN launches (GE_SYNCBN_SEGS=0) beside the default
for f in 8 16; do
  python tools/per_rank_step.py $f auto dist force 2>&1 | grep "per-rank"
  GE_SYNCBN_SEGS=0 python tools/per_rank_step.py $f auto dist force 2>&1 | grep "per-rank" | sed 's/$/   <- GE_SYNCBN_SEGS=0/'
  GE_GRAPHS_DP=partial python tools/per_rank_step.py $f auto dist force 2>&1 | grep "per-rank"
  python tools/per_rank_step.py $f off dist force 2>&1 | grep "per-rank"
done > $OUT/per_rank_steps_forced.txt
cat $OUT/per_rank_steps_forced.txt
