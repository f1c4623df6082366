#!/bin/bash
# Round 4: where does the per-rank small-batch step (config 4 at 8 / 16 frames per GPU) spend its time?  Host vs device
# timeline (eager and --graphs), bench lines, and a ONE-STREAM rocprofv3 kernel summary (exclusive kernel times).
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp PYTHONWARNINGS=ignore
OUT=gpurun_out/${1:-r04_probe}
mkdir -p $OUT
for b in ${2:-8 16}; do
  for mode in eager graphs; do
    flag=""; [ $mode = graphs ] && flag="--graphs"
    python bench.py --no-cpu-baseline --no-kernel-timing --no-scaling-base --workload full --batch $b --steps 30 --warmup 8 $flag 2>/dev/null | tail -1 > $OUT/bench_b${b}_$mode.json
    python -c "import json; d=json.loads(open('$OUT/bench_b${b}_$mode.json').read()); print('full b=$b $mode', d['ms_per_step'], 'ms/step')"
  done
  python tools/step_timeline.py $b > $OUT/timeline_b${b}_eager.txt 2>&1
  GE_GRAPHS=1 python tools/step_timeline.py $b > $OUT/timeline_b${b}_graphs.txt 2>&1
  GE_SPLIT_BACKWARD=1 python tools/step_timeline.py $b > $OUT/timeline_b${b}_phased.txt 2>&1
  rm -rf $OUT/trace
  GE_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py --workload full --batch $b --steps 8 --warmup 6 --no-cpu-baseline --no-scaling-base --no-kernel-timing > /dev/null 2>&1
  cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_one_stream_b$b.csv
  rm -rf $OUT/trace
done
python tools/prof_aten.py full 4 > $OUT/aten_b8.txt 2>&1 || true
