#!/bin/bash
# A/B of GModule on its own stream (GE_GM_STREAM) for the full workload at 8 / 16 / 32 frames, eager and --graphs.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp PYTHONWARNINGS=ignore
OUT=gpurun_out/${1:-r04_gm}
mkdir -p $OUT
python -m pytest tests/test_models_gpu.py tests/test_graphs_gpu.py -x -q -k "phased or full or ddp or graphed or temporal or side_stream" 2>&1 | tail -5
for b in ${2:-8 16 32}; do
  for gm in 0 1; do
    for mode in eager graphs; do
      flag=""; [ $mode = graphs ] && flag="--graphs"
      for sb in auto 1; do
        [ $mode = graphs ] && [ $sb = 1 ] && continue
        r=$(GE_SPLIT_BACKWARD=$sb GE_GM_STREAM=$gm python bench.py --no-cpu-baseline --no-kernel-timing --no-scaling-base --workload full --batch $b --steps 30 --warmup 8 $flag 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
        echo "full b=$b gm_stream=$gm $mode split_backward=$sb: $r ms/step" | tee -a $OUT/ab.txt
      done
    done
  done
done
GE_GRAPHS=1 python tools/step_timeline.py 8 > $OUT/timeline_b8_graphs.txt 2>&1
GE_SPLIT_BACKWARD=1 python tools/step_timeline.py 16 > $OUT/timeline_b16_phased.txt 2>&1
