#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp PYTHONWARNINGS=ignore
OUT=gpurun_out/${1:-r04_j3}
mkdir -p $OUT
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -15 | tee $OUT/tests.txt
for b in 8 16 32; do
  for mode in eager graphs; do
    flag=""; [ $mode = graphs ] && flag="--graphs"
    r=$(python bench.py --no-cpu-baseline --no-kernel-timing --no-scaling-base --workload full --batch $b --steps 30 --warmup 8 $flag 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "full b=$b $mode: $r ms/step" | tee -a $OUT/ab.txt
  done
done
for gm in 0 1; do
  r=$(GE_GM_STREAM=$gm python bench.py --no-cpu-baseline --no-kernel-timing --no-scaling-base --workload temporal --steps 12 --warmup 4 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "temporal gm_stream=$gm: $r ms/step" | tee -a $OUT/ab.txt
done
GE_GRAPHS=1 python tools/step_timeline.py 8 > $OUT/timeline_b8_graphs.txt 2>&1
python tools/step_timeline.py 16 > $OUT/timeline_b16_eager.txt 2>&1
SORT=tottime TOP=45 python tools/prof_host.py full 8 > $OUT/prof_host_b8.txt 2>&1
