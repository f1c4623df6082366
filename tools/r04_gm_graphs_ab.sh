#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp PYTHONWARNINGS=ignore
OUT=gpurun_out/${1:-r04_gm2}
mkdir -p $OUT
run() { # label, env...
  label=$1; shift
  for b in 8 16; do
    r=$(env "$@" python bench.py --no-cpu-baseline --no-kernel-timing --no-scaling-base --workload full --batch $b --steps 30 --warmup 8 --graphs 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "graphs b=$b $label: $r ms/step" | tee -a $OUT/ab.txt
  done
}
run base GE_GM_STREAM=1
run prio-1 GE_GM_STREAM=1 GE_GM_PRIORITY=-1
run hwq8 GE_GM_STREAM=1 GPU_MAX_HW_QUEUES=8
run hwq8prio GE_GM_STREAM=1 GPU_MAX_HW_QUEUES=8 GE_GM_PRIORITY=-1
run nofork GE_GM_STREAM=1 GE_GRAPH_FORK=0
for b in 8 16; do
  r=$(GE_GM_PRIORITY=-1 python bench.py --no-cpu-baseline --no-kernel-timing --no-scaling-base --workload full --batch $b --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "eager b=$b prio-1: $r ms/step" | tee -a $OUT/ab.txt
done
GE_GRAPHS=1 GE_GM_PRIORITY=-1 python tools/step_timeline.py 8 > $OUT/timeline_b8_graphs_prio.txt 2>&1
python -m pytest tests/test_models_gpu.py tests/test_graphs_gpu.py -x -q -k "phased or full or ddp or graphed or temporal or side_stream or bench" 2>&1 | tail -5 | tee $OUT/tests.txt
