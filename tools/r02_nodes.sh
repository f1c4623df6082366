#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r02_nodes
export PYTHONWARNINGS=ignore
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_graphs_gpu.py tests/test_models_gpu.py -x -q -k "front_end or graphed or full_step or full_workload or gmodule or distributed or config5 or edge or hallucin or fewer or train_loop or ddp_world2" 2>&1 | grep -v "dist-packages\|^  File \"/usr" | tail -40 > gpurun_out/r02_nodes/pytest.txt
cat gpurun_out/r02_nodes/pytest.txt
[ "$1" = "tests" ] && exit 0
run() { python bench.py --no-cpu-baseline --no-kernel-timing --no-scaling-base "$@" 2>gpurun_out/r02_nodes/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'frames/s', d['ms_per_step'], 'ms/step')" 2>/dev/null || grep -v "^  File\|Warning" gpurun_out/r02_nodes/err.txt | tail -5; }
for b in ${BATCHES:-8 16 64}; do
  for g in 0 1; do for mp in ${MERGES:-0 1}; do
    echo -n "full b=$b graphs=$g merge=$mp: "; GE_GRAPHS=$g GE_MERGE_PASSES=$mp run --workload full --batch $b --steps 10 --warmup 6
  done; done
done
for g in 0 1; do
  echo -n "temporal graphs=$g: "; GE_GRAPHS=$g run --workload temporal --steps 10 --warmup 6
done
GE_GRAPHS=1 GE_MERGE_PASSES=1 python tools/step_timeline.py 8 2>/dev/null | tail -22
