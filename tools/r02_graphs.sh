#!/bin/bash
# HIP-graph replay of the FPN passes: tests, then the full workload at the per-GPU batch sizes of config 4 with and without.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r02_graphs
export PYTHONWARNINGS=ignore
timeout 900 python -X faulthandler -m pytest tests/test_graphs_gpu.py -x -q 2>&1 | grep -v "dist-packages\|^  File \"/usr" | tail -80 > gpurun_out/r02_graphs/pytest.txt
cat gpurun_out/r02_graphs/pytest.txt
[ "$1" = "tests" ] && exit 0
run() { python bench.py --no-cpu-baseline --no-kernel-timing --no-scaling-base "$@" 2>gpurun_out/r02_graphs/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'frames/s', d['ms_per_step'], 'ms/step')" 2>/dev/null || grep -v "^  File\|Warning" gpurun_out/r02_graphs/err.txt | tail -5; }
for b in ${BATCHES:-8 16 32 64}; do
  for g in 0 1; do for mp in 0 1; do
    echo -n "full b=$b graphs=$g merge=$mp: "; GE_GRAPHS=$g GE_MERGE_PASSES=$mp run --workload full --batch $b --steps 10 --warmup 5
  done; done
done
for g in 0 1; do
  echo -n "fpn_grapher b=32 graphs=$g: "; GE_GRAPHS=$g run --workload fpn_grapher --batch 32 --steps 10 --warmup 5
  echo -n "temporal graphs=$g: "; GE_GRAPHS=$g run --workload temporal --steps 10 --warmup 5
done
