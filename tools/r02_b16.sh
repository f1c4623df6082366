#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export PYTHONWARNINGS=ignore
run() { python bench.py --no-cpu-baseline --no-kernel-timing --no-scaling-base "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'frames/s', d['ms_per_step'], 'ms/step')"; }
for i in 1 2; do
for sk in 1 0; do
  echo -n "b=16 splitk=$sk merge=1: "; GE_SPLITK=$sk GE_MERGE_PASSES=1 run --workload full --batch 16 --steps 10 --warmup 6
done
echo -n "b=16 merge=0: "; GE_MERGE_PASSES=0 run --workload full --batch 16 --steps 10 --warmup 6
echo -n "b=16 graphs merge=1: "; GE_GRAPHS=1 GE_MERGE_PASSES=1 run --workload full --batch 16 --steps 10 --warmup 6
done
