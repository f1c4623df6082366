#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp PYTHONWARNINGS=ignore
for cfg in "384 1024" "600 1024" "1100 1024" "2100 1024" "600 600" "1100 2100" "2100 2100"; do
  set -- $cfg
  for b in 8 16; do
    r=$(GE_T128_MIN=$1 GE_T64X128_MIN=$2 python bench.py --no-cpu-baseline --no-kernel-timing --no-scaling-base --workload full --batch $b --steps 30 --warmup 8 --graphs on 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "T128_MIN=$1 T64X128_MIN=$2 full b=$b graphs: $r ms"
  done
done
