#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export PYTHONWARNINGS=ignore
run() { python bench.py --no-cpu-baseline --no-kernel-timing --no-scaling-base "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'frames/s', d['ms_per_step'], 'ms/step')"; }
for g in "" "--graphs"; do
echo -n "fpn_grapher b=32 $g: "; run --workload fpn_grapher --batch 32 --steps 20 --warmup 6 $g
echo -n "temporal b=16 $g: "; run --workload temporal --batch 16 --steps 10 --warmup 6 $g
echo -n "full b=32 $g: "; run --workload full --batch 32 --steps 10 --warmup 6 $g
done
echo -n "fpn_grapher b=32 graphs nofork: "; GE_GRAPH_FORK=0 run --workload fpn_grapher --batch 32 --steps 20 --warmup 6 --graphs
