#!/bin/bash
# What one rank of config 4 does at N = 8 / 4 / 2 / 1 (global batch 64): the full workload at 8 / 16 / 32 / 64 frames per GPU.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for b in 8 16 32 64; do
echo -n "full, $b frames per GPU: "; python bench.py --no-cpu-baseline --no-kernel-timing --workload full --batch $b --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'frames/s', d['ms_per_step'], 'ms/step')"
done
echo -n "merged passes, 8 frames: "; GE_MERGE_PASSES=1 python bench.py --no-cpu-baseline --no-kernel-timing --workload full --batch 8 --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'frames/s', d['ms_per_step'], 'ms/step')"
