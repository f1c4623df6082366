#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r02_direct
export PYTHONWARNINGS=ignore
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "dist-packages\|^  File \"/usr" | tail -15 > gpurun_out/r02_direct/pytest.txt
cat gpurun_out/r02_direct/pytest.txt
run() { python bench.py --no-cpu-baseline --no-kernel-timing --no-scaling-base "$@" 2>gpurun_out/r02_direct/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'frames/s', d['ms_per_step'], 'ms/step')" 2>/dev/null || grep -v "^  File\|Warning" gpurun_out/r02_direct/err.txt | tail -5; }
for b in 8 16 32; do
    echo -n "full b=$b merge=1: "; GE_MERGE_PASSES=1 run --workload full --batch $b --steps 10 --warmup 6
done
echo -n "full b=64 merge=0: "; run --workload full --batch 64 --steps 10 --warmup 6
echo -n "full b=64 merge=0 c1 off: "; GE_CONV_C1=0 run --workload full --batch 64 --steps 10 --warmup 6
echo -n "temporal: "; run --workload temporal --steps 10 --warmup 6
echo -n "fpn_grapher b=32: "; run --workload fpn_grapher --batch 32 --steps 20 --warmup 6
