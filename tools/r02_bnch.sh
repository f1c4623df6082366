#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r02_bnch
export PYTHONWARNINGS=ignore
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "dist-packages\|^  File \"/usr" | tail -25 > gpurun_out/r02_bnch/pytest.txt
cat gpurun_out/r02_bnch/pytest.txt
run() { python bench.py --no-cpu-baseline --no-kernel-timing --no-scaling-base "$@" 2>gpurun_out/r02_bnch/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'frames/s', d['ms_per_step'], 'ms/step')" 2>/dev/null || grep -v "^  File\|Warning" gpurun_out/r02_bnch/err.txt | tail -5; }
for on in 0 1 0 1; do
echo -n "bn_channel=$on fpn_grapher b=32: "; GE_BN_CHANNEL=$on run --workload fpn_grapher --batch 32 --steps 20 --warmup 6
done
for on in 0 1; do
echo -n "bn_channel=$on full b=8 merge=1: "; GE_BN_CHANNEL=$on GE_MERGE_PASSES=1 run --workload full --batch 8 --steps 10 --warmup 6
echo -n "bn_channel=$on full b=8 merge=1: "; GE_BN_CHANNEL=$on GE_MERGE_PASSES=1 run --workload full --batch 8 --steps 10 --warmup 6
echo -n "bn_channel=$on full b=32: "; GE_BN_CHANNEL=$on run --workload full --batch 32 --steps 10 --warmup 6
done
for lim in 32768 65536; do
echo -n "bn_channel max=$lim fpn_grapher b=32: "; GE_BN_CHANNEL_MAX=$lim run --workload fpn_grapher --batch 32 --steps 20 --warmup 6
done
