#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
OUT=gpurun_out/r02_diag3
mkdir -p $OUT
for i in 1 2; do
python bench.py --no-cpu-baseline --no-kernel-timing > $OUT/bench_a$i.json 2>&1; tail -1 $OUT/bench_a$i.json | cut -c1-120
GE_MAIN_PRIO=-1 python bench.py --no-cpu-baseline --no-kernel-timing > $OUT/bench_hp$i.json 2>&1; tail -1 $OUT/bench_hp$i.json | cut -c1-120
done
sed 's/GE_WGRAD_STREAM=0 rocprofv3/GE_MAIN_PRIO=-1 GE_WGRAD_STREAM=1 rocprofv3/' tools/prof_kernel_shapes.sh > /tmp/pks1.sh
bash /tmp/pks1.sh 'upsample|gn_|act_|wgrad' > $OUT/shapes_hp.txt 2>&1
head -14 $OUT/shapes_hp.txt
