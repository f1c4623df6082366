#!/bin/bash
# Round-2 diagnostic: is upsample_bwd_sep slow by itself or stretched by the wgrad side stream?
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
OUT=gpurun_out/r02_diag1
mkdir -p $OUT
python bench.py --no-cpu-baseline > $OUT/bench_base.json 2> $OUT/bench_base.err
tail -1 $OUT/bench_base.json | cut -c1-300
GE_WGRAD_STREAM=0 python bench.py --no-cpu-baseline --no-kernel-timing > $OUT/bench_nostream.json 2>&1
tail -1 $OUT/bench_nostream.json | cut -c1-200
bash tools/prof_kernel_shapes.sh 'upsample|bn_|slab_reduce|channel_sum|gn_|relu|act_' > $OUT/shapes_nostream.txt 2>&1
# same with the side stream on
sed 's/GE_WGRAD_STREAM=0 rocprofv3/GE_WGRAD_STREAM=1 rocprofv3/' tools/prof_kernel_shapes.sh > /tmp/pks1.sh
bash /tmp/pks1.sh 'upsample|bn_|slab_reduce|channel_sum|gn_|relu|act_' > $OUT/shapes_stream.txt 2>&1
head -50 $OUT/shapes_nostream.txt
echo ====
head -30 $OUT/shapes_stream.txt
