#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
OUT=gpurun_out/r02_diag2
mkdir -p $OUT
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())" > $OUT/prio.txt 2>&1
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "upsample" > $OUT/pytest_upsample.txt 2>&1
tail -3 $OUT/pytest_upsample.txt
python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.json | cut -c1-250
for prio in 1 -1; do
GE_WGRAD_PRIO=$prio python bench.py --no-cpu-baseline --no-kernel-timing > $OUT/bench_prio$prio.json 2>&1
tail -1 $OUT/bench_prio$prio.json | cut -c1-200
done
GE_UPSAMPLE_BWD_STREAM=0 python bench.py --no-cpu-baseline --no-kernel-timing > $OUT/bench_oldups.json 2>&1
tail -1 $OUT/bench_oldups.json | cut -c1-200
sed 's/GE_WGRAD_STREAM=0 rocprofv3/GE_WGRAD_STREAM=1 rocprofv3/' tools/prof_kernel_shapes.sh > /tmp/pks1.sh
bash /tmp/pks1.sh 'upsample|gn_|act_' > $OUT/shapes_stream.txt 2>&1
bash tools/prof_kernel_shapes.sh 'upsample|gn_|act_' > $OUT/shapes_nostream.txt 2>&1
head -12 $OUT/shapes_stream.txt; echo ===; head -12 $OUT/shapes_nostream.txt
cat $OUT/prio.txt
