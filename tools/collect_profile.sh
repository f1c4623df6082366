#!/bin/bash
# Round profile: bench line, rocprofv3 kernel-trace stats of the same command, and the PMC passes (separate runs,
# kernel-trace only) for HBM traffic and MFMA busy of the dominant conv kernel.  Output: gpurun_out/<tag>/ .
# usage: bash tools/collect_profile.sh r01
set -u
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.json | cut -c1-400
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-scaling-base > $OUT/bench_under_rocprof.json 2>/dev/null
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
# the same command with the weight-gradient side stream off: no two kernels overlap, so every average is the kernel's own
# duration -- the figure roofline.avg_launch_ms (timed the same way) has to agree with.  With two streams the co-running
# kernels stretch each other (a weight-gradient launch shares the CUs with the data-gradient chain).
rm -rf $OUT/trace
GE_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-scaling-base > $OUT/bench_under_rocprof_one_stream.json 2>/dev/null
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_one_stream.csv
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/pmc_$name -- python tools/pmc_target.py > /dev/null 2>&1
  cp $(find $OUT/pmc_$name -name "*counter_collection.csv" | head -1) $OUT/pmc_$name.csv
  # the same counters over the bench's own launches (mixture of shapes per kernel instantiation)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/pmcb_$name -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-scaling-base > /dev/null 2>&1
  python tools/pmc_summarize.py $(find $OUT/pmcb_$name -name "*counter_collection.csv" | head -1) --json > $OUT/pmc_bench_$name.json
  rm -rf $OUT/pmcb_$name
done
python tools/pmc_report.py $OUT > $OUT/pmc_summary.txt
cat $OUT/pmc_summary.txt
# bench.py fills roofline.traffic from the newest profiles/*_traffic.json: put this round's in place and re-run the
# headline line so that the stored bench.json quotes the PMC pass made on this very build
cp $OUT/traffic.json profiles/${TAG}_traffic.json
python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.json | cut -c1-200
rm -rf $OUT/trace $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_SQ_VALU_MFMA_BUSY_CYCLES
