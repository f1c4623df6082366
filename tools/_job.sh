export TMPDIR=/tmp
mkdir -p gpurun_out/h17
timeout 600 python -m pytest tests/test_half_gpu.py -q 2>&1 | tail -2
C5="--no-cpu-baseline --workload temporal --backbone VGG16 --in-channel 1 --seg-loss cardiac --batch 16 --steps 10 --warmup 8"
GE_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/h17/trace -- python bench.py $C5 --precision f16s --no-kernel-timing > /dev/null 2>&1
cp $(find gpurun_out/h17/trace -name "*kernel_stats.csv" | head -1) gpurun_out/h17/c5_f16s_kernel_stats_one_stream.csv
rm -rf gpurun_out/h17/trace
grep -E "h_from_f32|h_conv3x3|h_to_f32" gpurun_out/h17/c5_f16s_kernel_stats_one_stream.csv | cut -c1-140
