mkdir -p gpurun_out/h15
export TMPDIR=/tmp
python tools/bench_half.py 48 --json gpurun_out/h15/half_microbench.json 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['layer'], 'wgrad',d['h_wgrad_tflops'],'kernel',d['h_wgrad_kernel_tflops'],d['h_wgrad_kernel_us'],'reduce us',d['h_wgrad_reduce_us'])
"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/h15/pmc -- python tools/pmc_half.py > /dev/null 2>&1
python tools/pmc_summarize.py $(find gpurun_out/h15/pmc -name "*counter_collection.csv" | head -1) h_wgrad h_conv bnh_apply
rm -rf gpurun_out/h15/pmc
timeout 600 python -m pytest tests/test_half_gpu.py -q -k "conv3x3_forward_backward" 2>&1 | tail -2
