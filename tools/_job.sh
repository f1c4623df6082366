mkdir -p gpurun_out/h13
timeout 900 python -m pytest tests/test_half_gpu.py -q 2>&1 | tail -5
timeout 1200 python -m pytest tests/test_models_gpu.py -q -x -s -k "c5_f16_convs_dice" 2>&1 | grep -E "Dice|passed|failed"
python tools/h_grad_range.py 2>&1 | tail -10 | tee gpurun_out/h13/grad_range_dynamic.txt
GE_H_DYNAMIC_SCALE=0 GE_H_GRAD_SCALE=4096 python tools/h_grad_range.py 2>&1 | tail -10 | tee gpurun_out/h13/grad_range_fixed4096.txt
run() { python bench.py --no-cpu-baseline --no-kernel-timing "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
C5="--workload temporal --backbone VGG16 --in-channel 1 --seg-loss cardiac --batch 16 --steps 10 --warmup 8"
echo "C5 f16s dynamic scale: $(run $C5 --precision f16s)"
