timeout 900 python -m pytest tests/test_half_gpu.py -q -x -s 2>&1 | grep -E "^\[|^\{|passed|failed|Error|error|assert" | cut -c1-1200
C5="--no-cpu-baseline --no-kernel-timing --workload temporal --backbone VGG16 --in-channel 1 --seg-loss cardiac --batch 16 --steps 10 --warmup 4"
for i in 1 2; do timeout 400 python bench.py $C5 --precision f16s 2>/dev/null | tail -1 | cut -c1-200; done
