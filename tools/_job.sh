timeout 900 python -m pytest tests/test_half_gpu.py -q -k "vgg_stack_fp16_storage_vs_fp32" 2>&1 | tail -3
