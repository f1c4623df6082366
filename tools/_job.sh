timeout 900 python -m pytest tests/test_fullsize_properties_gpu.py -q -k "fp16_storage" 2>&1 | tail -8
