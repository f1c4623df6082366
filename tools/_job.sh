timeout 900 python -m pytest tests/test_half_gpu.py -q -k "saturated" 2>&1 | tail -6
