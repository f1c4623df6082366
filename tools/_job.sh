timeout 900 python -m pytest tests/test_half_gpu.py -q 2>&1 | tail -3
GE_H_DYNAMIC_SCALE=0 timeout 900 python -m pytest tests/test_half_gpu.py -q 2>&1 | tail -3
( time timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -4 ) 2>&1 | grep -E "passed|failed|real"
