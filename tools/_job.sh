mkdir -p gpurun_out/h5
python tools/debug_half_vgg.py 256 2>&1 | grep -E "^block_[0-9]:|block_5.6" | tee gpurun_out/h5/pool_ties.txt
( time timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -8 ) 2>&1 | tee gpurun_out/h5/test_all.log
