#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
OUT=gpurun_out/r02_wgrad
mkdir -p $OUT
timeout 900 python -m pytest tests/test_ops_gpu.py -q -k "wgrad_3x3 or conv2d_fwd_bwd or reproducib or group_norm" > $OUT/pytest.txt 2>&1
grep -n "^E  \|passed\|failed\|FAILED" $OUT/pytest.txt | cut -c1-250 | head -30
python tools/bench_wgrad3x3.py > $OUT/bench_wgrad.txt 2>&1; cat $OUT/bench_wgrad.txt
python bench.py --no-cpu-baseline --no-scaling-base > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['kernel'], d['roofline']['frac']); [print(k, v) for k,v in list(d['roofline']['per_instance'].items())[:6]]"
