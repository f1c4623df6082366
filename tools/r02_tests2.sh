#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
OUT=gpurun_out/r02_tests2
mkdir -p $OUT
timeout 2400 python -m pytest tests/test_models_gpu.py -q -k "ddp or bench_two or full_step_c3 or distributed" --durations=10 > $OUT/pytest.txt 2>&1
grep -n "^E  \|passed\|failed\|FAILED" $OUT/pytest.txt | cut -c1-300 | head -40
python bench.py --steps 10 --warmup 3 > $OUT/bench_n1.json 2> $OUT/bench_n1.err; tail -1 $OUT/bench_n1.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','scaling')}, d['roofline']['kernel'], d['roofline']['frac'], d.get('scaling_base'))"
tail -3 $OUT/bench_n1.err
