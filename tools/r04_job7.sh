#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp PYTHONWARNINGS=ignore
OUT=gpurun_out/${1:-r04_j7}
mkdir -p $OUT
python -m pytest tests/test_models_gpu.py -x -q -s -k "per_time_step or (forward_backward_vs_oracle and VGG16)" 2>&1 | grep -E "passed|failed|error|Error|assert|worst" | tail -15 | tee $OUT/tests.txt
for gr in 0 1 0 1; do
  r=$(GE_GRAPHER_STREAM=$gr python bench.py --no-cpu-baseline --no-kernel-timing --no-scaling-base --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "C2 bs32 grapher_stream=$gr: $r" | tee -a $OUT/ab.txt
done
