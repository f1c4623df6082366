#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp PYTHONWARNINGS=ignore
OUT=gpurun_out/${1:-r04_j6}
mkdir -p $OUT
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -15 | tee $OUT/tests.txt
for det in 1 0 1 0; do
  r=$(GE_MR_BWD_DET=$det python bench.py --no-cpu-baseline --no-kernel-timing --no-scaling-base --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "C2 bs32 mr_bwd_det=$det: $r" | tee -a $OUT/ab.txt
done
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python -c "
import json; d=json.load(open('$OUT/bench.json'))
print(d['value'], d['ms_per_step'], d.get('parity'))
print([(o['frames_per_step'], o['value'], o['ms_per_step'], o.get('hip_graphs')) for o in d.get('other_configs', [])], d.get('scaling_base'))
print(d['cpu_baseline'])
"
