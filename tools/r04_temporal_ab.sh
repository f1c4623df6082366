#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp PYTHONWARNINGS=ignore
run() { label=$1; shift
  r=$(env "$@" python bench.py --no-cpu-baseline --no-kernel-timing --workload temporal --batch 16 --steps 12 --warmup 4 $EXTRA 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "temporal $EXTRA $label: $r ms"; }
for EXTRA in "--precision f16" ""; do
  export EXTRA
  run default A=1
  run gm_stream_off GE_GM_STREAM=0
  run mr_scatter GE_MR_BWD_DET=0
  run gm_off_mr_scatter GE_GM_STREAM=0 GE_MR_BWD_DET=0
  run split_auto_gm_off GE_GM_STREAM=0 GE_SPLIT_BACKWARD=0
done
