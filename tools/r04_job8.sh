#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp PYTHONWARNINGS=ignore
OUT=gpurun_out/${1:-r04_j8}
mkdir -p $OUT
python -m pytest tests/test_models_gpu.py tests/test_graphs_gpu.py -x -q -k "temporal or ddp or phased or full or graph or c5 or config5 or distributed" 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -12 | tee $OUT/tests.txt
for gm in 1 0 1 0; do
  for cfg in "" "--backbone VGG16 --in-channel 1 --seg-loss cardiac" "--backbone VGG16 --in-channel 1 --seg-loss cardiac --precision f16"; do
    r=$(GE_GM_STREAM=$gm python bench.py --no-cpu-baseline --no-kernel-timing --workload temporal --batch 16 --steps 12 --warmup 4 $cfg 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "temporal [$cfg] gm_stream=$gm: $r ms" | tee -a $OUT/ab.txt
  done
done
WL=temporal python tools/step_timeline.py 16 2>&1 | tail -62 > $OUT/timeline_temporal.txt
