#!/bin/bash
# The 1 / 2 / 4 / 8-GPU curve of config 4 on ONE node, both gradient-exchange modes, into one JSON:
#   bash tools/scale_sweep.sh [out.json] [extra bench.py args, e.g. --graphs]
# Each point is `bench.py --gpus N` as the driver launches it (torch.distributed.run, one rank per GPU over RCCL, strong
# scaling of global batch 64); the line's `comm` block carries world size, per-rank ms/step, bus bandwidth of the gradient
# buckets, SyncBN collective counts / latency and the exposed exchange time (step - compute-only step).
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=${1:-gpurun_out/scale_sweep.json}
shift
mkdir -p $(dirname $OUT)
NG=$(python -c "import torch; print(torch.cuda.device_count())")
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONWARNINGS=ignore
echo "[" > $OUT
first=1
for mode in allreduce rs_ag; do
  for n in 1 2 4 8; do
    [ $n -gt $NG ] && continue
    if [ $n -eq 1 ]; then
      [ $mode = rs_ag ] && continue
      line=$(python bench.py --gpus 1 --workload full --batch 64 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing "$@" 2>/dev/null | tail -1)
    else
      line=$(python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) \
             bench.py --gpus $n --steps 10 --warmup 3 --ddp-mode $mode "$@" 2>/dev/null | grep '^{' | tail -1)
    fi
    [ -z "$line" ] && line="{\"n_gpus\": $n, \"ddp_mode\": \"$mode\", \"error\": \"no JSON line\"}"
    [ $first -eq 0 ] && echo "," >> $OUT
    first=0
    echo "$line" >> $OUT
    echo "$line" | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d.get('n_gpus'), '$mode', d.get('value'), d.get('ms_per_step'), (d.get('comm') or {}).get('exposed_ms_per_step'))"
  done
done
echo "]" >> $OUT
