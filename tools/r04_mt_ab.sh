cd ${GRAFT_REPO_ROOT:-/root/repo}
python tools/host_cost_per_op.py 2>&1 | grep "backward"
for mt in 1 0 1 0; do
  for b in 8 16; do
  r=$(GE_AUTOGRAD_MT=$mt python bench.py --no-cpu-baseline --no-kernel-timing --no-scaling-base --workload full --batch $b --steps 30 --warmup 8 --graphs off 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "full b$b eager autograd_mt=$mt: $r ms"
  done
done
