"""Which ATen kernels still run inside a training step (name, input shapes, python source line)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch.profiler import profile, ProfilerActivity
from graphecho_amd.trainer import GraphEchoTrainer, synthetic_batch
dev = torch.device("cuda:0")
wl = sys.argv[1] if len(sys.argv) > 1 else "fpn_grapher"
tr = GraphEchoTrainer(dev, workload=wl, seed=0)
BS = int(sys.argv[2]) if len(sys.argv) > 2 else 32
x, m = synthetic_batch(BS, 3, 4, 256, dev, 1)
kw = {}
if wl == "full":
    xt, _ = synthetic_batch(BS, 3, 4, 256, dev, 2)
    kw = {"imgs_target": xt}
for _ in range(3):
    tr.step(x, m, **kw)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    tr.step(x, m, **kw)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=6):
    if not e.key.startswith("aten::"):
        continue
    dt = getattr(e, "self_device_time_total", 0) or getattr(e, "self_cuda_time_total", 0)
    if dt <= 0:
        continue
    stack = [s for s in e.stack if "graphecho_amd" in s or "bench" in s][:2]
    rows.append((dt, e.count, e.key, str(e.input_shapes)[:70], " <- ".join(s.split("/")[-1][:60] for s in stack)))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"ATen device time in one step: {tot/1e3:.3f} ms")
rows.sort(key=lambda r: -r[1])
for dt, n, k, shp, st in rows[:70]:
    print(f"{dt/1e3:7.3f} ms {n:4d}x {k:28s} {shp:70s} {st}")
