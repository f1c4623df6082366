"""Per-shape conv timing inside the real training step (HIP events), sorted by total time."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphecho_amd import functional as GF
from graphecho_amd.trainer import GraphEchoTrainer, synthetic_batch
dev = torch.device("cuda:0")
tr = GraphEchoTrainer(dev, workload="fpn_grapher", seed=0)
x, m = synthetic_batch(32, 3, 4, 256, dev, 1)
for _ in range(3):
    tr.step(x, m)
GF.TIMER_DETAIL = True
GF.KERNEL_TIMER = GF.KernelTimer()
for _ in range(3):
    tr.step(x, m)
torch.cuda.synchronize()
s = GF.KERNEL_TIMER.summary(157.3)
rows = sorted(s["per_kernel"].items(), key=lambda kv: -kv[1]["ms"])
tot = sum(v["ms"] for _, v in rows)
print("total conv ms/step", tot / 3)
for k, v in rows[:45]:
    print(f"{v['ms']/3:7.3f} ms/step {v['n']//3:3d}x {v['tflops']:6.1f} TF  {k}")
