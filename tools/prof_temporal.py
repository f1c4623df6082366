"""Timing + host profile of the temporal workload (FPN on folded clips, GModule, TGCN + SinkhornDistance)."""
import cProfile, pstats, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphecho_amd.trainer import GraphEchoTrainer, synthetic_batch
dev = torch.device("cuda:0")
b, t = int(sys.argv[1]) if len(sys.argv) > 1 else 4, 8
tr = GraphEchoTrainer(dev, workload="temporal", clip_len=t, seed=0)
x, m = synthetic_batch(8, 3, 4, 256, dev, 1)
xt, _ = synthetic_batch(8, 3, 4, 256, dev, 2)
def clip(seed):
    f, mk = synthetic_batch(b // 2 * t, 3, 4, 256, dev, seed)
    f = f.reshape(b // 2, t, 3, 256, 256).permute(0, 2, 3, 4, 1).contiguous()
    mk = mk.reshape(b // 2, t, 4, 256, 256).permute(0, 2, 3, 4, 1).contiguous()
    return f, mk
cs, cm = clip(3)
ct, _ = clip(4)
clips = {"source": cs, "target": ct, "masks": cm}
for _ in range(3):
    tr.step(x, m, xt, clips)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    tr.step(x, m, xt, clips)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
frames = 16 + b * t
print(f"temporal step: {dt*1e3:.1f} ms, {frames} frames -> {frames/dt:.1f} frames/s")
pr = cProfile.Profile(); pr.enable()
for _ in range(2):
    tr.step(x, m, xt, clips)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
