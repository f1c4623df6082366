"""cProfile of the host side of a training step (which Python/launch paths keep the GPU waiting)."""
import cProfile, pstats, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphecho_amd.trainer import GraphEchoTrainer, synthetic_batch
dev = torch.device("cuda:0")
wl = sys.argv[1] if len(sys.argv) > 1 else "full"
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 32
tr = GraphEchoTrainer(dev, workload=wl, seed=0)
x, m = synthetic_batch(bs, 3, 4, 256, dev, 1)
kw = {}
if wl == "full":
    x, m = x[: bs // 2], m[: bs // 2]
    xt, _ = synthetic_batch(bs // 2, 3, 4, 256, dev, 2)
    kw = {"imgs_target": xt}
for _ in range(3):
    tr.step(x, m, **kw)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(5):
    tr.step(x, m, **kw)
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) / 5 * 1e3)
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    tr.step(x, m, **kw)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(60)
