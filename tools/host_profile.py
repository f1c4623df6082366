"""cProfile of ten 4+4-frame steps (graphs="auto"): host time by function, own and cumulative."""
import os, sys, torch, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphecho_amd.trainer import GraphEchoTrainer, synthetic_batch
dev = torch.device("cuda:0")
tr = GraphEchoTrainer(dev, workload="full", seed=0, graphs="auto")
x, m = synthetic_batch(4, 3, 4, 256, dev, 1); xt, _ = synthetic_batch(4, 3, 4, 256, dev, 2)
for _ in range(8): tr.step(x, m, xt)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10): tr.step(x, m, xt)
torch.cuda.synchronize()
pr.disable()
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).strip_dirs().sort_stats(key).print_stats(38)
    print("\n".join(l[:150] for l in s.getvalue().splitlines()[4:50]))
