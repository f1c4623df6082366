import os, sys, torch, traceback, collections
sys.path.insert(0, "/root/repo")
from graphecho_amd.trainer import GraphEchoTrainer, synthetic_batch
dev = torch.device("cuda:0")
tr = GraphEchoTrainer(dev, workload="full", seed=0)
x, m = synthetic_batch(4, 3, 4, 256, dev, 1)
xt, _ = synthetic_batch(4, 3, 4, 256, dev, 2)
for _ in range(3):
    tr.step(x, m, imgs_target=xt)
torch.cuda.synchronize()
from torch.utils._python_dispatch import TorchDispatchMode
cnt = collections.Counter()
class M(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__ if hasattr(func, "__name__") else str(func)
        name = str(func)
        st = traceback.extract_stack(limit=14)
        fr = [f for f in st if "graphecho_amd" in f.filename]
        where = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in fr[-2:]) if fr else "(autograd engine / torch internals)"
        cnt[(name, where)] += 1
        return func(*args, **(kwargs or {}))
with M():
    tr.step(x, m, imgs_target=xt)
torch.cuda.synchronize()
tot = sum(cnt.values())
print("aten ops dispatched in one step:", tot)
KERNELS = ("copy_", "fill_", "zero_", "add", "mul", "cat", "div", "sum", "clone", "where", "index", "neg", "sub", "stack", "norm", "dropout", "clamp", "pow", "mean", "zeros", "ones", "full", "scatter", "gather")
rows = [(k, n) for k, n in cnt.most_common() if any(("aten." + t) in k[0] for t in KERNELS)]
print("ops that launch kernels (by name):", sum(n for _, n in rows))
for (name, where), n in rows[:90]:
    print(f"{n:4d}  {name:40s} {where}")
