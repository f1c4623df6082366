"""ATen launches of one 4+4-frame step by call site (TorchDispatchMode + the nearest graphecho_amd frame): where the host-bound
GModule branch still spends launches.  '?' = issued by the autograd engine (backward)."""
import os, sys, torch, traceback, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphecho_amd.trainer import GraphEchoTrainer, synthetic_batch
from torch.utils._python_dispatch import TorchDispatchMode
dev = torch.device("cuda:0")
tr = GraphEchoTrainer(dev, workload="full", seed=0, graphs="auto")
x, m = synthetic_batch(4, 3, 4, 256, dev, 1); xt, _ = synthetic_batch(4, 3, 4, 256, dev, 2)
for _ in range(8): tr.step(x, m, xt)
torch.cuda.synchronize()
LAUNCH = ("copy_", "fill_", "add", "mul", "div", "cat", "sub", "sum", "mean", "neg", "exp", "log", "pow", "where", "index", "gather", "max", "zero_", "clone",
          "native_dropout", "norm", "clamp", "abs", "rsub", "sigmoid", "eq", "ne", "lt", "gt", "ones", "zeros", "arange", "_to_copy", "stack", "sqrt", "std", "var", "softmax", "argsort", "sort", "index_select", "scatter", "masked")
cnt = collections.Counter()
class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if any(name.startswith(p) for p in LAUNCH):
            site = "?"
            for fr in reversed(traceback.extract_stack()):
                if "graphecho_amd" in fr.filename and "_prof" not in fr.filename:
                    site = f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}"
                    break
            cnt[(name, site)] += 1
        return func(*args, **(kwargs or {}))
with Mode():
    tr.step(x, m, xt)
torch.cuda.synchronize()
print("note: backward-pass ops run in the autograd thread and are not seen here")
for (name, site), c in cnt.most_common(70):
    print(f"{c:4d} {name:18s} {site}")
print("total", sum(cnt.values()))
