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

# ---- custom autograd Functions (graphecho_amd.functional) applied during one step, by class and call site
import torch.autograd.function as _af
apps = collections.Counter()
_orig_apply = {}
from graphecho_amd import functional as _GF
for _name in dir(_GF):
    _obj = getattr(_GF, _name)
    if isinstance(_obj, type) and issubclass(_obj, torch.autograd.Function) and _obj is not torch.autograd.Function:
        def _mk(cls, orig):
            def _apply(*a, **k):
                site = "?"
                for fr in reversed(traceback.extract_stack()):
                    if "graphecho_amd" in fr.filename and "functional.py" not in fr.filename and "nn.py" not in fr.filename:
                        site = f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}"
                        break
                apps[(cls.__name__, site)] += 1
                return orig(*a, **k)
            return staticmethod(_apply)
        _orig_apply[_obj] = _obj.apply
        _obj.apply = _mk(_obj, _obj.apply)
tr.step(x, m, xt)
torch.cuda.synchronize()
print("custom Function applies in one step (forward side):", sum(apps.values()))
for (cls, site), c in apps.most_common(45):
    print(f"{c:4d} {cls:22s} {site}")
