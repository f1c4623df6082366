"""Eager trainer vs graphs="auto" trainer on the step sequence of tests/helpers/graph_cases.py
(case_graphs_auto_switches_with_the_batch_size): per step, losses, the flat gradients right before each optimizer step and
the flat parameters are compared bit for bit.  GE_DBG_ONLY=pyr|head|dis graphs only those modules; GE_DBG_KEEP=b|f|bf
keeps every torch.empty* of the backward / forward capture alive and scans them for garbage after the first replay (this
is how the captured-memset bug of round 5 was located: first bad tensor = dx of layer4.0.downsample)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphecho_amd.trainer import GraphEchoTrainer, synthetic_batch
dev = torch.device("cuda:0")
def batch(n, seed):
    (x, m), xt = synthetic_batch(n, 3, 4, 128, dev, seed), synthetic_batch(n, 3, 4, 128, dev, seed + 1)[0]
    return x, m, xt
small, big = batch(2, 40), batch(10, 50)
mode = sys.argv[1] if len(sys.argv) > 1 else "eager"
a = GraphEchoTrainer(dev, workload="full", image_size=128, seed=6)
b = GraphEchoTrainer(dev, workload="full", image_size=128, seed=6, graphs="auto" if mode == "auto" else False)
keep_mode = os.environ.get("GE_DBG_KEEP", "")
if keep_mode:
    from graphecho_amd import graphs as _G
    KEEP = []
    def _wrap_capture(name):
        orig = getattr(_G._Slot, name)
        def f(self, *a, **k):
            saved = {}
            for fn in ("empty", "empty_like", "zeros", "zeros_like", "empty_strided"):
                saved[fn] = getattr(torch, fn)
                def mk(o):
                    def g(*aa, **kk):
                        r = o(*aa, **kk)
                        KEEP.append(r)
                        return r
                    return g
                setattr(torch, fn, mk(saved[fn]))
            try:
                return orig(self, *a, **k)
            finally:
                for fn, o in saved.items():
                    setattr(torch, fn, o)
        setattr(_G._Slot, name, f)
    if "b" in keep_mode:
        _wrap_capture("_capture_backward")
    if "f" in keep_mode:
        _wrap_capture("_capture_forward")
only = os.environ.get("GE_DBG_ONLY")
if only:
    orig_set = b._set_graphs
    def _sg(on):
        orig_set(on)
        b._net.enabled = False
        b._pyr.enabled = b._pyr.enabled and "pyr" in only
        b._head.enabled = b._head.enabled and "head" in only
        for gm in b._dis.values():
            gm.enabled = gm.enabled and "dis" in only
    b._set_graphs = _sg
    _sg(b.use_graphs)
out = []
snap = {"a": {}, "b": {}}
def wrap(tr, tag):
    for n, o in tr.optimizers.items():
        orig = o.step
        def st(orig=orig, o=o, n=n):
            torch.cuda.synchronize()
            snap[tag][n] = o.fp.grad.clone()
            return orig()
        o.step = st
wrap(a, "a"); wrap(b, "b")
for s, bt in enumerate([small, small, small, small, big, small, big, small]):
    torch.manual_seed(100 + s); la = a.step(*bt).item()
    torch.manual_seed(100 + s); lb = b.step(*bt).item()
    out.append((la, lb, {k: float(v) for k, v in a.losses.items()}, {k: float(v) for k, v in b.losses.items()}))
    torch.cuda.synchronize()
    if keep_mode and s == 3:
        nbad = 0
        for i, tt in enumerate(KEEP):
            if tt.is_floating_point() and tt.numel():
                f = tt.float()
                mx = float(f.abs().max()) if torch.isfinite(f).all() else float("nan")
                if not (mx < 1e5):
                    nbad += 1
                    if nbad <= 6:
                        print("KEEP bad", i, tuple(tt.shape), tt.dtype, mx, "ctx:", [(j, tuple(KEEP[j].shape)) for j in range(max(0, i - 6), min(len(KEEP), i + 3))])
        print("KEEP total", len(KEEP), "bad", nbad)
    pd = {n: bool(torch.equal(a.optimizers[n].fp.flat, b.optimizers[n].fp.flat)) for n in a.optimizers}
    for n in a.optimizers:
        ga, gb = snap["a"].get(n), snap["b"].get(n)
        if ga is not None and not torch.equal(ga, gb):
            fp = a.optimizers[n].fp
            names = [nm for m in fp.modules for nm, _ in m.named_parameters()] if hasattr(fp, "modules") else None
            bad_p = []
            for i, (p, off) in enumerate(zip(fp.params, fp.offsets)):
                da = ga[off:off + p.numel()]; db = gb[off:off + p.numel()]
                if not torch.equal(da, db):
                    bad_p.append((i, tuple(p.shape), float((da - db).abs().max()), float(da.abs().max())))
            print("grads differ at step", s, n, len(bad_p), "of", len(fp.params), "first", bad_p[0][:3], "last", bad_p[-1][:3])
    if not all(pd.values()):
        print("params differ after step", s, [n for n, v in pd.items() if not v])
bad = [s for s, o in enumerate(out) if o[0] != o[1]]
print(mode, "first differing step:", bad[:1], [f"{o[0]:.9f}/{o[1]:.9f}" for o in out])
if bad:
    s = bad[0]
    print({k: (out[s][2][k], out[s][3].get(k)) for k in out[s][2] if out[s][2][k] != out[s][3].get(k)})
