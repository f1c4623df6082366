"""Host and device timeline of one training step: at each marker the host clock and an event on the stream; prints, per
marker, when the host passed it and when the GPU reached it (ms from the start of the step)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphecho_amd.trainer import GraphEchoTrainer, synthetic_batch
import graphecho_amd.trainer as T
dev = torch.device("cuda:0")
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
if os.environ.get("PG") == "nccl_eager":      # what does an initialised RCCL communicator do to the step?
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29612")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
wl = os.environ.get("WL", "full")                     # WL=temporal: + 2 clips x 16 frames through FPN, GModule, TGCN
# BB=VGG16 CIN=1 SEG=cardiac PREC=f16s: config 5 as the reference runs it, in its stated dtype
bb, cin, prec = os.environ.get("BB", "resnet"), int(os.environ.get("CIN", "3")), os.environ.get("PREC", "f32")
tr = GraphEchoTrainer(dev, workload=wl, seed=0, back_bone=bb, in_channel=cin, conv_precision=prec, seg_loss=os.environ.get("SEG", "camus"),
                      graphs=os.environ.get("GRAPHS", "auto") if os.environ.get("GRAPHS", "auto") == "auto" else os.environ["GRAPHS"] == "on",
                      **({"clip_len": 16, "transport_method": "sinkhorn_distance"} if wl == "temporal" else {}))
x, m = synthetic_batch(bs // 2, cin, 4, 256, dev, 1)
xt, _ = synthetic_batch(bs // 2, cin, 4, 256, dev, 2)
extra = ()
if wl == "temporal":
    def clip(seed, t=16):
        f, mk = synthetic_batch(t, cin, 4, 256, dev, seed)
        return (f.reshape(1, t, cin, 256, 256).permute(0, 2, 3, 4, 1).contiguous(),
                mk.reshape(1, t, 4, 256, 256).permute(0, 2, 3, 4, 1).contiguous())
    cs, cm = clip(77)
    ct, _ = clip(78)
    extra = ({"source": cs, "target": ct, "masks": cm},)
marks = []
def mark(name):
    ev = torch.cuda.Event(enable_timing=True); ev.record()
    marks.append((name, time.perf_counter(), ev))
def wrap(obj, attr, name, pre=False):
    f = getattr(obj, attr)
    def g(*a, **k):
        if pre: mark(name + ":begin")
        r = f(*a, **k)
        mark(name + ":end")
        return r
    setattr(obj, attr, g)
gm = tr.graph_model
for attr, name in (("_net", "fpn"), ("_pyr", "pyramid"), ("_head", "head")):
    obj = getattr(tr, attr)
    f = obj.__call__
    def g(*a, _f=obj, _n=name, **k):
        mark(_n + ":begin"); r = type(_f).__call__(_f, *a, **k); mark(_n + ":end"); return r
    # GraphedModule instances are called through type(obj).__call__: wrap with a small proxy object
    class _P:
        def __init__(self, inner, n): self.__dict__["_i"], self.__dict__["_n"] = inner, n
        def __call__(self, *a, **k):
            mark(self._n + ":begin"); r = self._i(*a, **k); mark(self._n + ":end"); return r
        def __getattr__(self, k): return getattr(self._i, k)
        def __setattr__(self, k, v): setattr(self._i, k, v)
    setattr(tr, attr, _P(obj, name))
for k in list(tr._dis):
    tr._dis[k] = _P(tr._dis[k], k)
wrap(gm, "prepare", "prepare", True)
wrap(gm.graph_generator, "label_maps", "label_maps") if hasattr(gm.graph_generator, "label_maps") else None
wrap(gm, "_forward_preprocessing_source_target", "preprocess", True)
wrap(gm, "update_seed", "update_seed", True)
wrap(gm, "_forward_cross_domain_graph", "cross", True)
wrap(gm, "_forward_aff", "aff", True)
wrap(gm, "_forward_train", "gmodule", True)
wrap(tr, "seg_loss", "seg_loss")
wrap(tr, "_backward", "backward", True)
wrap(tr, "_finish_step", "finish", True)
if wl == "temporal":
    wrap(tr, "_temporal", "temporal", True)
    wrap(tr.tgcn, "_roll", "tgcn_roll", True)
for _ in range(6):
    tr.step(x, m, xt, *extra)
torch.cuda.synchronize()
for rep in range(2):
    marks.clear()
    mark("step:begin")
    tr.step(x, m, xt, *extra)
    mark("step:end")
    torch.cuda.synchronize()
    t0, e0 = marks[0][1], marks[0][2]
    print(f"--- step {rep} (graphs={tr.use_graphs}, merge={tr.merge_passes}) ---")
    for name, t, ev in marks:
        print(f"{name:28s} host {1e3 * (t - t0):7.2f} ms   gpu {e0.elapsed_time(ev):7.2f} ms")
