"""cProfile of the host side of a training step (which Python/launch paths keep the GPU waiting)."""
import cProfile, pstats, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphecho_amd.trainer import GraphEchoTrainer, synthetic_batch
dev = torch.device("cuda:0")
wl = sys.argv[1] if len(sys.argv) > 1 else "full"
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 32
# BB=VGG16 CIN=1 SEG=cardiac PREC=f16s: config 5 as the reference runs it
cin = int(os.environ.get("CIN", "3"))
tr = GraphEchoTrainer(dev, workload=wl, seed=0, back_bone=os.environ.get("BB", "resnet"), in_channel=cin,
                      conv_precision=os.environ.get("PREC", "f32"), seg_loss=os.environ.get("SEG", "camus"),
                      **({"clip_len": 16, "transport_method": os.environ.get("TRANSPORT", "node_discriminate")} if wl == "temporal" else {}))
x, m = synthetic_batch(bs, cin, 4, 256, dev, 1)
kw = {}
if wl in ("full", "temporal"):
    x, m = x[: bs // 2], m[: bs // 2]
    xt, _ = synthetic_batch(bs // 2, cin, 4, 256, dev, 2)
    kw = {"imgs_target": xt}
if wl == "temporal":
    def clip(seed, t=16):
        f, mk = synthetic_batch(t, cin, 4, 256, dev, seed)
        return (f.reshape(1, t, cin, 256, 256).permute(0, 2, 3, 4, 1).contiguous(),
                mk.reshape(1, t, 4, 256, 256).permute(0, 2, 3, 4, 1).contiguous())
    cs, cm = clip(77)
    ct, _ = clip(78)
    kw["clips"] = {"source": cs, "target": ct, "masks": cm}
for _ in range(5):
    tr.step(x, m, **kw)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(5):
    tr.step(x, m, **kw)
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) / 5 * 1e3)
if tr.use_graphs:
    for key, slot in tr._net.slots.items():
        for name, g in (("fwd", slot.fwd_graph), ("bwd", slot.bwd_graph)):
            if g is None:
                continue
            torch.cuda.synchronize()
            t0 = time.perf_counter(); g.replay(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
            print(f"graph {key[0]} {name}: replay() host {1e3 * (t1 - t0):.2f} ms, until done {1e3 * (t2 - t0):.2f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    tr.step(x, m, **kw)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats(os.environ.get("SORT", "cumulative")).print_stats(int(os.environ.get("TOP", "60")))
