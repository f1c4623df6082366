"""Soak run: N steps of a workload, watching for non-finite losses, memory growth and step-time drift.
   python tools/soak.py [full|temporal|fpn_grapher] [steps]      (SOAK_GRAPHS=auto|on|off, SOAK_FRAMES=frames per step,
   SOAK_BB=VGG16 SOAK_CIN=1 SOAK_PREC=f16s SOAK_SEG=cardiac: config 5 as the reference runs it, in its stated dtype -- also
   checks every parameter for non-finite values at each report)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphecho_amd.trainer import GraphEchoTrainer, synthetic_batch
wl = sys.argv[1] if len(sys.argv) > 1 else "full"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 150
dev = torch.device("cuda:0")
graphs = {"auto": "auto", "on": True, "off": False}[os.environ.get("SOAK_GRAPHS", "off")]
cin = int(os.environ.get("SOAK_CIN", "3"))
tr = GraphEchoTrainer(dev, workload=wl, seed=0, clip_len=16, graphs=graphs, back_bone=os.environ.get("SOAK_BB", "resnet"), in_channel=cin,
                      conv_precision=os.environ.get("SOAK_PREC", "f32"), seg_loss=os.environ.get("SOAK_SEG", "camus"),
                      **({"transport_method": "sinkhorn_distance"} if os.environ.get("SOAK_BB") == "VGG16" and wl == "temporal" else {}))
nb_env = int(os.environ.get("SOAK_FRAMES", "0"))
args = []
def batch(i):
    nb = (nb_env // 2 if nb_env else 8) if wl != "fpn_grapher" else 16
    xs, ms = synthetic_batch(nb, cin, 4, 256, dev, 1000 + i)
    if wl == "fpn_grapher":
        return [xs, ms]
    xt, _ = synthetic_batch(nb, cin, 4, 256, dev, 5000 + i)
    out = [xs, ms, xt]
    if wl == "temporal":
        def clip(seed, t=16):
            f, mk = synthetic_batch(t, cin, 4, 256, dev, seed)
            return (f.reshape(1, t, cin, 256, 256).permute(0, 2, 3, 4, 1).contiguous(),
                    mk.reshape(1, t, 4, 256, 256).permute(0, 2, 3, 4, 1).contiguous())
        cs, cm = clip(9000 + i)
        ct, _ = clip(12000 + i)
        out.append({"source": cs, "target": ct, "masks": cm})
    return out
t0 = time.time()
if os.environ.get("SOAK_EVERY") and wl == "temporal":      # what does the Sinkhorn call see, and is TGCN still finite before the step?
    real_sk = tr.sinkhorn
    def spy(x, y):
        out = real_sk(x, y)
        spy.last = (float(x.abs().max()), float(y.abs().max()), bool(torch.isfinite(x).all() and torch.isfinite(y).all()),
                    float(out[0].detach().abs().max()) if torch.isfinite(out[0]).all() else float("nan"))
        return out
    tr.sinkhorn = spy
    first_bad = []
    def mk(name):
        def hook(mod, inp, out):
            o = out[0] if isinstance(out, (tuple, list)) else out
            if torch.is_tensor(o) and not first_bad and not torch.isfinite(o).all():
                i0 = inp[0] if inp else None
                first_bad.append((name, type(mod).__name__, bool(torch.isfinite(i0).all()) if torch.is_tensor(i0) else None,
                                  float(i0.abs().max()) if torch.is_tensor(i0) and torch.isfinite(i0).all() else None, tuple(o.shape)))
        return hook
    for n, m in tr.tgcn.named_modules():
        if n:
            m.register_forward_hook(mk(n))
for i in range(steps):
    if os.environ.get("SOAK_EVERY") and wl == "temporal":
        pre = [n for n, p in tr.tgcn.named_parameters() if not torch.isfinite(p).all()]
        if pre:
            print(f"step {i}: TGCN parameters non-finite BEFORE the step: {pre[:4]}")
    loss = tr.step(*batch(i % 8))          # 8 distinct batches, fresh every step
    if os.environ.get("SOAK_EVERY") and wl == "temporal":
        print(f"step {i}: sinkhorn inputs max |x| {spy.last[0]:.3g} |y| {spy.last[1]:.3g} finite {spy.last[2]} cost {spy.last[3]:.4g}; tgcn grad max "
              f"{max(float(p.grad.abs().max()) for p in tr.tgcn.parameters() if p.grad is not None):.3g}")
    if os.environ.get("SOAK_EVERY") and not (float(loss) == float(loss)):
        print(f"step {i}: loss {float(loss)}", {k: round(float(v), 4) for k, v in tr.losses.items()}, "scale", __import__("graphecho_amd.functional", fromlist=["x"]).h_scale_value(dev))
        bad = [n for n, p in tr.network.named_parameters() if not torch.isfinite(p).all()]
        print("   non-finite params", bad[:6])
        if os.environ.get("SOAK_EVERY") and wl == "temporal":
            print("   first non-finite module output inside TGCN (name, type, input finite, input max, output shape):", first_bad)
        lt = getattr(tr, "last_temporal", None)
        if lt:
            print("   tgcn", {k: float(v) for k, v in lt["tgcn"].items()}, "clip graph", {k: float(v) for k, v in lt["graph"].items()})
        sys.exit(1)
    if (i + 1) % 25 == 0:
        torch.cuda.synchronize()
        l = float(loss)
        assert l == l and abs(l) < 1e6, f"step {i}: loss {l}"
        bad = [n for n, p in tr.network.named_parameters() if not torch.isfinite(p).all()]
        assert not bad, f"step {i}: non-finite parameters {bad[:4]}"
        print(f"step {i + 1:4d} loss {l:9.4f}  alloc {torch.cuda.memory_allocated() / 2**20:8.0f} MiB  "
              f"reserved {torch.cuda.memory_reserved() / 2**20:8.0f} MiB  {1e3 * (time.time() - t0) / 25:7.1f} ms/step", flush=True)
        t0 = time.time()
if wl != "fpn_grapher":
    sd = tr.graph_model.state_dict()
    assert torch.isfinite(sd["sr_seed"]).all() and torch.isfinite(sd["tg_seed"]).all()
print("soak ok")
