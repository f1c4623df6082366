"""Per-shape conv timing inside the real training step (HIP events), sorted by total time."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphecho_amd import functional as GF
from graphecho_amd.trainer import GraphEchoTrainer, synthetic_batch
dev = torch.device("cuda:0")
wl = sys.argv[1] if len(sys.argv) > 1 else "fpn_grapher"          # usage: conv_shapes.py [workload [frames per step]]
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 32
tr = GraphEchoTrainer(dev, workload=wl, seed=0)
if wl == "full":
    x, m = synthetic_batch(bs // 2, 3, 4, 256, dev, 1)
    args = (x, m, synthetic_batch(bs // 2, 3, 4, 256, dev, 2)[0])
else:
    args = synthetic_batch(bs, 3, 4, 256, dev, 1)
for _ in range(3):
    tr.step(*args)
GF.TIMER_DETAIL = True
GF.KERNEL_TIMER = GF.KernelTimer()
for _ in range(3):
    tr.step(*args)
torch.cuda.synchronize()
s = GF.KERNEL_TIMER.summary(157.3)
rows = sorted(s["per_kernel"].items(), key=lambda kv: -kv[1]["ms"])
tot = sum(v["ms"] for _, v in rows)
print("total conv ms/step", tot / 3)
import re
lost = []
for k, v in rows:
    m = re.match(r"conv_(\w+)_k(\d)s(\d)_M(\d+)_N(\d+)_K(\d+)", k)
    if m is None:
        print(f"{v['ms']/3:7.3f} ms/step {v['n']//3:3d}x  {k}")
        continue
    kind, kh, st, M, N, K = m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4)), int(m.group(5)), int(m.group(6))
    n = v["n"]
    t = v["ms"] / n * 1e-3
    # GEMM extents as the timer labels them: forward M=Cout, N=B*Ho*Wo, K=Cin/g*kh*kw; data gradient M=Cin,
    # N=B*Hi*Wi (the INPUT plane), K=Cout/g*kh*kw; weight gradient M=Cout, N=Cin/g*kh*kw, K=B*Ho*Wo.
    # A stride-s data gradient is decomposed by output parity: only 1/s^2 of the (position, tap) pairs
    # of the un-decomposed 2*M*N*K product exist, and dy has N/s^2 positions -- the roof uses the true FLOPs / bytes
    # (round 3's table divided the full product by the measured time and printed "eff" > 1 for these rows).
    s2 = st * st
    if kind == "wgrad":   # reads dy (M*K) and x (Cin*K, the strided input counted at its full size), writes dw (M*N)
        fl = 2.0 * M * N * K
        byt = (M * K + (N // (kh * kh)) * K * s2 + M * N) * 4
    elif kind == "dgrad":  # reads w (M*K), dy (Cout * N/s^2), writes dx (M*N)
        fl = 2.0 * M * N * K / s2
        byt = (M * K + (K // (kh * kh)) * (N // s2) + M * N) * 4
    else:                 # reads w (M*K), x (Cin * N*s^2), writes y (M*N)
        fl = 2.0 * M * N * K
        byt = (M * K + (K // (kh * kh)) * N * s2 + M * N) * 4
    t_m, t_h = fl / 147e12, byt / 6.0e12
    roof = max(t_m, t_h)
    lost.append((v["ms"] / 3 - roof * n / 3 * 1e3, k))
    print(f"{v['ms']/3:7.3f} ms/step {n//3:3d}x {v['tflops']:6.1f} TF  eff {roof/t:5.2f} ({'mfma' if t_m > t_h else 'hbm '}) {k}")
print("time above roofline, ms/step:", sum(l for l, _ in lost))
for l, k in sorted(lost, reverse=True)[:25]:
    print(f"  {l:6.3f}  {k}")
