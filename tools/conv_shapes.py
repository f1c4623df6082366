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
    fl = 2.0 * M * N * K
    # compulsory bytes of the implicit GEMM: both operands once + the result (3x3 input counted once, not 9x)
    a_b = M * K * 4
    b_b = (K // (kh * kh) if kind != "wgrad" else K) * N * 4 if kind != "wgrad" else (K * N // (kh * kh) if False else K * N * 4)
    if kind == "wgrad":   # M=Cout, N=Cin*kh*kw, K=B*Ho*Wo: reads dy (M*K) and x (Cin*K), writes dw (M*N)
        byt = (M * K + (N // (kh * kh)) * K + M * N) * 4
    else:                 # reads w (M*K), act (K/(kh*kw) channels * N), writes M*N
        byt = (M * K + (K // (kh * kh)) * N * (st * st if kind == "fwd" else 1) + M * N) * 4
    t_m, t_h = fl / 147e12, byt / 6.0e12
    roof = max(t_m, t_h)
    lost.append((v["ms"] / 3 - roof * n / 3 * 1e3, k))
    print(f"{v['ms']/3:7.3f} ms/step {n//3:3d}x {v['tflops']:6.1f} TF  eff {roof/t:5.2f} ({'mfma' if t_m > t_h else 'hbm '}) {k}")
print("time above roofline, ms/step:", sum(l for l, _ in lost))
for l, k in sorted(lost, reverse=True)[:25]:
    print(f"  {l:6.3f}  {k}")
