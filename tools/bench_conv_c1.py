"""Discriminator.cls_logits (256 -> 1, 3x3): one-output-channel kernels vs the implicit-GEMM path (GE_CONV_C1=0), forward and
weight gradient, per pyramid level.  One process per setting.  usage: python tools/bench_conv_c1.py [frames]"""
import os, subprocess, sys
CHILD = r'''
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from graphecho_amd import functional as GF
dev = torch.device("cuda:0")
B = int(sys.argv[1])
ball = torch.randn(8192, 8192, device=dev)
for H in (64, 32, 16, 8):
    x = torch.randn(B, 256, H, H, device=dev, requires_grad=True); w = (torch.randn(1, 256, 3, 3, device=dev) * 0.05).requires_grad_(True)
    cache = GF.PackCache()
    y = GF.conv2d(x, w, None, 1, 1, 1, cache)
    g = torch.randn_like(y)
    def fwd(): return GF.conv2d(x, w, None, 1, 1, 1, cache)
    def bwd(): torch.autograd.grad(y, (w,), g, retain_graph=True)
    for name, f in (("fwd", fwd), ("wgrad", bwd)):
        for _ in range(3): f()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        GF.matmul(ball, ball); s.record()
        for _ in range(20): f()
        e.record(); torch.cuda.synchronize()
        print(f"{H} {name} {s.elapsed_time(e)/20*1e3:.1f}")
'''
B = sys.argv[1] if len(sys.argv) > 1 else "64"
res = {}
for name, extra in (("gemm", {"GE_CONV_C1": "0"}), ("c1", {})):
    env = dict(os.environ); env.update(extra)
    out = subprocess.run([sys.executable, "-c", CHILD, B], env=env, capture_output=True, text=True)
    if out.returncode: print(name, "failed", out.stderr[-400:])
    for line in out.stdout.splitlines():
        h, k, v = line.split()
        res.setdefault((int(h), k), {})[name] = float(v)
print(f"frames={B}: us per launch (implicit GEMM with M = 1 | one-output-channel kernel), activation bytes / time")
for (h, k), v in res.items():
    byt = int(B) * 256 * h * h * 4
    print(f"  {h:2d}x{h:<2d} {k:6s} {v.get('gemm', 0):8.1f} {v.get('c1', 0):8.1f}   {byt / 1e6 / max(v.get('c1', 1), 1e-9) * 1e6 / 1e6:6.2f} TB/s")
