"""Forward conv time per small-N shape without split-K and under forced split plans (GE_SPLITK_S x GE_SPLITK_TILE), one
process per setting (the overrides are read once).  Times include the BatchNorm statistics either way: fused in the conv
epilogue (no split) or as the separate ge_bn_stats_partial pass the split path needs.  usage: python tools/bench_splitk.py"""
import os, subprocess, sys
CHILD = r'''
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from graphecho_amd import functional as GF
from graphecho_amd._lib import lib
dev = torch.device("cuda:0")
# (B, Cin, H, Cout, k): layer3 / layer4 / discriminator-p4,p5 shapes at 8, 16 and 32 frames
SH = [(8,256,16,256,3),(16,256,16,256,3),(32,256,16,256,3),(8,512,8,512,3),(16,512,8,512,3),(32,512,8,512,3),(8,256,8,256,3),(16,256,8,256,3),
      (8,1024,16,256,1),(8,256,16,1024,1),(16,1024,16,256,1),(32,1024,16,256,1),(32,256,16,1024,1),(8,2048,8,512,1),(8,512,8,2048,1),(32,2048,8,512,1),(32,512,8,2048,1),
      (8,512,32,128,1),(8,128,32,512,1),(8,128,32,128,3)]
ball = torch.randn(8192, 8192, device=dev)
for (B,Cin,H,Cout,k) in SH:
    x = torch.randn(B,Cin,H,H,device=dev); w = torch.randn(Cout,Cin,k,k,device=dev)*0.05
    cache = GF.PackCache()
    nb = lib.ge_bn_num_partials(B, H*H)
    own = torch.empty(Cout*nb*3, device=dev)
    def f():
        out = GF.conv2d(x, w, None, 1, k//2, 1, cache, True)
        if out[1] is None:
            lib.ge_bn_stats_partial(out[0].data_ptr(), own.data_ptr(), B, Cout, H*H, GF._stream())
    for _ in range(3): f()
    torch.cuda.synchronize()
    s,e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    GF.matmul(ball, ball); s.record()
    for _ in range(20): f()
    e.record(); torch.cuda.synchronize()
    print(f"{(B,Cin,H,Cout,k)} {s.elapsed_time(e)/20*1e3:.1f}")
'''
res = {}
plans = [("off", {"GE_SPLITK": "0"}), ("auto", {})]
for t in ("2", "1", "0"):
    for sp in ("2", "4", "8"):
        plans.append((f"t{t}s{sp}", {"GE_SPLITK_S": sp, "GE_SPLITK_TILE": t}))
for name, extra in plans:
    env = dict(os.environ)
    env.update(extra)
    out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    for line in out.stdout.splitlines():
        if line.startswith("("):
            k, v = line.rsplit(" ", 1)
            res.setdefault(k, {})[name] = float(v)
    if out.returncode:
        print(name, "failed:", out.stderr[-300:])
names = [n for n, _ in plans]
print(f"{'shape (B,Cin,H,Cout,k)':24s} " + " ".join(f"{n:>6s}" for n in names) + "   (us: conv forward + BatchNorm moments)")
for k, v in res.items():
    best = min(v, key=v.get)
    print(f"{k:24s} " + " ".join(f"{v.get(n, 0):6.1f}" for n in names) + f"   best={best}")
