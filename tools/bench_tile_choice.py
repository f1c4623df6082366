"""Forward conv time per shape under each forced tile (GE_FORCE_TILE = 0: 128x128, 1: 64x128, 2: 64x64), one process per
setting (the override is read once).  usage: python tools/bench_tile_choice.py"""
import os, subprocess, sys
CHILD = r'''
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(sys.argv[0]))) if False else "/root/repo")
from graphecho_amd import functional as GF
dev = torch.device("cuda:0")
SH = [(32,64,64,256,1),(32,256,64,64,1),(32,256,64,256,1),(32,128,32,512,1),(32,512,32,128,1),(32,256,64,512,1),(32,512,32,256,1),
      (32,1024,16,256,1),(32,256,16,1024,1),(32,2048,8,512,1),(32,512,8,2048,1),(32,64,64,64,3),(32,128,32,128,3),(32,256,16,256,3),(32,512,8,512,3)]
ball = torch.randn(8192, 8192, device=dev)
for (B,Cin,H,Cout,k) in SH:
    x = torch.randn(B,Cin,H,H,device=dev); w = torch.randn(Cout,Cin,k,k,device=dev)*0.05
    cache = GF.PackCache()
    f = lambda: GF.conv2d(x, w, None, 1, k//2, 1, cache, True)
    for _ in range(3): f()
    torch.cuda.synchronize()
    s,e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    GF.matmul(ball, ball); s.record()
    for _ in range(20): f()
    e.record(); torch.cuda.synchronize()
    print(f"{(B,Cin,H,Cout,k)} {s.elapsed_time(e)/20*1e3:.1f}")
'''
res = {}
for force in ("", "0", "1", "2"):
    env = dict(os.environ)
    if force: env["GE_FORCE_TILE"] = force
    out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True).stdout
    for line in out.splitlines():
        if line.startswith("("):
            k, v = line.rsplit(" ", 1)
            res.setdefault(k, {})[force or "auto"] = float(v)
print(f"{'shape (B,Cin,H,Cout,k)':28s} {'auto':>8s} {'128x128':>8s} {'64x128':>8s} {'64x64':>8s}   (us, forward + fused stats)")
for k, v in res.items():
    best = min(v, key=v.get)
    print(f"{k:28s} {v.get('auto',0):8.1f} {v.get('0',0):8.1f} {v.get('1',0):8.1f} {v.get('2',0):8.1f}   best={best}")
