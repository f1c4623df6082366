"""Host-side cost of one call of the common ops (issue time only: the GPU is kept behind by a ballast GEMM)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphecho_amd import functional as GF
from graphecho_amd import nn as gnn
dev = torch.device("cuda:0")
ball = torch.randn(8192, 8192, device=dev)


def host_us(fn, n=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    for _ in range(3):
        GF.matmul(ball, ball)          # ~20 ms of GPU work: the host never waits for the device below
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    return 1e6 * dt / n


x = torch.randn(4, 64, 16, 16, device=dev)
conv = gnn.Conv2d(64, 64, 3, padding=1).to(dev)
bn = gnn.BatchNorm2d(64).to(dev).train()
lin = gnn.Linear(256, 256).to(dev)
ln = gnn.LayerNorm(256).to(dev)
rows = torch.randn(300, 256, device=dev)
xg = x.clone().requires_grad_(True)
rg = rows.clone().requires_grad_(True)
print(f"aten add (no grad)            {host_us(lambda: x + x):7.1f} us")
print(f"aten add (grad)               {host_us(lambda: xg + xg):7.1f} us")
print(f"torch.empty_like              {host_us(lambda: torch.empty_like(x)):7.1f} us")
with torch.no_grad():
    print(f"GF.relu no grad               {host_us(lambda: GF.relu(x)):7.1f} us")
    print(f"conv2d module no grad         {host_us(lambda: conv(x)):7.1f} us")
    print(f"linear module no grad         {host_us(lambda: lin(rows)):7.1f} us")
    print(f"layernorm module no grad      {host_us(lambda: ln(rows)):7.1f} us")
print(f"GF.relu (grad)                {host_us(lambda: GF.relu(xg)):7.1f} us")
print(f"conv2d module (grad)          {host_us(lambda: conv(xg)):7.1f} us")
print(f"conv_bn fused pair (grad)     {host_us(lambda: gnn.conv_bn(conv, bn, xg, relu=True)):7.1f} us")
print(f"linear module (grad)          {host_us(lambda: lin(rg)):7.1f} us")
print(f"layernorm module (grad)       {host_us(lambda: ln(rg)):7.1f} us")


def fwd_bwd():
    y = lin(rg)
    y.sum().backward()


print(f"linear fwd + sum + backward   {host_us(fwd_bwd, 100):7.1f} us")


def conv_fwd_bwd():
    y = gnn.conv_bn(conv, bn, xg, relu=True)
    y.backward(x)


print(f"conv_bn fwd + backward        {host_us(conv_fwd_bwd, 100):7.1f} us")

# the autograd engine hands CUDA nodes to a per-device worker thread; Python-defined Functions then take the GIL from
# there for every node.  Same measurements with the engine in the calling thread:
with torch.autograd.set_multithreading_enabled(False):
    print(f"[single-thread engine] linear fwd + sum + backward   {host_us(fwd_bwd, 100):7.1f} us")
    print(f"[single-thread engine] conv_bn fwd + backward        {host_us(conv_fwd_bwd, 100):7.1f} us")
