"""Race / hang hunt for the co-operative sinkhorn_rpm kernels (16 workgroups meeting at a counter barrier every iteration): forward +
backward N times beside convolution and copy traffic on two other streams; every result must equal the first bit for bit.
usage: stress_rpm.py [repeats]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphecho_amd import functional as GF
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
noise_a = torch.randn(32 << 20, device=dev); noise_b = torch.empty_like(noise_a)
cx = torch.randn(32, 256, 64, 64, device=dev); cw = torch.randn(256, 256, 3, 3, device=dev) * 0.02
cache = GF.PackCache()
bad = 0
for (N1, N2) in [(270, 320), (230, 412), (100, 100), (401, 203), (640, 100)]:
    torch.manual_seed(N1)
    A = torch.randn(1, N1, N2, device=dev, requires_grad=True)
    W = torch.randn(1, N1, N2, device=dev)
    def run():
        A.grad = None
        X = GF.sinkhorn_rpm(A, 20)
        (X * W).sum().backward()
        return X.detach().clone(), A.grad.clone()
    first = run()
    diff = 0
    for i in range(N):
        with torch.cuda.stream(s1):
            noise_b.copy_(noise_a)
        with torch.cuda.stream(s2), torch.no_grad():
            GF.conv2d(cx, cw, None, 1, 1, 1, cache)
        out = run()
        if not (torch.equal(out[0], first[0]) and torch.equal(out[1], first[1])):
            diff += 1
    torch.cuda.synchronize()
    bad += diff
    print(f"N1={N1} N2={N2}: {diff} of {N} repeats differ", flush=True)
print("FAILED" if bad else "all repeats bit-identical")
sys.exit(1 if bad else 0)
