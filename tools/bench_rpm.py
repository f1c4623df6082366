"""sinkhorn_rpm forward (+ backward) at the sizes of the training step: the register-resident kernel (default) vs the 41-launch chain
(GE_RPM_RESIDENT=0 in a second process), result check against an fp64 torch restatement of graph_matching.py:637-689 (slack=True)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphecho_amd import functional as GF
dev = torch.device("cuda:0")
def ref(la, n_iters):
    la = la.double()
    B, N1, N2 = la.shape
    pad = torch.zeros(B, N1 + 1, N2 + 1, dtype=la.dtype, device=la.device)
    pad[:, :N1, :N2] = la
    for _ in range(n_iters):
        pad = torch.cat((pad[:, :-1, :] - torch.logsumexp(pad[:, :-1, :], dim=2, keepdim=True), pad[:, -1:, :]), dim=1)
        pad = torch.cat((pad[:, :, :-1] - torch.logsumexp(pad[:, :, :-1], dim=1, keepdim=True), pad[:, :, -1:]), dim=2)
    return pad[:, :N1, :N2]
for (N1, N2) in [(100, 100), (270, 320), (287, 283), (240, 390), (230, 412), (300, 500)]:
    torch.manual_seed(N1 + N2)
    A = torch.randn(1, N1, N2, device=dev)
    X = GF.sinkhorn_rpm(A, 20)
    err = (X.double() - ref(A, 20)).abs().max().item()
    for _ in range(5): GF.sinkhorn_rpm(A, 20)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): GF.sinkhorn_rpm(A, 20)
    e1.record(); torch.cuda.synchronize()
    Ar = A.clone().requires_grad_(True)
    W = torch.randn_like(A)
    def fb():
        Ar.grad = None
        (GF.sinkhorn_rpm(Ar, 20) * W).sum().backward()
    for _ in range(5): fb()
    torch.cuda.synchronize()
    e2, e3 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e2.record()
    for _ in range(30): fb()
    e3.record(); torch.cuda.synchronize()
    print(f"N1={N1} N2={N2}: forward {e0.elapsed_time(e1) / 50 * 1e3:.0f} us, forward+backward {e2.elapsed_time(e3) / 30 * 1e3:.0f} us, max |err| vs fp64 {err:.2e}", flush=True)
