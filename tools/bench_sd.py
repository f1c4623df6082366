"""SinkhornDistance forward kernels in isolation (preallocated buffers, direct library calls): one-launch form vs the
cost / iterate / finalize launches, with the iteration count and the feature width varied to separate the phases."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphecho_amd._lib import lib, check
dev = torch.device("cuda:0")
p = lambda t: t.data_ptr()
def run(B, P, D, T, fused, n=50):
    x, y = torch.rand(B, P, D, device=dev), torch.rand(B, P, D, device=dev)
    Cm, pi = torch.empty(B, P, P, device=dev), torch.empty(B, P, P, device=dev)
    cost, nits = torch.empty(B, device=dev), torch.empty(1, device=dev, dtype=torch.int32)
    uh, vh, err = torch.empty(B, T + 1, P, device=dev), torch.empty(B, T + 1, P, device=dev), torch.empty(B, T, device=dev)
    sync = torch.zeros(2, device=dev, dtype=torch.int32)
    st = torch.cuda.current_stream().cuda_stream
    def call():
        if fused:
            check(lib.ge_sinkhorn_distance_fwd_fused(p(x), p(y), p(Cm), p(pi), p(cost), p(nits), p(uh), p(vh), p(err), p(sync), B, P, P, D, 0.1, T, 0.1, st))
        else:
            check(lib.ge_sinkhorn_distance_fwd(p(x), p(y), p(Cm), p(pi), p(cost), p(nits), p(uh), p(vh), p(err), B, P, P, D, 0.1, T, 0.1, st))
    for _ in range(5): call()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): call()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for B in (1, 4, 64):
    for (D, T) in ((256, 5), (256, 1), (64, 5), (64, 1)):
        print(f"B{B} P64 D{D} T{T}: fused {run(B, 64, D, T, True):6.1f} us   3-launch {run(B, 64, D, T, False):6.1f} us")
