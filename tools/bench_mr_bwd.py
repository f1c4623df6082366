"""Max-relative backward on k-NN graphs: deterministic gather (inverse lists) vs LDS-atomic scatter, on random features
(hub-heavy graphs) and on smooth feature maps (what an FPN level looks like: neighbouring nodes share neighbours)."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphecho_amd import functional as GF
dev = torch.device("cuda:0")


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


B, C, K = int(sys.argv[1]) if len(sys.argv) > 1 else 32, 256, 9
torch.manual_seed(0)
for side, r in ((64, 4), (32, 2), (16, 1), (8, 1)):
    N = side * side
    for kind in ("random", "smooth"):
        if kind == "random":
            x = torch.randn(B, C, side, side, device=dev)
        else:
            x = F.interpolate(torch.randn(B, C, side // 8 + 2, side // 8 + 2, device=dev), size=(side, side), mode="bilinear",
                              align_corners=True) + 0.05 * torch.randn(B, C, side, side, device=dev)
        y = F.avg_pool2d(x, r, r).reshape(B, C, -1, 1).contiguous() if r > 1 else None
        xn = x.reshape(B, C, N, 1).contiguous()
        edge = GF.knn_graph(xn, y, K, 1)
        deg = torch.bincount(edge[0].reshape(B, -1)[0], minlength=(N if y is None else y.shape[2]))
        row = f"N{N} M{N if y is None else y.shape[2]} {kind:6s} max in-degree {int(deg.max())} (mean {float(deg.float().mean()):.0f})"
        for det in (True, False):
            GF.MR_BWD_DETERMINISTIC = det
            xg = xn.clone().requires_grad_(True)
            yg = y.clone().requires_grad_(True) if y is not None else None
            out = GF.mr_aggregate(xg, edge, yg)
            g = torch.randn_like(out)
            ins = (xg,) if yg is None else (xg, yg)
            t = timeit(lambda: torch.autograd.grad(out, ins, g, retain_graph=True))
            row += f" | {'gather ' if det else 'scatter'} {1e3 * t:7.1f} us"
        print(row, flush=True)
GF.MR_BWD_DETERMINISTIC = True
