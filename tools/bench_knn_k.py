import sys, torch
sys.path.insert(0, "/root/repo")
from graphecho_amd import functional as GF
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
B, C, N, M = 32, 256, 4096, 256
x = torch.randn(B, C, N, 1, device=dev); y = torch.randn(B, C, M, 1, device=dev)
for k in (1, 2, 4, 9, 16):
    print(f"k={k}: {timeit(lambda: GF.knn_graph(x, y, k, 1))*1e3:.1f} us")
