import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphecho_amd import functional as GF
from graphecho_amd._lib import lib, check
dev = torch.device("cuda:0")
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for (B, C, N, M) in [(32, 256, 4096, 256), (32, 256, 1024, 256), (32, 256, 256, 256), (32, 256, 64, 64)]:
    x = torch.randn(B, C, N, device=dev); y = torch.randn(B, C, M, device=dev)
    xn, yn = torch.empty_like(x), torch.empty_like(y)
    sqx, sqy = torch.empty(B, N, device=dev), torch.empty(B, M, device=dev)
    edge = torch.empty(2, B, N, 9, device=dev, dtype=torch.int64)
    tp = timeit(lambda: check(lib.ge_knn_prepare(x.data_ptr(), xn.data_ptr(), sqx.data_ptr(), B, C, N, 1, None)))
    tpy = timeit(lambda: check(lib.ge_knn_prepare(y.data_ptr(), yn.data_ptr(), sqy.data_ptr(), B, C, M, 1, None)))
    tt = timeit(lambda: check(lib.ge_knn_topk(xn.data_ptr(), sqx.data_ptr(), yn.data_ptr(), sqy.data_ptr(), None, edge.data_ptr(), B, C, N, M, 9, 1, None)))
    tf = timeit(lambda: check(lib.ge_knn_topk_fused(x.data_ptr(), xn.data_ptr(), yn.data_ptr(), sqy.data_ptr(), None, edge.data_ptr(), B, C, N, M, 9, 1, 1, None)))
    tot = timeit(lambda: GF.knn_graph(x.unsqueeze(-1), y.unsqueeze(-1), 9, 1))
    byt = 4 * B * C * (N + M) + 8 * B * N * 9
    print(f"knn B{B} C{C} N{N} M{M}: prep_x {tp:.3f} prep_y {tpy:.3f} topk {tt:.3f} fused_topk {tf:.3f} total {tot:.3f} ms | {2*B*N*M*C/tot/1e9:.1f} TFLOP/s, compulsory {byt/tot/1e6:.0f} GB/s")
