"""Does a captured hipMemsetAsync / hipMemcpyAsync keep its place among the kernel nodes when the graph is replayed?
kernel -> memory node -> strided kernel, 200 replays over poisoned memory each.  On ROCm 7.2 / gfx950 the memset node
fails 199 of 200 replays (profiles/r05_memset_node.txt); memcpy nodes and ATen's fill / copy kernels are fine.  This is
why libgraphecho_hip.so initialises with a kernel (ge_common.h: ge_init_async)."""
import ctypes, torch
hip = ctypes.CDLL("libamdhip64.so")
dev = torch.device("cuda:0")
def trial(kind, n):
    x = torch.full((n,), 7.0, device=dev); z = torch.full((n,), 5.0, device=dev); y = torch.empty(n, device=dev)
    w = torch.randn(2048, 2048, device=dev)
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        st = torch.cuda.current_stream().cuda_stream
        t = (w * w).sum()                  # some kernel work ahead
        y.copy_(x * 3)
        if kind == "memset":
            assert hip.hipMemsetAsync(ctypes.c_void_p(x.data_ptr()), 0, ctypes.c_size_t(4 * n), ctypes.c_void_p(st)) == 0
        elif kind == "memcpy":
            assert hip.hipMemcpyAsync(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(z.data_ptr()), ctypes.c_size_t(4 * n), 3, ctypes.c_void_p(st)) == 0
        elif kind == "aten_copy":
            x.copy_(z)
        elif kind == "aten_zero":
            x.zero_()
        x[::2] += 1                        # strided kernel after the memory node
    want = {"memset": (1.0, 0.0), "aten_zero": (1.0, 0.0), "memcpy": (6.0, 5.0), "aten_copy": (6.0, 5.0)}[kind]
    bad = 0
    for it in range(200):
        x.fill_(float("nan")) if it % 2 else x.fill_(1e30)
        g.replay()
        torch.cuda.synchronize()
        ok = bool((x[::2] == want[0]).all()) and bool((x[1::2] == want[1]).all())
        bad += not ok
    print(kind, n, "bad replays:", bad, "of 200")
for kind in ("memset", "memcpy", "aten_copy", "aten_zero"):
    for n in (1 << 18, 1 << 22):
        trial(kind, n)
