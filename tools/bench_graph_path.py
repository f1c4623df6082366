"""Microbench of the Grapher k-NN / edge-gather / Sinkhorn / affinity / upsample kernels against the compulsory-byte
(and FLOP) figures of SURVEY.md section 8(d).  Prints one line per (op, shape): ms, GB/s of compulsory bytes, fraction
of the 8 TB/s HBM peak and -- where the op has a dense contraction -- fraction of the 157.3 TFLOP/s fp32 MFMA peak.

    python tools/bench_graph_path.py [--json out.json]
"""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphecho_amd import functional as GF

HBM, MFMA = 8.0e12, 157.3e12
dev = torch.device("cuda:0")


_ballast = None


def timeit(fn, n=20):
    """GPU time per call.  A ~10 ms ballast GEMM is queued in front of the timed region so that the host finishes
    enqueueing all n calls while the GPU is still busy: the interval between the two events then holds no launch gaps
    (small kernels called through autograd would otherwise measure the ~70 us host path, not the kernel)."""
    global _ballast
    if _ballast is None:
        _ballast = torch.randn(8192, 8192, device=dev)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    GF.matmul(_ballast, _ballast)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3


rows = []


def report(op, shape, t, byt, flops=0.0):
    r = {"op": op, "shape": shape, "ms": round(t * 1e3, 4), "GBps": round(byt / t / 1e9, 1),
         "hbm_frac": round(byt / t / HBM, 4), "roof_ms": round(max(byt / HBM, flops / MFMA) * 1e3, 4)}
    if flops:
        r["TFLOPs"] = round(flops / t / 1e12, 2)
        r["mfma_frac"] = round(flops / t / MFMA, 4)
    r["frac_of_roof"] = round(r["roof_ms"] / r["ms"], 4)
    rows.append(r)
    print(" ".join(f"{k}={v}" for k, v in r.items()), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default=None)
    ap.add_argument("--batch", type=int, default=32)
    args = ap.parse_args()
    B, C, K = args.batch, 256, 9
    torch.manual_seed(0)
    for (N, M) in [(4096, 256), (1024, 256), (256, 256), (64, 64)]:
        x = torch.randn(B, C, N, 1, device=dev)
        y = torch.randn(B, C, M, 1, device=dev)
        t = timeit(lambda: GF.knn_graph(x, y, K, 1))
        report("knn_graph(normalise+dist+topk)", f"B{B} C{C} N{N} M{M} k{K}", t, 4 * B * C * (N + M) + 8 * B * N * K,
               2.0 * B * N * M * C)
        edge = GF.knn_graph(x, y, K, 1)
        t = timeit(lambda: GF.mr_aggregate(x, edge, y))
        byt = 4 * B * C * (N + M) + 8 * B * N * K + 4 * B * 2 * C * N
        report("mr_gather_fwd", f"B{B} C{C} N{N} M{M} k{K}", t, byt)
        xg = x.clone().requires_grad_(True)
        yg = y.clone().requires_grad_(True)
        out = GF.mr_aggregate(xg, edge, yg)
        g = torch.randn_like(out)
        t = timeit(lambda: torch.autograd.grad(out, (xg, yg), g, retain_graph=True))
        report("mr_gather_bwd", f"B{B} C{C} N{N} M{M} k{K}", t, byt + B * C * N)   # + uint8 arg-max slots
    for (Bs, P, D) in [(4, 64, 256), (64, 64, 256)]:
        x = torch.rand(Bs, P, D, device=dev)
        y = torch.rand(Bs, P, D, device=dev)
        t = timeit(lambda: GF.sinkhorn_distance(x, y, 0.1, 5))
        report("sinkhorn_distance_fwd", f"B{Bs} P{P} D{D}", t, 4 * Bs * D * 2 * P + 2 * 4 * Bs * P * P)
    for N in (128, 512, 1024):
        la = torch.randn(1, N, N, device=dev)
        t = timeit(lambda: GF.sinkhorn_rpm(la, 20))
        report("sinkhorn_rpm_fwd(20 it)", f"N{N}", t, 2 * 4 * N * N)
    for (h, H) in [(8, 16), (16, 32), (32, 64), (64, 256), (8, 64), (16, 64)]:
        ch = 256 if H != 256 else 4
        a = torch.randn(B, ch, h, h, device=dev)
        lat = torch.randn(B, ch, H, H, device=dev)
        t = timeit(lambda: GF.upsample_bilinear(a, (H, H), lat))
        report("upsample_add_fwd", f"B{B} C{ch} {h}->{H}", t, 4 * B * ch * (h * h + 2 * H * H))
        ag = a.clone().requires_grad_(True)
        out = GF.upsample_bilinear(ag, (H, H), lat)
        go = torch.randn_like(out)
        t = timeit(lambda: torch.autograd.grad(out, ag, go, retain_graph=True))
        report("upsample_bwd", f"B{B} C{ch} {h}->{H}", t, 4 * B * ch * (h * h + H * H))
    for (C2, HW) in [(256, 64), (64, 128), (512, 32)]:
        xb = torch.randn(B, C2, HW, HW, device=dev)
        w = torch.ones(C2, device=dev)
        b0 = torch.zeros(C2, device=dev)
        t = timeit(lambda: GF.batch_norm(xb, w, b0, None, None, True, 0.1, 1e-5, None, True))
        report("bn_train_fwd(stats+apply+relu)", f"B{B} C{C2} {HW}x{HW}", t, 3 * 4 * B * C2 * HW * HW)
        xg = xb.clone().requires_grad_(True)
        out = GF.batch_norm(xg, w, b0, None, None, True, 0.1, 1e-5, None, True)
        go = torch.randn_like(out)
        t = timeit(lambda: torch.autograd.grad(out, xg, go, retain_graph=True))
        report("bn_train_bwd(reduce+apply)", f"B{B} C{C2} {HW}x{HW}", t, 5 * 4 * B * C2 * HW * HW)
    for (C2, HW, G) in [(256, 64, 32), (256, 32, 32), (128, 64, 128), (256, 16, 32)]:
        xb = torch.randn(B, C2, HW, HW, device=dev, requires_grad=True)
        w = torch.ones(C2, device=dev)
        b0 = torch.zeros(C2, device=dev)
        t = timeit(lambda: GF.group_norm(xb.detach(), G, w, b0, 1e-5, True))
        report("gn_fwd(+relu)", f"B{B} C{C2} {HW}x{HW} G{G}", t, 2 * 4 * B * C2 * HW * HW)
        out = GF.group_norm(xb, G, w, b0, 1e-5, True)
        go = torch.randn_like(out)
        t = timeit(lambda: torch.autograd.grad(out, xb, go, retain_graph=True))
        report("gn_bwd", f"B{B} C{C2} {HW}x{HW} G{G}", t, 4 * 4 * B * C2 * HW * HW)
    if args.json:
        with open(args.json, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
