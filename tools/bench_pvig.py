"""Training throughput of the pyramid ViG classifiers on one MI355X: forward + cross-entropy +
backward + Adam on synthetic 224x224 images, plus the live conv-kernel roofline summary bench.py prints.

    python tools/bench_pvig.py [--model ti|s|m|b] [--batch 32] [--steps 20] [--warmup 5] [--json out.json]
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphecho_amd import functional as GF  # noqa: E402
from graphecho_amd.models import vig  # noqa: E402
from graphecho_amd.optim import FlatAdam  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="ti", choices=["ti", "s", "m", "b"])
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--json", default=None)
    ap.add_argument("--graphs", action="store_true", help="replay the network's forward / backward from HIP graphs "
                                                          "(graphecho_amd.graphs.GraphedModule): the tiny models are host-bound")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = getattr(vig, f"pvig_{args.model}_224_gelu")(num_classes=1000).to(dev).train()
    opt = FlatAdam(net, lr=1e-4, weight_decay=1e-4)
    x = torch.rand(args.batch, 3, 224, 224, device=dev)
    t = torch.randint(0, 1000, (args.batch,), device=dev)

    fwd = net
    if args.graphs:
        from graphecho_amd.graphs import GraphedModule

        fwd = GraphedModule(net, [opt.fp])

    def step():
        opt.zero_grad()
        GF.DIRECT_GRAD_ACCUM = True
        try:
            F.cross_entropy(fwd(x), t).backward()
        finally:
            GF.DIRECT_GRAD_ACCUM = False
        opt.step()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    GF.KERNEL_TIMER = GF.KernelTimer()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    roof = GF.KERNEL_TIMER.summary(157.3)
    flops_step = sum(r[2] for r in GF.KERNEL_TIMER.records) / 3
    GF.KERNEL_TIMER = None
    out = {"model": f"pvig_{args.model}_224_gelu", "batch": args.batch, "hip_graphs": bool(args.graphs), "ms_per_step": round(dt * 1e3, 3),
           "images_per_s": round(args.batch / dt, 1), "params_M": round(sum(p.numel() for p in net.parameters()) / 1e6, 2),
           "conv_gflop_per_step": round(flops_step / 1e9, 1),
           "whole_step_mfma_frac": round(flops_step / dt / 157.3e12, 4),
           "conv_kernels": roof["all_conv_kernels"], "per_kernel": roof["per_kernel"]}
    print(json.dumps(out))
    if args.json:
        with open(args.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
