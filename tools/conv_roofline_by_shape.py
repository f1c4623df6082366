"""Per conv shape of one training step: measured time against max(MFMA time, HBM time) -- shows which layers are
compute-bound, which are bandwidth-bound, and how far each is from its own bound.

    python tools/conv_roofline_by_shape.py [--workload fpn_grapher] [--batch 32] [--top 40]
"""
import argparse
import os
import re
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphecho_amd import functional as GF  # noqa: E402
from graphecho_amd.trainer import GraphEchoTrainer, synthetic_batch  # noqa: E402

PEAK, HBM = 157.3e12, 8.0e12


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="fpn_grapher")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--top", type=int, default=40)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    tr = GraphEchoTrainer(dev, workload=args.workload, seed=0)
    x, t = synthetic_batch(args.batch, 3, 4, 256, dev, 1234)
    for _ in range(3):
        tr.step(x, t)
    GF.TIMER_DETAIL = True
    GF.KERNEL_TIMER = GF.KernelTimer()
    n = 3
    for _ in range(n):
        tr.step(x, t)
    torch.cuda.synchronize()
    agg = {}
    for kind, name, _alg, s, e, nbytes, flops in GF.KERNEL_TIMER.records:      # flops: executed by the matrix pipe
        a = agg.setdefault(kind, [0.0, 0.0, 0, 0.0, name])
        a[0] += flops
        a[1] += s.elapsed_time(e) * 1e-3
        a[2] += 1
        a[3] += nbytes
    GF.KERNEL_TIMER = None
    tot = sum(a[1] for a in agg.values())
    bound_tot = 0.0
    rows = []
    for kind, (f, tt, c, nb, name) in agg.items():
        t_mfma, t_hbm = f / PEAK, nb / HBM
        bound = max(t_mfma, t_hbm)
        bound_tot += bound
        rows.append((tt / n, c / n, f / tt / 1e12, nb / tt / 1e12, bound / tt, "hbm" if t_hbm > t_mfma else "mfma", kind,
                     re.sub(r"conv_(gemm|wgrad)_kernel<TileCfg<", "<", name)))
    rows.sort(reverse=True)
    print(f"conv time {tot / n * 1e3:.2f} ms/step; sum of per-shape roofline bounds {bound_tot / n * 1e3:.2f} ms/step "
          f"({bound_tot / tot:.3f} of measured)")
    print(f"{'ms/step':>8} {'n':>4} {'TF':>6} {'TB/s':>5} {'of bound':>8} bound  shape / kernel")
    for r in rows[:args.top]:
        print(f"{r[0] * 1e3:8.3f} {r[1]:4.0f} {r[2]:6.1f} {r[3]:5.2f} {r[4]:8.3f} {r[5]:5s}  {r[6]}  {r[7]}")


if __name__ == "__main__":
    main()
