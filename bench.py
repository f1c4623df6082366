#!/usr/bin/env python
"""Headline benchmark: training frames/sec @256x256, batch 32 per GPU (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU over RCCL; RANK / LOCAL_RANK / WORLD_SIZE from env)

A "step" is one optimisation step of the hot path on one synthetic batch already resident in HBM:
forward + losses + backward (+ gradient all-reduce) + optimizer.  Default workload = BASELINE config 2
("FPN + ViG Grapher forward/backward") at the metric's batch size 32; --workload selects the others.
Prints ONE JSON line on rank 0 with `roofline` (dominant kernel, timed live with HIP events on the launch
stream) and `cpu_baseline` (the CPU oracle timed on the host cores, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_FP16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_f16 dense peak (--precision f16 only)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="frames per GPU per step")
    ap.add_argument("--workload", default="fpn_grapher", choices=["fpn", "fpn_grapher", "full", "temporal"])
    ap.add_argument("--clip-len", type=int, default=16, help="frames per clip (temporal workload, config-5 shape)")
    ap.add_argument("--clips", type=int, default=2, help="clips per GPU and step, half source half target (temporal)")
    ap.add_argument("--transport", default="sinkhorn_distance", choices=["sinkhorn_distance", "node_discriminate"],
                    help="TGCN transport loss of the temporal workload: config 5's fp32 SinkhornDistance (default) or the "
                         "reference trainers' default node discriminator")
    ap.add_argument("--backbone", default="resnet", choices=["resnet", "VGG16"])
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--precision", default="f32", choices=["f32", "f16"],
                    help="f32 (headline): exact fp32 MFMA.  f16: BASELINE config 5's conv path -- fp16 MFMA inputs, fp32 "
                         "accumulation and storage (reported with its own dtype; never the headline number)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--probe", default=None, help=argparse.SUPPRESS)   # state_dict + frames for the oracle's logits
    ap.add_argument("--no-kernel-timing", action="store_true")
    return ap.parse_args()


def cpu_baseline_worker(args):
    """CPU oracle ("port" of the reference's algorithm in PyTorch-CPU ops) on a bounded sample of the workload."""
    from graphecho_amd.models.fpnseg import FPN
    from graphecho_amd.trainer import PyramidGraphers, synthetic_batch
    from oracle.steps import CpuTrainer

    torch.manual_seed(0)
    threads = min(os.cpu_count() or 1, 32)     # torch-CPU stops scaling (and collapses) far below 256 threads
    torch.set_num_threads(threads)
    b = 2
    net = FPN([2, 4, 23, 3], 4, 3, back_bone=args.backbone)
    gsd = None
    if args.workload != "fpn":
        s = args.size // 4
        gsd = PyramidGraphers(256, (s, s // 2, s // 4, s // 8)).state_dict()
    tr = CpuTrainer(net.state_dict(), gsd, "camus", exact_knn=False)
    x, m = synthetic_batch(b, 3, 4, args.size, "cpu", 1234)
    t0 = time.time()
    tr.step(x, m)                               # warm-up (also bounds the budget below)
    warm = time.time() - t0
    n, t0 = 0, time.time()
    while n < 1 or (n < 10 and (time.time() - t0) + warm < 20.0):
        tr.step(x, m)
        n += 1
    dt = (time.time() - t0) / n
    if args.probe:
        # Dice-vs-oracle leg: the oracle's logits for the parent's initial weights on its probe frames
        from oracle.fpn import fpn_forward
        blob = torch.load(args.probe)
        with torch.no_grad():
            ref_logits = fpn_forward({k: v.clone() for k, v in blob["state_dict"].items()}, blob["frames"], True)[0]
        torch.save(ref_logits, args.probe + ".out")
    print(json.dumps({"value": round(b / dt, 3), "unit": "frames/s", "cores": threads, "kind": "port",
                      "sample": f"{n} steps of batch {b} @{args.size}x{args.size}, FPN-{args.backbone}"
                                f"{'+Grapher' if gsd else ''} fwd+loss+bwd+Adam/SGD, torch-CPU fp32, "
                                f"{threads} threads of {os.cpu_count()} host CPUs"}), flush=True)


def probe_parity(logits, ref_logits, args, eps=1e-5):
    """Dice of the HIP path's thresholded prediction against the oracle's on the same frames and weights
    ((2TP+eps)/(2TP+FP+FN+eps), sigmoid > 0.5, train_camus_echo.py:402-417 -- SURVEY.md 8d) and the logit error."""
    p, r = logits > 0, ref_logits > 0
    tp = (p & r).sum((0, 2, 3)).double()
    fp = (p & ~r).sum((0, 2, 3)).double()
    fn = (~p & r).sum((0, 2, 3)).double()
    dice = (2 * tp + eps) / (2 * tp + fp + fn + eps)
    rel = ((logits - ref_logits).abs().max() / ref_logits.abs().max()).item()
    return {"dice_vs_oracle": round(dice.mean().item(), 6), "dice_per_class": [round(d, 6) for d in dice.tolist()],
            "pixels_differing": int((p != r).sum()), "logits_rel_err": float(f"{rel:.3e}"),
            "sample": f"2 seeded frames @{args.size}x{args.size}, FPN-{args.backbone} train-mode forward from the initial "
                      "weights, HIP vs oracle/fpn.py"}


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed PMC summary (profiles/*_traffic.json, written by
    tools/collect_profile.sh: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this same command,
    calibrated on known-byte kernels as MI355X_MICROARCH.md prescribes).  (None, reason) when there is none."""
    import glob

    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*_traffic.json")))
    if not files:
        return None, "no profiles/*_traffic.json"
    try:
        rec = json.load(open(files[-1])).get("bench_launch_average", {}).get(kernel)
    except Exception as exc:   # a malformed summary must not break the bench line
        return None, f"{os.path.basename(files[-1])}: {exc}"
    if not rec:
        return None, f"{os.path.basename(files[-1])}: kernel not sampled"
    return round(rec["hbm_bytes_per_launch"]), f"profiles/{os.path.basename(files[-1])} (rocprofv3 --pmc, offline pass)"


def cpu_baseline(args, probe=None):
    """Run the CPU baseline in a child process with a hard timeout so it can never eat the GPU budget."""
    import subprocess

    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--workload", args.workload,
           "--backbone", args.backbone, "--size", str(args.size)] + (["--probe", probe] if probe else [])
    try:
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=180)
        line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
        return json.loads(line)
    except Exception as e:  # timeout or failure: report it, never fake a number
        return {"value": None, "unit": "frames/s", "cores": min(os.cpu_count() or 1, 32), "kind": "port",
                "sample": f"cpu baseline did not finish: {type(e).__name__}"}


def main():
    args = parse()
    if args.cpu_baseline_only:
        cpu_baseline_worker(args)
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    backend = os.environ.get("GE_DIST_BACKEND", "nccl")   # "gloo": rehearsal of the N > 1 path on a 1-GPU box (tests)
    if backend != "nccl":
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from graphecho_amd import functional as GF
    from graphecho_amd.trainer import GraphEchoTrainer, synthetic_batch

    tr = GraphEchoTrainer(dev, workload=args.workload, back_bone=args.backbone, in_channel=3, num_classes=4,
                          image_size=args.size, distributed=world > 1, seed=0, conv_precision=args.precision,
                          clip_len=args.clip_len, transport_method=args.transport)
    # parity probe (rank 0, N = 1, with the CPU leg): this network's logits on two seeded frames, from the initial weights;
    # the CPU-baseline child computes the oracle's logits for the same weights and frames
    probe = None
    if world == 1 and not args.no_cpu_baseline and args.precision == "f32":
        import tempfile

        probe = {"path": os.path.join(tempfile.mkdtemp(prefix="ge_probe_"), "probe.pt")}
        sd0 = {k: v.detach().cpu().clone() for k, v in tr.network.state_dict().items()}
        px, _ = synthetic_batch(2, 3, 4, args.size, "cpu", 4242)
        with torch.no_grad():
            probe["logits"] = tr.network(px.to(dev))[0].float().cpu()
        tr.network.load_state_dict(sd0)          # undo the probe forward's running-statistics update
        torch.save({"state_dict": sd0, "frames": px}, probe["path"])
    frames_per_step = args.batch
    if args.workload == "temporal":
        # config-5 shape: a source + a target frame batch (as config 3) and `clips` clips of `clip_len` frames that go
        # through FPN (folded into the batch), GModule and TGCN + SinkhornDistance (train_camus_echo.py:232-290)
        nb, t, c = args.batch // 2, args.clip_len, args.clips
        xs, ms = synthetic_batch(nb, 3, 4, args.size, dev, 1234 + rank * 1000)
        xt, _ = synthetic_batch(nb, 3, 4, args.size, dev, 4321 + rank * 1000)

        def clip(seed):
            f, mk = synthetic_batch(c // 2 * t, 3, 4, args.size, dev, seed)
            f = f.reshape(c // 2, t, 3, args.size, args.size).permute(0, 2, 3, 4, 1).contiguous()
            mk = mk.reshape(c // 2, t, 4, args.size, args.size).permute(0, 2, 3, 4, 1).contiguous()
            return f, mk

        cs, cm = clip(77 + rank)
        ct, _ = clip(78 + rank)
        clips = {"source": cs, "target": ct, "masks": cm}
        frames_per_step = 2 * nb + c * t
        step = lambda: tr.step(xs, ms, xt, clips)
    elif args.workload == "full":
        xs, ms = synthetic_batch(args.batch // 2, 3, 4, args.size, dev, 1234 + rank * 1000)
        xt, _ = synthetic_batch(args.batch // 2, 3, 4, args.size, dev, 4321 + rank * 1000)
        step = lambda: tr.step(xs, ms, xt)
    else:
        xs, ms = synthetic_batch(args.batch, 3, 4, args.size, dev, 1234 + rank * 1000)
        step = lambda: tr.step(xs, ms)

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    # live per-kernel timing of the conv kernels (HIP events on the launch stream), a few extra steps
    roof = None
    if not args.no_kernel_timing:
        GF.KERNEL_TIMER = GF.KernelTimer()
        n_timed = min(3, args.steps)
        for _ in range(n_timed):
            step()
        torch.cuda.synchronize()
        roof = GF.KERNEL_TIMER.summary(PEAK_FP32_MFMA_TFLOPS if args.precision == "f32" else PEAK_FP16_MFMA_TFLOPS)
        if roof is not None:
            # MFMA_util of the whole step = conv FLOPs per step / wall step time / peak
            flops_step = sum(r[2] for r in GF.KERNEL_TIMER.records) / n_timed
            ach = flops_step / (elapsed / args.steps) / 1e12
            roof["whole_step"] = {"conv_gflop_per_step": round(flops_step / 1e9, 1), "achieved": round(ach, 2),
                                  "frac": round(ach / roof["peak"], 4)}
        GF.KERNEL_TIMER = None
        if roof is not None:
            roof["traffic"], roof["traffic_source"] = pmc_traffic(roof["kernel"])

    if rank == 0:
        out = {
            "metric": "training frames/sec @256x256 bs32",
            "value": round(frames_per_step * world * args.steps / elapsed, 2),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.precision == "f32" else "f16 MFMA inputs, f32 accumulate/storage (config 5 conv path)",
            "data": "synthetic",
            "config": {"workload": {"fpn": "C1-shaped: FPN-only 4-class seg",
                                    "fpn_grapher": "C2: FPN(" + args.backbone + ")+ViG Grapher fwd/bwd+Adam/SGD",
                                    "full": "C3: full GraphEcho (FPN src+tgt, GModule, 4 Discriminators)",
                                    "temporal": f"C5-shaped: full GraphEcho + temporal branch ({args.clips} clips x "
                                                f"{args.clip_len} frames through FPN, GModule, TGCN, SinkhornDistance)"}[args.workload],
                       "per_gpu_batch": frames_per_step, "global_batch": frames_per_step * world, "image": f"3x{args.size}x{args.size}",
                       "parallelism": f"dp{world}" + ("+syncbn" if world > 1 else "")},
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, probe["path"] if probe else None)
            if probe and os.path.exists(probe["path"] + ".out"):
                out["parity"] = probe_parity(probe["logits"], torch.load(probe["path"] + ".out"), args)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
