#!/usr/bin/env python
"""Headline benchmark: training frames/sec @256x256 (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU over RCCL; RANK / LOCAL_RANK / WORLD_SIZE from env)

A "step" is one optimisation step of the hot path on one synthetic batch already resident in HBM:
forward + losses + backward (+ gradient exchange) + optimizer.

  N = 1 (default): BASELINE config 2 ("FPN + ViG Grapher forward/backward") at the metric's batch size 32, with
      `roofline` (dominant kernel, timed live with HIP events on the launch stream), `cpu_baseline` (the CPU oracle
      timed on the host cores), `other_configs` (config 3 at 8 + 8 and 16 + 16 frames), `config5` (config 5 as the reference
      runs it, fp32 and in its stated dtype: fp16 MFMA conv path + fp16 activation storage) and `scaling_base` (config 4's
      workload on this one GPU: the N = 1 point of its curve).
  N > 1 (default): BASELINE config 4 -- full GraphEcho (FPN on source + target frames, GModule, 4 Discriminators) under
      data parallelism with SyncBN, GLOBAL batch 64 split 64 / N per rank, half source half target (strong scaling,
      train_camus_echo.py:129-142), with `comm` (gradient-exchange bus bandwidth, SyncBN collectives per step).
  --workload / --batch / --scaling weak select anything else (weak: --batch frames per GPU whatever N is).

`roofline.achieved` / `frac` count the FLOPs the matrix pipe EXECUTES over the dominant kernel's time: for a direct convolution
kernel that is SURVEY 8d's algorithmic figure (2 M N K); for the Winograd kernels of the large fp32 3x3 layers (ge_wino.hip) it is
16 / 36 of it, and the same launches' rate in algorithmic FLOPs is reported beside it (`algorithmic_tflops`,
`algorithmic_over_peak` -- above 1 by construction, never called a fraction of the roof).
"""
import argparse
import json
import os
import sys
import time

# the host driver of these nodes only supports dmabuf IPC: RCCL / cross-process buffer sharing needs this before the HIP
# runtime starts (already exported on the boxes; kept here so a bare `torchrun bench.py` works too)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
GRAPH_MODES = {"on": True, "off": False, "auto": "auto"}     # --graphs -> GraphEchoTrainer(graphs=...)
PEAK_FP16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_f16 dense peak (--precision f16 only)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None, help="frames per GPU per step (default 32; strong scaling: "
                                                             "--global-batch / N)")
    ap.add_argument("--workload", default=None, choices=["fpn", "fpn_grapher", "full", "temporal"],
                    help="default: fpn_grapher (config 2) at N = 1, full (config 4) at N > 1")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="N > 1: strong (default; global batch fixed, config 4) or weak (per-GPU batch fixed)")
    ap.add_argument("--global-batch", type=int, default=64, help="strong scaling: frames per step over all ranks")
    ap.add_argument("--ddp-mode", default=None, choices=["allreduce", "rs_ag"],
                    help="gradient exchange: bucketed all-reduce (default) or reduce-scatter + sharded optimizer + "
                         "parameter all-gather")
    ap.add_argument("--no-scaling-base", action="store_true", help="N = 1: skip the config-4 leg")
    ap.add_argument("--no-comm-report", action="store_true", help="N > 1: skip the collective microbenchmarks")
    ap.add_argument("--weak-batch", type=int, default=32,
                    help="N > 1 default (strong-scaling) run: also measure 5 steps at this many frames per GPU "
                         "(`weak_point`); 0 = skip")
    ap.add_argument("--clip-len", type=int, default=16, help="frames per clip (temporal workload, config-5 shape)")
    ap.add_argument("--clips", type=int, default=2, help="clips per GPU and step, half source half target (temporal)")
    ap.add_argument("--transport", default="sinkhorn_distance", choices=["sinkhorn_distance", "node_discriminate"],
                    help="TGCN transport loss of the temporal workload: config 5's fp32 SinkhornDistance (default) or the "
                         "reference trainers' default node discriminator")
    ap.add_argument("--backbone", default="resnet", choices=["resnet", "VGG16"])
    ap.add_argument("--in-channel", type=int, default=3, help="input channels (CardiacUDA / config 5: 1)")
    ap.add_argument("--seg-loss", default="camus", choices=["camus", "cardiac"],
                    help="0.1 (Dice + BCE) / 2 on CAMUS (train_camus_echo.py:212) or Dice + BCE over all channels "
                         "(train_cardiac_uda.py:228)")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--precision", default="f32", choices=["f32", "f16", "f16s"],
                    help="f32 (headline): exact fp32 MFMA.  f16: BASELINE config 5's conv path -- fp16 MFMA inputs, fp32 "
                         "accumulation and storage; f16s: + fp16 activation storage inside the VGG16 stacks (each reported "
                         "with its own dtype)")
    ap.add_argument("--graphs", nargs="?", const="on", default="auto", choices=["on", "off", "auto"],
                    help="replay the FPN / discriminator passes from HIP graphs (graphecho_amd/graphs.py); pays when the "
                         "host, not the GPU, bounds the step.  auto (default): for the full / temporal workloads at "
                         "<= 16 frames per step; under data parallelism over RCCL the SyncBN exchanges are captured inside "
                         "the graphs (GE_GRAPHS_DP=partial: only the collective-free pieces)")
    ap.add_argument("--ring", type=int, default=8, help="number of pre-generated resident batches cycled through the steps "
                                                        "(seed 1234 + rank * 1000 + i; SURVEY 8d)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--probe", default=None, help=argparse.SUPPRESS)   # state_dict + frames for the oracle's logits
    ap.add_argument("--probe-only", action="store_true", help=argparse.SUPPRESS)   # only the oracle's logits, nothing timed
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--dp-probe", action="store_true", help=argparse.SUPPRESS)   # sacrificial child of an N > 1 run (run_dp_probe)
    return ap.parse_args()


def cpu_baseline_worker(args):
    """CPU oracle ("port" of the reference's algorithm in PyTorch-CPU ops) on a bounded sample of the workload."""
    from graphecho_amd.models.fpnseg import FPN
    from graphecho_amd.trainer import PyramidGraphers, synthetic_batch
    from oracle.steps import CpuTrainer

    if args.probe_only:
        # checker leg of an auxiliary configuration (config 5's VGG16 network): the oracle's logits for the parent's weights on
        # its probe frames, nothing timed
        from oracle.fpn import fpn_forward
        blob = torch.load(args.probe)
        torch.set_num_threads(min(32, os.cpu_count() or 1))
        with torch.no_grad():
            ref = {"logits": fpn_forward({k: v.clone() for k, v in blob["state_dict"].items()}, blob["frames"], True)[0]}
        torch.save(ref, args.probe + ".out")
        print(json.dumps({"probe_only": True}), flush=True)
        return
    torch.manual_seed(0)
    b = 8
    net = FPN([2, 4, 23, 3], 4, 3, back_bone=args.backbone)
    gsd = None
    if args.workload != "fpn":
        s = args.size // 4
        gsd = PyramidGraphers(256, (s, s // 2, s // 4, s // 8)).state_dict()
    x, m = synthetic_batch(b, 3, 4, args.size, "cpu", 1234)
    # torch-CPU stops scaling (and collapses) far below 256 threads: the best of four thread counts is reported, each
    # measured on one warm-up step + as many steps as fit its share of a ~25 s budget (at least one)
    cands = sorted({t for t in (8, 16, 32, 64) if t <= (os.cpu_count() or 1)} or {os.cpu_count() or 1})
    best, sweep = None, {}
    for threads in cands:
        torch.set_num_threads(threads)
        tr = CpuTrainer(net.state_dict(), gsd, "camus", exact_knn=False)
        t0 = time.time()
        tr.step(x, m)                               # warm-up (also bounds the budget below)
        warm = time.time() - t0
        n, t0 = 0, time.time()
        while n < 1 or (n < 4 and (time.time() - t0) + warm < 25.0 / len(cands)):
            tr.step(x, m)
            n += 1
        dt = (time.time() - t0) / n
        sweep[threads] = round(b / dt, 3)
        if best is None or b / dt > best[0]:
            best = (b / dt, threads, n)
    dt, threads, n = b / best[0], best[1], best[2]
    if args.probe:
        # Dice-vs-oracle leg: the oracle's logits for the parent's initial weights on its probe frames
        from oracle.fpn import fpn_forward
        blob = torch.load(args.probe)
        with torch.no_grad():
            ref = {"logits": fpn_forward({k: v.clone() for k, v in blob["state_dict"].items()}, blob["frames"], True)[0]}
            if "grapher_sd" in blob:
                # the headline workload's other half: the p2 Grapher (k-NN 4096 x 256, max-relative conv) on the p2 map
                # the HIP path produced for these frames -- same input on both sides, so index disagreements are the
                # k-NN's own (near-ties), not upstream rounding
                from oracle import vig as ovig
                seen = {}
                knn = ovig.edge_index
                ovig.edge_index = lambda *a, **k: seen.setdefault("edge", knn(*a, **k))
                try:
                    ref["grapher"] = ovig.grapher_forward({k: v.clone() for k, v in blob["grapher_sd"].items()}, "",
                                                          blob["p2"], 9, 1, 4, "gelu", True, True)
                finally:
                    ovig.edge_index = knn
                ref["edge"] = seen["edge"]
        torch.save(ref, args.probe + ".out")
    print(json.dumps({"value": round(b / dt, 3), "unit": "frames/s", "cores": threads, "kind": "port",
                      "threads_sweep_frames_per_s": sweep,
                      "sample": f"{n} steps of batch {b} @{args.size}x{args.size}, FPN-{args.backbone}"
                                f"{'+Grapher' if gsd else ''} fwd+loss+bwd+Adam/SGD, torch-CPU fp32, best of "
                                f"{sorted(sweep)} threads = {threads} of {os.cpu_count()} host CPUs"}), flush=True)


def probe_parity(probe, ref, args, eps=1e-5):
    """Dice of the HIP path's thresholded prediction against the oracle's on the same frames and weights
    ((2TP+eps)/(2TP+FP+FN+eps), sigmoid > 0.5, train_camus_echo.py:402-417 -- SURVEY.md 8d) and the logit error; for the
    config-2 workload also the p2 Grapher's output and its k-NN neighbour indices against the oracle's on the same p2."""
    logits, ref_logits = probe["logits"], ref["logits"]
    out = _logit_parity(logits, ref_logits, args, eps)
    if "grapher" in ref and "grapher" in probe:
        g, r = probe["grapher"], ref["grapher"]
        e, er = probe["edge"][0], ref["edge"][0]              # neighbour ids (B, N, k); [1] is the centre index
        rows = (e == er).all(-1)
        # a row whose neighbour SET agrees but whose order differs is a tie in distance (order of equal distances)
        same_set = (e.sort(-1)[0] == er.sort(-1)[0]).all(-1)
        out["grapher_p2"] = {"rel_err": float(f"{((g - r).abs().max() / r.abs().max()).item():.3e}"),
                             "knn_index_agreement": round((e == er).float().mean().item(), 6),
                             "knn_rows_identical": round(rows.float().mean().item(), 6),
                             "knn_rows_same_neighbour_set": round(same_set.float().mean().item(), 6),
                             "knn_shape": list(e.shape),
                             "sample": "Grapher(256, k=9, 'mr', gelu, BN, r=4) train-mode forward on the HIP path's p2 of "
                                       "the probe frames (k-NN 4096 x 256 per frame), HIP vs oracle/vig.py + knn_ref.c"}
    return out


def _logit_parity(logits, ref_logits, args, eps=1e-5):
    p, r = logits > 0, ref_logits > 0
    tp = (p & r).sum((0, 2, 3)).double()
    fp = (p & ~r).sum((0, 2, 3)).double()
    fn = (~p & r).sum((0, 2, 3)).double()
    dice = (2 * tp + eps) / (2 * tp + fp + fn + eps)
    rel = ((logits - ref_logits).abs().max() / ref_logits.abs().max()).item()
    return {"dice_vs_oracle": round(dice.mean().item(), 6), "dice_per_class": [round(d, 6) for d in dice.tolist()],
            "pixels_differing": int((p != r).sum()), "logits_rel_err": float(f"{rel:.3e}"),
            "sample": f"2 seeded frames @{args.size}x{args.size}, FPN-{args.backbone} train-mode forward from the initial "
                      "weights (3x3 layers on the kernels of the timed step: Winograd where covered), HIP vs oracle/fpn.py"}


def csrc_sha16():
    """Fingerprint of the kernel sources (graphecho_amd/csrc/*.hip, *.h): what a PMC collection is valid for.  (The GPU boxes
    get a snapshot without .git, so a commit id is not available where the counters are collected.)"""
    import glob
    import hashlib

    h = hashlib.sha256()
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "graphecho_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(root, "*.hip")) + glob.glob(os.path.join(root, "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed PMC summary (profiles/*_traffic.json, written by
    tools/collect_profile.sh: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this same command,
    calibrated on known-byte kernels as MI355X_MICROARCH.md prescribes).  (None, reason) when there is none."""
    import glob

    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*_traffic.json")))
    if not files:
        return None, "no profiles/*_traffic.json", None
    try:
        blob = json.load(open(files[-1]))
        rec = blob.get("bench_launch_average", {}).get(kernel)
    except Exception as exc:   # a malformed summary must not break the bench line
        return None, f"{os.path.basename(files[-1])}: {exc}", None
    if not rec:
        return None, f"{os.path.basename(files[-1])}: kernel not sampled", None
    # the counters are an OFFLINE pass: stamped with the kernel sources they were collected on; stale = the sources have
    # changed since (the number then describes an older build of the kernel)
    stamp = blob.get("collected_on", {}).get("csrc_sha16")
    return (round(rec["hbm_bytes_per_launch"]), f"profiles/{os.path.basename(files[-1])} (rocprofv3 --pmc, offline pass)",
            {"csrc_sha16": stamp, "current_csrc_sha16": csrc_sha16(), "stale": stamp != csrc_sha16()})


def cpu_baseline(args, probe=None):
    """Run the CPU baseline in a child process with a hard timeout so it can never eat the GPU budget."""
    import subprocess

    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--workload", args.workload,
           "--backbone", args.backbone, "--size", str(args.size)] + (["--probe", probe] if probe else [])
    try:
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
        line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
        return json.loads(line)
    except Exception as e:  # timeout or failure: report it, never fake a number
        return {"value": None, "unit": "frames/s", "cores": min(os.cpu_count() or 1, 32), "kind": "port",
                "sample": f"cpu baseline did not finish: {type(e).__name__}"}


def resolve_defaults(args, world):
    """Fill in workload / scaling / per-GPU batch: config 2 at N = 1, config 4 (strong scaling) at N > 1."""
    explicit = args.workload is not None or args.batch is not None
    if args.workload is None:
        args.workload = "fpn_grapher" if world == 1 else "full"
    if args.scaling is None:
        args.scaling = "weak" if (world == 1 or explicit) else "strong"
    if args.batch is None:
        if args.scaling == "strong":
            if args.global_batch % (2 * world):
                raise SystemExit(f"--global-batch {args.global_batch} does not split into source/target halves over {world} ranks")
            args.batch = args.global_batch // world
        else:
            args.batch = 32
    return args


def comm_report(tr, dev, world, syncbn_per_step):
    """What the step's collectives cost on their own (no overlap): every model's gradient buckets exchanged once more,
    timed with events on the current stream, as bus bandwidth (all-reduce: 2 (n-1)/n x bytes / t, the figure to compare
    with the 7 x ~153 GB/s xGMI links); and the latency of one SyncBN-sized all-gather / all-reduce."""
    import torch.distributed as dist

    sync = tr.sync
    out = {"world_size": dist.get_world_size(), "backend": dist.get_backend(),
           "mode": sync.mode, "grad_collectives_per_step": sync.comm_stats["collectives"],
           "grad_bytes_per_rank": sync.comm_stats["bytes"], "buckets": len(sync.buckets)}
    reps = 5
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    scratch = [torch.zeros(b - a, device=dev) for _, a, b, _ in sync.buckets]
    for t in scratch:
        dist.all_reduce(t)
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(reps):
        for t in scratch:
            dist.all_reduce(t)
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / reps
    nbytes = 4 * sum(t.numel() for t in scratch)
    out["allreduce_ms"] = round(ms, 3)
    out["allreduce_busbw_GBps"] = round(2 * (world - 1) / world * nbytes / (ms * 1e-3) / 1e9, 2)
    fwd, bwd, nb = syncbn_per_step
    out["syncbn"] = {"allgathers_per_step": fwd, "allreduces_per_step": bwd, "bytes_per_step": nb}
    small = torch.zeros(256 * 3, device=dev)
    gathered = torch.zeros(world * 256 * 3, device=dev)
    for _ in range(5):
        dist.all_gather_into_tensor(gathered, small)
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(50):
        dist.all_gather_into_tensor(gathered, small)
    ev[1].record()
    torch.cuda.synchronize()
    out["syncbn"]["allgather_us"] = round(ev[0].elapsed_time(ev[1]) / 50 * 1e3, 1)
    sums = torch.zeros(256 * 2, device=dev)          # the backward's per-layer all-reduce of (sum dy, sum dy * xhat)
    for _ in range(5):
        dist.all_reduce(sums)
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(50):
        dist.all_reduce(sums)
    ev[1].record()
    torch.cuda.synchronize()
    out["syncbn"]["allreduce_us"] = round(ev[0].elapsed_time(ev[1]) / 50 * 1e3, 1)
    # one figure for "what a SyncBN exchange costs when nothing hides it": the mean over the step's mix of the two kinds
    per = (fwd * out["syncbn"]["allgather_us"] + bwd * out["syncbn"]["allreduce_us"]) / max(1, fwd + bwd)
    out["syncbn_us_per_collective"] = round(per, 1)
    out["syncbn"]["exposed_ms_per_step_estimate"] = round((fwd + bwd) * per * 1e-3, 3)
    return out


def _full_ring(args, dev, frames, cin, rank=0):
    """args.ring resident (source frames, masks, target frames) batches of the full workload, batch i seeded 1234 + rank * 1000 + i."""
    from graphecho_amd.trainer import synthetic_batch

    ring = [synthetic_batch(frames // 2, cin, 4, args.size, dev, 1234 + rank * 1000 + i) +
            synthetic_batch(frames // 2, cin, 4, args.size, dev, 4321 + rank * 1000 + i)[:1] for i in range(max(1, args.ring))]
    torch.cuda.synchronize()
    return ring


def other_configs(args, dev):
    """N = 1 only: the reference's REAL step (train_camus_echo.py:183-303: FPN on source + target frames, GModule, four
    Discriminators) next to the config-2 headline -- BASELINE config 3 at its own size (8 + 8 frames) and at the metric's
    batch (16 + 16) -- with the whole-step MFMA utilisation (conv FLOPs per step / step time / fp32-MFMA peak)."""
    from graphecho_amd import functional as GF
    from graphecho_amd.trainer import GraphEchoTrainer, synthetic_batch

    out = []
    for frames in (16, 32):
        tr = GraphEchoTrainer(dev, workload="full", back_bone=args.backbone, in_channel=3, num_classes=4,
                              image_size=args.size, seed=0, graphs="auto")
        ring = _full_ring(args, dev, frames, 3)      # seeded 1234 + i / 4321 + i, resident before the timed steps
        k = [0]

        def one():
            k[0] += 1
            return tr.step(*ring[k[0] % len(ring)])

        try:
            for _ in range(4):
                one()
            torch.cuda.synchronize()
        except RuntimeError:            # a capture the runtime refuses: this configuration runs eager
            torch.cuda.synchronize()
            tr._graphs_auto = False
            tr._set_graphs(False)
            for _ in range(4):
                one()
            torch.cuda.synchronize()
        n, t0 = 10, time.perf_counter()
        for _ in range(n):
            one()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        graphs_used = tr.graphs_in_use()
        GF.KERNEL_TIMER = GF.KernelTimer()           # one more step with per-launch records: conv FLOPs of the step
        one()
        torch.cuda.synchronize()
        flops = sum(r[2] for r in GF.KERNEL_TIMER.records)
        executed = sum(r[6] for r in GF.KERNEL_TIMER.records)      # what the matrix pipe ran (Winograd layers: 16 / 36 of the direct FLOPs)
        GF.KERNEL_TIMER = None
        ach = executed / dt / 1e12
        out.append({"workload": ("C3: " if frames == 16 else "") + f"full GraphEcho, source {frames // 2} + target {frames // 2} frames",
                    "frames_per_step": frames, "value": round(frames / dt, 2), "unit": "frames/s",
                    "ms_per_step": round(1e3 * dt, 3), "steps": n, "ring": len(ring), "hip_graphs": graphs_used,
                    "whole_step": {"conv_gflop_per_step": round(flops / 1e9, 1), "executed_gflop_per_step": round(executed / 1e9, 1),
                                   "achieved": round(ach, 2), "frac": round(ach / PEAK_FP32_MFMA_TFLOPS, 4),
                                   "algorithmic_tflops": round(flops / dt / 1e12, 2)}})
        del tr
        torch.cuda.empty_cache()
    return out


def config5(args, dev):
    """N = 1 only: BASELINE config 5 as the reference runs it (train_cardiac_uda.py:73,222-320: FPN(in_channel=1,
    back_bone="VGG16"), Dice + BCE over all channels, GModule + Discriminators on 8 + 8 frames, a source and a target clip of
    16 frames through FPN, GModule, TGCN and the fp32 SinkhornDistance) on this one GPU: fp32, and its stated dtype -- fp16
    MFMA conv path with fp16 activation storage in the VGG16 stacks ("f16s", csrc/ge_half.hip), fp32 Sinkhorn / statistics --
    with that run's dominant conv kernel against the 2.5 PFLOP/s fp16-MFMA peak."""
    from graphecho_amd import functional as GF
    from graphecho_amd.trainer import GraphEchoTrainer, synthetic_batch

    out = []
    nb, t, c, size = 8, 16, 2, args.size

    def clip(seed):
        f, mk = synthetic_batch(c // 2 * t, 1, 4, size, dev, seed)
        return (f.reshape(c // 2, t, 1, size, size).permute(0, 2, 3, 4, 1).contiguous(),
                mk.reshape(c // 2, t, 4, size, size).permute(0, 2, 3, 4, 1).contiguous())

    ring = []
    for i in range(max(1, args.ring)):      # step i on the batch seeded 1234 + i, all resident before the timed steps
        xs, ms = synthetic_batch(nb, 1, 4, size, dev, 1234 + i)
        xt, _ = synthetic_batch(nb, 1, 4, size, dev, 4321 + i)
        cs, cm = clip(770000 + i)
        ct, _ = clip(780000 + i)
        ring.append((xs, ms, xt, {"source": cs, "target": ct, "masks": cm}))
    torch.cuda.synchronize()
    kk = [0]

    def one(tr):
        kk[0] += 1
        return tr.step(*ring[kk[0] % len(ring)])

    frames = 2 * nb + c * t
    # "Dice vs ref" of this configuration in EACH dtype (BASELINE's metric; VERDICT r4: the fp16 path had no reference-anchored
    # number in this line): the network's logits on two seeded frames from the initial weights, in the precision the timed steps
    # run, against oracle/fpn.py on the same weights and frames (computed by the CPU child: the checker, never the thing measured)
    probe_ref, probe_path = None, None
    px, _ = synthetic_batch(2, 1, 4, size, "cpu", 4242)
    for prec in ("f32", "f16s"):
        tr = GraphEchoTrainer(dev, workload="temporal", back_bone="VGG16", in_channel=1, num_classes=4, image_size=size,
                              seed=0, conv_precision=prec, clip_len=t, transport_method="sinkhorn_distance",
                              seg_loss="cardiac", graphs=GRAPH_MODES[args.graphs])
        parity = None
        if not args.no_cpu_baseline:
            sd0 = {k: v.detach().cpu().clone() for k, v in tr.network.state_dict().items()}
            saved = (GF.CONV_PRECISION, GF.ACT_STORAGE)
            GF.CONV_PRECISION, GF.ACT_STORAGE = ("f16", "f16") if prec == "f16s" else (prec, "f32")
            try:
                with torch.no_grad():
                    lg = tr.network(px.to(dev))[0].float().cpu()
            finally:
                GF.CONV_PRECISION, GF.ACT_STORAGE = saved
            tr.load_states({"Net": sd0})          # undo the probe forward's running-statistics update
            if probe_ref is None:                  # both precisions start from the same seeded weights: one oracle pass
                import subprocess
                import tempfile

                probe_path = os.path.join(tempfile.mkdtemp(prefix="ge_probe5_"), "probe.pt")
                torch.save({"state_dict": sd0, "frames": px}, probe_path)
                cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--probe-only", "--probe", probe_path,
                       "--backbone", "VGG16", "--size", str(size)]
                try:
                    subprocess.run(cmd, capture_output=True, text=True, timeout=180)
                    probe_ref = torch.load(probe_path + ".out")
                except Exception as exc:
                    probe_ref = {"error": f"{type(exc).__name__}: {str(exc)[:100]}"}
            if "logits" in probe_ref:
                pa = argparse.Namespace(size=size, backbone="VGG16")
                parity = _logit_parity(lg, probe_ref["logits"], pa)
                parity["sample"] = (f"2 seeded 1-channel frames @{size}x{size}, FPN-VGG16 train-mode forward from the initial weights "
                                    f"in this row's dtype, HIP vs oracle/fpn.py (fp32 CPU)")
            else:
                parity = {"dice_vs_oracle": None, "note": probe_ref.get("error", "oracle leg did not finish")}
        for _ in range(8):      # (the TGCN recurrence is captured into a HIP graph at its third call, its backward one call later)
            one(tr)
        torch.cuda.synchronize()
        n, t0 = 8, time.perf_counter()
        for _ in range(n):
            one(tr)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        GF.KERNEL_TIMER = GF.KernelTimer()           # one more step with per-launch records (weight gradients on the main stream)
        one(tr)
        torch.cuda.synchronize()
        roof = GF.KERNEL_TIMER.summary(PEAK_FP16_MFMA_TFLOPS if prec == "f16s" else PEAK_FP32_MFMA_TFLOPS)
        GF.KERNEL_TIMER = None
        row = {"workload": "C5: FPN(VGG16, 1 channel) + GModule + Discriminators on 8 + 8 frames, 2 clips x 16 frames "
                           "through FPN / GModule / TGCN / SinkhornDistance",
               "dtype": {"f32": "f32", "f16s": "f16 MFMA inputs + f16 activation storage in the VGG16 stacks, f32 accumulate / "
                                               "statistics / Sinkhorn"}[prec],
               "frames_per_step": frames, "value": round(frames / dt, 2), "unit": "frames/s",
               "ms_per_step": round(1e3 * dt, 3), "steps": n, "ring": len(ring)}
        if roof:
            row["roofline"] = {k: roof[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "launches",
                                                    "avg_launch_ms", "all_conv_kernels", "per_kernel")}
        if parity is not None:
            row["parity"] = parity
        out.append(row)
        del tr
        torch.cuda.empty_cache()
    return out


def compute_only_step_ms(args, dev, batch, steps):
    """N > 1: the same per-rank batch stepped WITHOUT any exchange (local BatchNorm statistics, no gradient collectives) on
    this rank's GPU: step time - this = what the exchange costs the step after overlap (`comm.exposed_ms_per_step`)."""
    from graphecho_amd.trainer import GraphEchoTrainer, synthetic_batch

    tr = GraphEchoTrainer(dev, workload=args.workload, back_bone=args.backbone, in_channel=args.in_channel, num_classes=4,
                          image_size=args.size, distributed=False, seed=0, conv_precision=args.precision,
                          clip_len=args.clip_len, transport_method=args.transport, seg_loss=args.seg_loss,
                          graphs=GRAPH_MODES[args.graphs])
    if args.workload in ("full", "temporal"):
        xs, ms = synthetic_batch(batch // 2, args.in_channel, 4, args.size, dev, 1234)
        xt, _ = synthetic_batch(batch // 2, args.in_channel, 4, args.size, dev, 4321)
        if args.workload == "temporal":
            return None          # (clips: not rebuilt here)
        step = lambda: tr.step(xs, ms, xt)
    else:
        xs, ms = synthetic_batch(batch, args.in_channel, 4, args.size, dev, 1234)
        step = lambda: tr.step(xs, ms)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


def distributed_leg(args, dev, world, rank, batch, steps, warmup, local_bn=False):
    """N > 1, collective: one more full-GraphEcho trainer under data parallelism at `batch` frames per rank, stepped
    `steps` times between barriers; returns max-over-ranks ms/step.  local_bn: the same step and the same gradient
    exchange with rank-LOCAL BatchNorm statistics (no SyncBN collectives) -- the difference to the SyncBN step is what
    SyncBN's 100 latency-bound collectives cost the step after overlap, measured instead of estimated."""
    import torch.distributed as dist
    from graphecho_amd import nn as gnn
    from graphecho_amd.trainer import GraphEchoTrainer, synthetic_batch

    tr = GraphEchoTrainer(dev, workload="full", back_bone=args.backbone, in_channel=args.in_channel, num_classes=4,
                          image_size=args.size, distributed=True, seed=0, conv_precision=args.precision,
                          seg_loss=args.seg_loss, graphs=GRAPH_MODES[args.graphs])
    if local_bn:
        for m in tr.network.modules():
            if isinstance(m, gnn.BatchNorm2d):
                m.sync = False
    ring = _full_ring(args, dev, batch, args.in_channel, rank)
    for i in range(warmup):
        tr.step(*ring[i % len(ring)])
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        tr.step(*ring[(warmup + i) % len(ring)])
    dist.barrier()
    torch.cuda.synchronize()
    t = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    del tr
    torch.cuda.empty_cache()
    return 1e3 * float(t.item()) / steps


def scaling_base(args, dev):
    """N = 1 only: BASELINE config 4's workload (full GraphEcho, global batch 64) on this one GPU -- the N = 1 point of
    the strong-scaling curve `--gpus N` measures for N > 1."""
    from graphecho_amd.trainer import GraphEchoTrainer, synthetic_batch

    gb = args.global_batch
    tr = GraphEchoTrainer(dev, workload="full", back_bone=args.backbone, in_channel=3, num_classes=4, image_size=args.size,
                          seed=0)
    ring = _full_ring(args, dev, gb, 3)
    for i in range(4):
        tr.step(*ring[i % len(ring)])
    torch.cuda.synchronize()
    n, t0 = 20, time.perf_counter()      # (20 steps: the N = 1 anchor of the scaling curve moved 11 % between boxes on 8)
    for i in range(n):
        tr.step(*ring[i % len(ring)])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    return {"workload": "C4: full GraphEcho (FPN src+tgt, GModule, 4 Discriminators)", "global_batch": gb, "n_gpus": 1,
            "value": round(gb / dt, 2), "unit": "frames/s", "ms_per_step": round(1e3 * dt, 3), "steps": n}


def dp_probe_worker(args):
    """Sacrificial child of an N > 1 run (one per rank, a rendezvous of its own): the full-GraphEcho trainer under data
    parallelism at the parent's per-rank batch with everything static replayed from HIP graphs -- the SyncBN exchanges
    captured inside them, on their own RCCL communicator, beside eager gradient buckets.  Six steps, finite losses on every
    rank, exit code 0.  That form has run on a one-rank RCCL group only (no N > 1 box was available to the build): the
    parent runs it here first, in a process it can kill, and falls back to the round-5 form (GE_GRAPHS_DP=partial) on any
    failure -- an error, a refused capture, a hang -- instead of losing the line."""
    import datetime
    import math

    import torch.distributed as dist

    fake = os.environ.get("GE_DP_PROBE_FAKE", "")      # tests of the parent's handling
    if fake == "hang":
        time.sleep(3600)
    if fake == "fail":
        raise SystemExit(3)
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("GE_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    limit = datetime.timedelta(seconds=120)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev, timeout=limit)
    else:
        dist.init_process_group(backend, timeout=limit)
    from graphecho_amd.trainer import GraphEchoTrainer

    tr = GraphEchoTrainer(dev, workload="full", back_bone=args.backbone, in_channel=args.in_channel, num_classes=4,
                          image_size=args.size, distributed=True, seed=0, conv_precision=args.precision,
                          seg_loss=args.seg_loss, graphs=GRAPH_MODES[args.graphs])
    if world == 1:      # (the one-rank self-test of this worker: every collective call issued all the same)
        from graphecho_amd import nn as gnn

        tr.sync.force = True
        for model in tr.modules.values():
            for mod in model.modules():
                if isinstance(mod, gnn.BatchNorm2d):
                    mod.force_sync = True
    ring = _full_ring(args, dev, args.batch, args.in_channel, rank)
    ok = 1
    for i in range(6):
        if not math.isfinite(float(tr.step(*ring[i % len(ring)]))):      # (the step's summed loss)
            ok = 0
    torch.cuda.synchronize()
    flag = torch.tensor([ok], device=dev, dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    ok = int(flag.item())
    print(f"DP_PROBE rank {rank} ok={ok} hip_graphs={tr.graphs_in_use()}", flush=True)
    os._exit(0 if ok else 4)      # (no teardown: the parent only wants the verdict)


def dp_probe_wanted(args, world, backend):
    """The probe runs when the default would replay the SyncBN backbone with captured RCCL exchanges: N > 1 over nccl, a
    workload with the phased step, graphs on (or auto at a per-rank batch under the auto threshold), nothing forced by the
    environment.  GE_DP_PROBE=0 skips it, GE_DP_PROBE=force runs it whatever the backend (the gloo rehearsal's test)."""
    from graphecho_amd.trainer import GraphEchoTrainer

    sw = os.environ.get("GE_DP_PROBE", "1")
    if world == 1 or sw == "0" or args.workload not in ("full", "temporal") or args.graphs == "off":
        return False
    if sw == "force":
        return True
    if backend != "nccl" or "GE_GRAPHS_DP" in os.environ or os.environ.get("GE_GRAPHS") == "0":
        return False
    frames = args.batch + (args.clips * args.clip_len if args.workload == "temporal" else 0)
    cap = GraphEchoTrainer.GRAPHS_AUTO_MAX_FRAMES_F16 if args.precision in ("f16", "f16s") else GraphEchoTrainer.GRAPHS_AUTO_MAX_FRAMES
    return args.graphs == "on" or frames <= cap


def run_dp_probe(args, world):
    """Parent side: (ok, note, seconds).  The child gets this rank's environment with a rendezvous port of its own (a fixed
    function of MASTER_PORT: every rank computes the same one) and without torchrun's agent store; it is killed (its exact
    PID) when GE_DP_PROBE_TIMEOUT_S [240] pass without a verdict."""
    import subprocess

    port = 20000 + (int(os.environ.get("MASTER_PORT", "29500")) * 7 + 13) % 20000
    env = dict(os.environ, MASTER_PORT=str(port))
    env.pop("TORCHELASTIC_USE_AGENT_STORE", None)
    cmd = [sys.executable, os.path.abspath(__file__), "--dp-probe", "--gpus", str(world), "--batch", str(args.batch),
           "--backbone", args.backbone, "--in-channel", str(args.in_channel), "--size", str(args.size), "--precision",
           args.precision, "--seg-loss", args.seg_loss, "--graphs", args.graphs, "--ring", "2"]
    limit = int(os.environ.get("GE_DP_PROBE_TIMEOUT_S", "240"))
    t0 = time.perf_counter()
    try:
        res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=limit)
        ok = res.returncode == 0
        note = None if ok else f"exit code {res.returncode}: {(res.stderr or res.stdout).strip()[-160:]}"
    except subprocess.TimeoutExpired:
        ok, note = False, f"no verdict within {limit} s (killed)"
    return ok, note, time.perf_counter() - t0


def main():
    args = parse()
    if args.dp_probe:
        dp_probe_worker(args)
        return
    if args.cpu_baseline_only:
        args.workload = args.workload or "fpn_grapher"
        cpu_baseline_worker(args)
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args = resolve_defaults(args, world)
    if args.ddp_mode:
        os.environ["GE_DDP_MODE"] = args.ddp_mode
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    backend = os.environ.get("GE_DIST_BACKEND", "nccl")   # "gloo": rehearsal of the N > 1 path on a 1-GPU box (tests)
    if backend != "nccl":
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dp_probe = None
    if dp_probe_wanted(args, world, backend):
        dp_probe = run_dp_probe(args, world)
    if world > 1:
        import torch.distributed as dist

        import datetime

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # a collective that never completes (a rank died, a mismatched exchange) must end the run with an error instead
        # of hanging the node until the driver's own limit: 5-minute collective timeout, asynchronous error handling on
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
        limit = datetime.timedelta(seconds=int(os.environ.get("GE_DIST_TIMEOUT_S", "300")))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=limit)
        else:
            dist.init_process_group(backend, timeout=limit)
    dp_probe_note = None
    if dp_probe is not None:      # every rank takes the same form
        flag = torch.tensor([1 if dp_probe[0] else 0], device=dev if backend == "nccl" else "cpu", dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            os.environ["GE_GRAPHS_DP"] = "partial"
            dp_probe_note = ("replay of the SyncBN backbone with captured RCCL exchanges failed its probe (" +
                             (dp_probe[1] or "on another rank") + "); head + discriminators replayed only")

    from graphecho_amd import functional as GF
    from graphecho_amd.trainer import GraphEchoTrainer, synthetic_batch

    cin = args.in_channel
    tr = GraphEchoTrainer(dev, workload=args.workload, back_bone=args.backbone, in_channel=cin, num_classes=4,
                          image_size=args.size, distributed=world > 1, seed=0, conv_precision=args.precision,
                          clip_len=args.clip_len, transport_method=args.transport, graphs=GRAPH_MODES[args.graphs],
                          seg_loss=args.seg_loss)
    # parity probe (rank 0, N = 1, with the CPU leg): this network's logits on two seeded frames, from the initial weights;
    # the CPU-baseline child computes the oracle's logits for the same weights and frames
    probe = None
    if world == 1 and not args.no_cpu_baseline and args.precision == "f32":
        import tempfile

        probe = {"path": os.path.join(tempfile.mkdtemp(prefix="ge_probe_"), "probe.pt")}
        sd0 = {k: v.detach().cpu().clone() for k, v in tr.network.state_dict().items()}
        px, _ = synthetic_batch(2, cin, 4, args.size, "cpu", 4242)
        blob = {"state_dict": sd0, "frames": px}
        # (two frames would fall under the Winograd kernels' grid threshold: lowered for the probe, so that the convolution
        # kernels it checks are the ones the timed batch runs -- functional.WINOGRAD_MIN_BLOCKS)
        wino_min, GF.WINOGRAD_MIN_BLOCKS = GF.WINOGRAD_MIN_BLOCKS, 1
        with torch.no_grad():
            lg, pyr = tr.network(px.to(dev))
            GF.WINOGRAD_MIN_BLOCKS = wino_min
            probe["logits"] = lg.float().cpu()
            if args.workload == "fpn_grapher":
                blk = tr.graphers.blocks[0]
                gsd0 = {k: v.detach().cpu().clone() for k, v in blk.state_dict().items()}
                seen = []
                hook = blk.graph_conv.dilated_knn_graph.register_forward_hook(lambda _m, _i, o: seen.append(o))
                probe["grapher"] = blk(pyr[0]).float().cpu()
                hook.remove()
                probe["edge"] = seen[0].cpu()
                blk.load_state_dict(gsd0)
                blob.update(grapher_sd=gsd0, p2=pyr[0].float().cpu())
        tr.network.load_state_dict(sd0)          # undo the probe forward's running-statistics update
        torch.save(blob, probe["path"])
    frames_per_step = args.batch
    # step i of rank r runs on the batch seeded 1234 + r * 1000 + i.  A ring of RING pre-generated batches, all
    # resident in HBM before the timed region, is cycled through warm-up and timed steps alike: GModule's data-dependent node
    # counts, its hallucination branch and the seed-bank clustering see different masks every step (the convolutions do not care).
    RING = max(1, args.ring)
    it = {"i": 0}

    def nxt(ring):
        b = ring[it["i"] % len(ring)]
        it["i"] += 1
        return b

    if args.workload == "temporal":
        # config-5 shape: a source + a target frame batch (as config 3) and `clips` clips of `clip_len` frames that go
        # through FPN (folded into the batch), GModule and TGCN + SinkhornDistance (train_camus_echo.py:232-290)
        nb, t, c = args.batch // 2, args.clip_len, args.clips

        def clip(seed):
            f, mk = synthetic_batch(c // 2 * t, cin, 4, args.size, dev, seed)
            f = f.reshape(c // 2, t, cin, args.size, args.size).permute(0, 2, 3, 4, 1).contiguous()
            mk = mk.reshape(c // 2, t, 4, args.size, args.size).permute(0, 2, 3, 4, 1).contiguous()
            return f, mk

        ring = []
        for i in range(RING):
            xs, ms = synthetic_batch(nb, cin, 4, args.size, dev, 1234 + rank * 1000 + i)
            xt, _ = synthetic_batch(nb, cin, 4, args.size, dev, 4321 + rank * 1000 + i)
            cs, cm = clip(770000 + rank * 1000 + i)
            ct, _ = clip(780000 + rank * 1000 + i)
            ring.append((xs, ms, xt, {"source": cs, "target": ct, "masks": cm}))
        frames_per_step = 2 * nb + c * t
        step = lambda: tr.step(*nxt(ring))
    elif args.workload == "full":
        ring = [synthetic_batch(args.batch // 2, cin, 4, args.size, dev, 1234 + rank * 1000 + i) +
                synthetic_batch(args.batch // 2, cin, 4, args.size, dev, 4321 + rank * 1000 + i)[:1] for i in range(RING)]
        step = lambda: tr.step(*nxt(ring))
    else:
        ring = [synthetic_batch(args.batch, cin, 4, args.size, dev, 1234 + rank * 1000 + i) for i in range(RING)]
        step = lambda: tr.step(*nxt(ring))
    torch.cuda.synchronize()      # every batch of the ring is resident before anything is timed

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    graphs_note = None
    try:
        for _ in range(args.warmup):
            step()
        ok = 1
    except RuntimeError as exc:        # a capture the runtime refuses (the eager path has no such failure mode)
        if not tr.use_graphs:
            raise
        ok, graphs_note = 0, f"HIP-graph capture failed on rank {rank} ({type(exc).__name__}: {str(exc)[:120]}); eager steps"
    if tr.use_graphs and world > 1:    # every rank takes the same path
        flag = torch.tensor([ok], device=dev, dtype=torch.int32)
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
        if int(flag.item()) == 0 and ok:
            ok, graphs_note = 0, "HIP-graph capture failed on another rank; eager steps"
    if not ok:
        torch.cuda.synchronize()
        tr.use_graphs = False
        tr._graphs_auto = False
        tr._set_graphs(False)
        for _ in range(args.warmup):
            step()
    fence()
    GF.SYNC_BN_STATS[:] = [0, 0, 0]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    syncbn_per_step = [v // max(1, args.steps) for v in GF.SYNC_BN_STATS]
    rank_ms = None
    if world > 1:
        mine = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        every = torch.zeros(world, device=dev, dtype=torch.float64)
        torch.distributed.all_gather_into_tensor(every, mine)
        rank_ms = [round(1e3 * v / args.steps, 3) for v in every.tolist()]
        elapsed = float(every.max().item())

    # live per-kernel timing of the conv kernels (HIP events on the launch stream), a few extra steps
    roof = None
    if not args.no_kernel_timing:
        GF.KERNEL_TIMER = GF.KernelTimer()
        n_timed = min(3, args.steps)
        for _ in range(n_timed):
            step()
        torch.cuda.synchronize()
        roof = GF.KERNEL_TIMER.summary(PEAK_FP16_MFMA_TFLOPS if args.precision in ("f16", "f16s") else PEAK_FP32_MFMA_TFLOPS)
        if roof is not None:
            # MFMA_util of the whole step = conv FLOPs per step / wall step time / peak
            # (SURVEY 8d's algorithmic FLOPs: direct convolution; `achieved` / `frac` count what the matrix pipe executed --
            # the Winograd layers run 16 / 36 of their direct FLOPs -- so the fraction stays a utilisation)
            flops_step = sum(r[2] for r in GF.KERNEL_TIMER.records) / n_timed
            exec_step = sum(r[6] for r in GF.KERNEL_TIMER.records) / n_timed
            ach = exec_step / (elapsed / args.steps) / 1e12
            roof["whole_step"] = {"conv_gflop_per_step": round(flops_step / 1e9, 1),
                                  "executed_gflop_per_step": round(exec_step / 1e9, 1), "achieved": round(ach, 2),
                                  "frac": round(ach / roof["peak"], 4),
                                  "algorithmic_tflops": round(flops_step / (elapsed / args.steps) / 1e12, 2)}
        GF.KERNEL_TIMER = None
        if roof is not None:
            roof["traffic"], roof["traffic_source"], stamp = pmc_traffic(roof["kernel"])
            if stamp is not None:
                roof["traffic_collected_on"] = stamp["csrc_sha16"]
                roof["stale"] = stamp["stale"]

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
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": {"f32": "f32", "f16": "f16 MFMA inputs, f32 accumulate/storage (config 5 conv path)",
                      "f16s": "f16 MFMA inputs, f32 accumulate; VGG16 conv stacks store activations and activation gradients as "
                              "channel-blocked f16 (config 5 conv path, csrc/ge_half.hip); everything else f32"}[args.precision],
            "data": f"synthetic (ring of {RING} resident batches, seed 1234 + rank * 1000 + i)",
            "config": {"ring": RING, "workload": {"fpn": "C1-shaped: FPN-only 4-class seg",
                                    "fpn_grapher": "C2: FPN(" + args.backbone + ")+ViG Grapher fwd/bwd+Adam/SGD",
                                    "full": ("C4" if world > 1 else "C3") + ": full GraphEcho (FPN src+tgt, GModule, 4 Discriminators)",
                                    "temporal": f"C5-shaped: full GraphEcho + temporal branch ({args.clips} clips x "
                                                f"{args.clip_len} frames through FPN, GModule, TGCN, SinkhornDistance)"}[args.workload],
                       "per_gpu_batch": frames_per_step, "global_batch": frames_per_step * world, "image": f"{cin}x{args.size}x{args.size}",
                       "parallelism": f"dp{world}" + ("+syncbn" if world > 1 else ""),
                       "hip_graphs": tr.graphs_in_use(), **({"hip_graphs_note": graphs_note} if graphs_note else {}),
                       **({"dp_probe": {"ok": dp_probe_note is None, "seconds": round(dp_probe[2], 1),
                                        **({"note": dp_probe_note} if dp_probe_note else {})}} if dp_probe is not None else {}),
                       "merged_fpn_passes": "n/a (one FPN pass per step)" if args.workload in ("fpn", "fpn_grapher") else
                       (("source+target+clips" if (tr.merge_clips and args.workload == "temporal") else "source+target")
                        if tr.merge_passes else False)},
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, probe["path"] if probe else None)
            if probe and os.path.exists(probe["path"] + ".out"):
                try:
                    out["parity"] = probe_parity(probe, torch.load(probe["path"] + ".out"), args)
                except Exception as exc:      # noqa: BLE001  (never lose the line to the probe's bookkeeping)
                    out["parity"] = {"error": f"{type(exc).__name__}: {str(exc)[:200]}"}
    # ---- N > 1: the auxiliary legs (collective microbenchmarks, exchange-free step, local-BN step, the weak-scaling point).
    # The timed region is over and the line's required fields are complete: nothing below may cost the line.  An exception
    # in a leg is recorded in its place and ends the legs (the ranks' collectives may no longer pair up); a leg that STALLS
    # (one rank failed, the others wait in a collective) is ended by a watchdog after GE_AUX_TIMEOUT_S [300]: rank 0 prints
    # the line with what it has and every rank exits 0.
    import threading

    emit_lock, emitted = threading.Lock(), [False]

    def emit():
        with emit_lock:
            if rank == 0 and not emitted[0]:
                for attempt in range(5):      # (the watchdog may find the main thread adding a key)
                    try:
                        line = json.dumps(out)
                        break
                    except RuntimeError:
                        time.sleep(0.05)
                print(line, flush=True)
            emitted[0] = True

    def bail():
        if rank == 0 and not emitted[0]:
            out["aux_note"] = "auxiliary legs did not finish within GE_AUX_TIMEOUT_S; line printed by the watchdog"
        emit()
        os._exit(0)

    watchdog = None
    if world > 1:
        watchdog = threading.Timer(float(os.environ.get("GE_AUX_TIMEOUT_S", "300")), bail)
        watchdog.daemon = True
        watchdog.start()
    comm, weak_point, aux_error = None, None, None
    default_c4 = world > 1 and args.workload == "full" and args.scaling == "strong" and not args.no_comm_report
    if world > 1 and os.environ.get("GE_AUX_FAKE") == "hang" and rank == world - 1:      # tests of the watchdog
        time.sleep(3600)
    try:
        if world > 1 and not args.no_comm_report:     # collective: all ranks
            comm = comm_report(tr, dev, world, syncbn_per_step)
            comm["per_rank_ms_per_step"] = {"min": min(rank_ms), "max": max(rank_ms), "all": rank_ms}
            co = compute_only_step_ms(args, dev, args.batch, max(3, args.steps // 2))      # every rank: equal load on the node
            if co is not None:
                t = torch.tensor([co], device=dev, dtype=torch.float64)
                torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
                comm["compute_only_ms_per_step"] = round(float(t.item()), 3)
                comm["exposed_ms_per_step"] = round(1e3 * elapsed / args.steps - float(t.item()), 3)
        if comm is not None and args.workload == "full":
            # SyncBN's cost after overlap, MEASURED: the same per-rank batch and gradient exchange with local statistics
            lb = distributed_leg(args, dev, world, rank, args.batch, max(3, args.steps // 2), 3, local_bn=True)
            comm["syncbn"]["local_bn_ms_per_step"] = round(lb, 3)
            comm["syncbn"]["exposed_ms"] = round(1e3 * elapsed / args.steps - lb, 3)
        if default_c4 and args.weak_batch > 0:
            # the other curve's point in the same run: config 4's workload at a FIXED 32 frames per GPU (weak scaling); its
            # N = 1 value is the "source 16 + target 16" entry of the N = 1 line's `other_configs`
            wb = args.weak_batch
            ms_w = distributed_leg(args, dev, world, rank, wb, 5, 3)
            weak_point = {"workload": "C4: full GraphEcho (FPN src+tgt, GModule, 4 Discriminators)", "scaling": "weak",
                          "per_gpu_batch": wb, "global_batch": wb * world, "n_gpus": world, "steps": 5,
                          "ms_per_step": round(ms_w, 3), "value": round(wb * world / (ms_w * 1e-3), 2), "unit": "frames/s",
                          "n1_reference": f"other_configs[frames_per_step={wb}] of the N = 1 line (same workload, one GPU)"}
    except Exception as exc:      # noqa: BLE001
        aux_error = f"{type(exc).__name__}: {str(exc)[:200]}"
    if rank == 0:
        if aux_error is not None:
            out["aux_error"] = aux_error
        if comm is not None:
            out["comm"] = comm
        if weak_point is not None:
            out["weak_point"] = weak_point
        if default_c4:
            # both curves' points of this N side by side (the line's `value` is the strong one: config 4 fixes the GLOBAL batch)
            out["strong"] = {"scaling": "strong", "global_batch": frames_per_step * world, "per_gpu_batch": frames_per_step,
                             "value": out["value"], "ms_per_step": out["ms_per_step"], "unit": "frames/s"}
            out["weak"] = weak_point
        if world == 1 and args.workload == "fpn_grapher" and not args.no_scaling_base:
            del tr, step
            torch.cuda.empty_cache()
            # auxiliary legs: a failure in one of them (a graph capture the runtime refuses, memory) must never cost the
            # headline line -- it is reported in place of the leg's numbers
            for key, leg in (("other_configs", other_configs), ("config5", config5), ("scaling_base", scaling_base)):
                try:
                    out[key] = leg(args, dev)
                except Exception as exc:      # noqa: BLE001
                    out[key] = {"error": f"{type(exc).__name__}: {str(exc)[:200]}"}
                    torch.cuda.synchronize()
    emit()
    if world > 1:
        # (the watchdog stays armed over the teardown: a rank that left early must not hold the others in it)
        if aux_error is None:
            torch.distributed.destroy_process_group()
        watchdog.cancel()
        if aux_error is not None:
            os._exit(0)


if __name__ == "__main__":
    main()
