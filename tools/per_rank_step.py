"""Per-rank step of config 4 under data parallelism (one-rank RCCL group: the trainer's distributed code path, partial
graph replay, GModule stream) at a given number of frames.  usage: per_rank_step.py FRAMES [auto|off|on] [dist|local] [force]
force: issue every collective of the step although the group has one rank (SyncBN all-gathers / all-reduces, gradient buckets, the
"used" map): without it a one-rank group skips them and the figure is the distributed CODE PATH without its exchanges."""
import os, sys, time, torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
from graphecho_amd.trainer import GraphEchoTrainer, synthetic_batch
dev = torch.device("cuda:0")
pg = os.environ.get("PG", "nccl_eager")     # none | gloo | nccl_lazy | nccl_eager | nccl_used (one all-reduce issued)
if pg == "gloo":
    dist.init_process_group("gloo", rank=0, world_size=1)
elif pg == "nccl_lazy":
    dist.init_process_group("nccl", rank=0, world_size=1)
elif pg in ("nccl_eager", "nccl_used"):
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    if pg == "nccl_used":
        t = torch.ones(1024, device=dev); dist.all_reduce(t); torch.cuda.synchronize()
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 8
mode = {"auto": "auto", "off": False, "on": True}[sys.argv[2] if len(sys.argv) > 2 else "auto"]
distributed = (sys.argv[3] if len(sys.argv) > 3 else "dist") == "dist" and pg != "none"
tr = GraphEchoTrainer(dev, workload="full", distributed=distributed, seed=0, graphs=mode)
force = len(sys.argv) > 4 and sys.argv[4] == "force" and distributed
if force:
    from graphecho_amd import nn as gnn
    tr.sync.force = True
    for model in tr.modules.values():
        for mod in model.modules():
            if isinstance(mod, gnn.BatchNorm2d):
                mod.force_sync = True
xs, ms = synthetic_batch(frames // 2, 3, 4, 256, dev, 1)
xt, _ = synthetic_batch(frames // 2, 3, 4, 256, dev, 2)
for _ in range(8):
    tr.step(xs, ms, xt)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = int(os.environ.get("PRS_STEPS", "30"))      # (PRS_STEPS=400: a soak of the replayed distributed step; loss and memory printed)
mem0 = torch.cuda.memory_allocated()
for _ in range(n):
    last = tr.step(xs, ms, xt)
torch.cuda.synchronize()
if n > 100:
    print(f"  after {n} steps: loss {float(last):.4f} (finite: {bool(torch.isfinite(last))}), allocated {mem0 >> 20} -> {torch.cuda.memory_allocated() >> 20} MiB")
from graphecho_amd import functional as GF
print(f"per-rank step, {frames} frames, pg={pg}, distributed={distributed}, collectives={'forced' if force else 'skipped (one rank)'}, "
      f"graphs={tr.graphs_in_use()}: {1e3 * (time.perf_counter() - t0) / n:.2f} ms"
      + (f"  [SyncBN exchanges issued by the host per step: {GF.SYNC_BN_STATS[0] // (n + 8)} + {GF.SYNC_BN_STATS[1] // (n + 8)}]" if force else ""))
if pg != 'none':
    dist.destroy_process_group()
