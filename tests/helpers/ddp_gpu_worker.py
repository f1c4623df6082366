"""Worker of tests/test_models_gpu.py::test_ddp_world2_gloo_on_one_gpu: one of two ranks sharing cuda:0, gloo backend.

Runs two distributed training steps on this rank's half of a fixed batch and saves the flat parameters, the first
BatchNorm's running statistics and the losses.  usage: ddp_gpu_worker.py RANK WORLD PORT OUTDIR"""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
workload = sys.argv[5] if len(sys.argv) > 5 else "fpn_grapher"
variant = sys.argv[6] if len(sys.argv) > 6 else ""      # "rs_ag": sharded exchange; "few1": rank 1's masks yield < 6 nodes
if variant == "rs_ag":
    os.environ["GE_DDP_MODE"] = "rs_ag"
if variant.endswith("_phased"):                          # backward cut at the pyramid into three autograd calls
    os.environ["GE_SPLIT_BACKWARD"] = "1"
    variant = variant[:-len("_phased")]
os.environ["MASTER_ADDR"] = "127.0.0.1"
os.environ["MASTER_PORT"] = port
dist.init_process_group("gloo", rank=rank, world_size=world)
from graphecho_amd.trainer import GraphEchoTrainer, synthetic_batch

dev = torch.device("cuda:0")
cin = 1 if variant.startswith("vgg_") else 3
x, m = synthetic_batch(4, cin, 4, 128, dev, 7)
per = x.shape[0] // world
xs, ms = x[rank * per:(rank + 1) * per], m[rank * per:(rank + 1) * per]
if variant == "few1" and rank == 1:                      # 2 x 2 pixel masks contain no sampling location: GModule returns
    ms = torch.zeros_like(ms)                            # early on this rank only (graph_matching.py:258-260)
    ms[:, :, 5:7, 5:7] = 1
vgg_f16s = variant.startswith("vgg_")      # "vgg_f16s": config 5's backbone and dtype -- VGG16, 1 channel, fp16 MFMA + fp16 activation storage
pgraphs = variant == "pgraphs"         # graphs="auto": the collective-free pieces replay from HIP graphs from the 3rd step
tr = GraphEchoTrainer(dev, workload=workload, image_size=128, distributed=True, seed=1, clip_len=4,
                      graphs="auto" if pgraphs else False,
                      **({"back_bone": "VGG16", "in_channel": 1, "conv_precision": variant[4:], "seg_loss": "cardiac",
                          "transport_method": "sinkhorn_distance"} if vgg_f16s else {}))
bn = tr.network.back_bone.block_2[1] if vgg_f16s else tr.network.back_bone.bn1
extra = ()
if workload in ("full", "temporal"):   # target-domain frames: a different half of another batch per rank
    xt, _ = synthetic_batch(4, cin, 4, 128, dev, 8)
    extra = (xt[rank * per:(rank + 1) * per],)
if workload == "temporal":             # one source + one target clip of 4 frames per rank
    def clip(seed):
        f, mk = synthetic_batch(4, cin, 4, 128, dev, seed)
        return (f.reshape(1, 4, cin, 128, 128).permute(0, 2, 3, 4, 1).contiguous(),
                mk.reshape(1, 4, 4, 128, 128).permute(0, 2, 3, 4, 1).contiguous())
    cs, cm = clip(20 + rank)
    ct, _ = clip(30 + rank)
    extra = extra + ({"source": cs, "target": ct, "masks": cm},)
losses = [float(tr.step(xs, ms, *extra))]
rm1, rv1 = bn.running_mean.cpu().clone(), bn.running_var.cpu().clone()
losses.append(float(tr.step(xs, ms, *extra)))
for _ in range(3 if pgraphs else 0):   # warm-up calls are eager: capture at the third step, replays after it
    losses.append(float(tr.step(xs, ms, *extra)))
bn._flush_batches()
second = "Grapher" if workload == "fpn_grapher" else "Graph"
torch.save({"flat": tr.optimizers["Net"].fp.flat.cpu(), "gflat": tr.optimizers[second].fp.flat.cpu(),
            "all": {k: o.fp.flat.cpu() for k, o in tr.optimizers.items()},
            "rm1": rm1, "rv1": rv1, "rm": bn.running_mean.cpu(), "rv": bn.running_var.cpu(), "nbt": int(bn.num_batches_tracked),
            "losses": losses, "buckets": len(tr.sync.buckets), "loss_keys": sorted(tr.losses),
            "used": dict(zip(tr.optimizers, tr.sync.agreed_used().values())),
            "local_used": {k: list(o.fp.used) for k, o in tr.optimizers.items()}, "mode": tr.sync.mode,
            "graphs": tr.graphs_in_use(),
            "tgcn_graphs": tr.tgcn.__dict__["_roll_runner"].graphs() if workload == "temporal" and
            "_roll_runner" in tr.tgcn.__dict__ else None,
            "captured": sum(g.graphs()[0] for g in [tr._head] + list(tr._dis.values()))},
           os.path.join(out, f"rank{rank}.pt"))
dist.barrier()
dist.destroy_process_group()
