"""Worker of tests/test_half_gpu.py::test_sync_batchnorm_world2_fp16_storage: one of two ranks sharing cuda:0 (gloo).

A VGG16 backbone under functional.ACT_STORAGE = "f16" with SyncBN: this rank's half of a fixed batch forward + backward;
saves its features and parameter gradients.  usage: ddp_half_worker.py RANK WORLD PORT OUTDIR"""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
os.environ["MASTER_ADDR"] = "127.0.0.1"
os.environ["MASTER_PORT"] = port
dist.init_process_group("gloo", rank=rank, world_size=world)
from graphecho_amd import functional as GF
from graphecho_amd import nn as gnn
from graphecho_amd.models.fpnseg import VGG16

dev = torch.device("cuda:0")
torch.manual_seed(11)
net = gnn.convert_sync_batchnorm(VGG16(1).to(dev).train())
gen = torch.Generator().manual_seed(12)
x = torch.randn(4, 1, 128, 128, generator=gen).to(dev)
per = x.shape[0] // world
GF.ACT_STORAGE = "f16"
if rank == 1:
    GF.H_GRAD_SCALE *= 8.0      # the ranks' loss scales need not agree: what SyncBN exchanges is in true units
feats = net(x[rank * per:(rank + 1) * per])
proj = [(torch.randn(4, *f.shape[1:], generator=gen) / (4 * f[0].numel()) ** 0.5).to(dev) for f in feats]
sum((f * r[rank * per:(rank + 1) * per]).sum() for f, r in zip(feats, proj)).backward()
GF.ACT_STORAGE = "f32"
torch.save({"feats": [f.detach().cpu() for f in feats], "grads": {n: p.grad.cpu() for n, p in net.named_parameters()},
            "rm": net.block_2[1].running_mean.cpu(), "sync": list(GF.SYNC_BN_STATS)}, os.path.join(out, f"rank{rank}.pt"))
dist.barrier()
dist.destroy_process_group()
