"""Training / validation / checkpoint loop: the build's counterpart of ``Trainer.train`` and ``Trainer.validation``
(reference train_camus_echo.py:183-417, 452-515) for the part that follows the data loaders.

``run(config, source, target=None, val=None)`` takes the reference's nested config dict (keys used:
``train.batch_size / num_epochs / save_dir / graph_matching / discriminator / temporal_graph``, ``net.opt.lr``,
``*.sch``) and iterables of RAW batches -- ``(frames uint8 (N,C,H,W[,T]), label maps uint8 (N,H,W[,T]))`` -- which are
formatted on the GPU (`graphecho_amd.data`: nearest resize to ``spatial_size``, random / centre crop to ``crop_size``,
/255, one-hot, clip fold) instead of on DataLoader workers with MONAI.  Per epoch: training steps, scheduler step,
validation Dice per class from device-side TP/FP/FN/TN counts, checkpoint ``{'network': state_dict}`` as
``net_%05d.pth`` + ``latest.ckpt`` (train_camus_echo.py:472-489).  ``python -m graphecho_amd.train`` runs it on seeded
synthetic raw data (the datasets themselves are out of scope).
"""
import argparse
import json
import os
import time

import torch

from . import data as gdata
from .trainer import GraphEchoTrainer

DEFAULT_CONFIG = {   # the subset of train_camus_echo.py:551-637 this loop reads, with the reference's values
    "train": {"batch_size": 8, "num_epochs": 400, "graph_matching": True, "discriminator": True,
              "temporal_graph": False, "seg_parts": True, "save_dir": "./result/model/seg/view_4",
              "spatial_size": 328, "crop_size": 256, "class_values": (0, 1, 2, 3), "in_channel": 3},
    "net": {"opt": {"opt_name": "Adam", "lr": 3e-4, "weight_decay": 1e-4}},
}


class SyntheticRawSet:
    """Seeded raw batches shaped like the decoded datasets: uint8 frames and uint8 label maps with one blob per
    foreground class (every class present)."""

    def __init__(self, n_batches, batch, in_channel, num_classes, hw=(300, 400), seed=0, device="cuda", contrast=0):
        self.n, self.b, self.c, self.nc, self.hw, self.seed, self.device = n_batches, batch, in_channel, num_classes, hw, seed, device
        self.contrast = contrast     # > 0: class k is `contrast * k` grey levels brighter than speckle (a learnable task)

    def __len__(self):
        return self.n

    def __iter__(self):
        H, W = self.hw
        for i in range(self.n):
            g = torch.Generator().manual_seed(self.seed * 100003 + i)
            frames = torch.randint(0, 256, (self.b, self.c, H, W), generator=g, dtype=torch.uint8)
            labels = torch.zeros((self.b, H, W), dtype=torch.uint8)
            for b in range(self.b):
                for k in range(1, self.nc):
                    cy = int(torch.randint(H // 5, 4 * H // 5, (1,), generator=g))
                    cx = int(torch.randint(W // 5, 4 * W // 5, (1,), generator=g))
                    labels[b, max(0, cy - H // 8):cy + H // 8, max(0, cx - W // 8):cx + W // 8] = k
            if self.contrast:
                frames = (frames // 3 + (self.contrast * labels).unsqueeze(1)).clamp(max=255).to(torch.uint8)
            yield frames.to(self.device), labels.to(self.device)


def _format(frames, labels, cfg, train, generator):
    """Raw uint8 frames + label maps -> network inputs; lists (samples of different source sizes, as the on-disk
    datasets yield them) are formatted one sample at a time and concatenated."""
    if isinstance(frames, (list, tuple)):
        parts = [_format(f, m, cfg, train, generator) for f, m in zip(frames, labels)]
        return torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts])
    t = cfg["train"]
    S, crop = t["spatial_size"], t["crop_size"]
    offs = gdata.random_crop_origins(frames.shape[0], S, crop, generator) if train else None
    x = gdata.prepare_frames(frames, S, crop, offsets=offs, center=not train)
    m = gdata.onehot_labels(labels, t["class_values"], S, crop, offsets=offs, center=not train)
    return x, m


@torch.no_grad()
def validate(trainer, val, cfg):
    """Dice / pixel-acc / precision / specificity / recall per class over `val` (train_camus_echo.py:305-417)."""
    # `val_planes`: how many leading one-hot planes the validation set really labels -- the reference validates on
    # `masks[:, :1]` / `pred[:, :1]` (train_camus_echo.py:358,365): EchoNet traces the LV only, a second (LA) plane would
    # be empty in every mask and report a Dice of eps/eps
    nc = int(cfg["train"].get("val_planes") or len(cfg["train"]["class_values"]))
    meter = gdata.OverlapMeter(nc, trainer.device)
    net = trainer.network
    was_training = net.training
    net.eval()
    for frames, labels in val:
        x, m = _format(frames, labels, cfg, False, None)
        pred, _ = net(x)
        meter.update(pred[:, :nc].contiguous(), m[:, :nc].contiguous())
    net.train(was_training)
    return meter.metrics()


def init_distributed():
    """One process per GPU under torchrun (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment): binds the
    process to its GPU and opens the RCCL process group.  Returns (rank, world, device).  The reference builds its DDP
    wrappers without ever initialising a group on the single-node path; without this call the
    ranks would train independent replicas on 1/N of the data."""
    import torch.distributed as dist

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL across processes); no-op once HIP is up
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("GE_DIST_BACKEND", "nccl")     # "gloo": rehearsal of the N > 1 path on a 1-GPU box (tests)
    if backend != "nccl":
        local %= max(1, torch.cuda.device_count())
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    return rank, world, dev


def run(config, source, target=None, val=None, device=None, distributed=False, log=print):
    cfg = {k: dict(v) for k, v in DEFAULT_CONFIG.items()}
    for k, v in (config or {}).items():
        cfg.setdefault(k, {}).update(v)
    t = cfg["train"]
    device = device or torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    full = bool(t["graph_matching"] or t["discriminator"]) and target is not None
    trainer = GraphEchoTrainer(device, workload="full" if full else "fpn", in_channel=t["in_channel"],
                               num_classes=len(t["class_values"]), image_size=t["crop_size"], distributed=distributed,
                               seg_loss=t.get("seg_loss", "camus"), graphs=bool(t.get("hip_graphs", False)))
    gen = torch.Generator().manual_seed(1234 + int(os.environ.get("RANK", "0")))
    history = []
    for epoch in range(t["num_epochs"]):
        t0, frames_seen, loss = time.time(), 0, None
        tgt_iter = iter(target) if full else None
        for frames, labels in source:
            x, m = _format(frames, labels, cfg, True, gen)
            xt = None
            if full:
                try:
                    ft, lt = next(tgt_iter)
                except StopIteration:
                    tgt_iter = iter(target)
                    ft, lt = next(tgt_iter)
                xt, _ = _format(ft, lt, cfg, True, gen)
            loss = trainer.step(x, m, xt)
            frames_seen += x.shape[0] + (0 if xt is None else xt.shape[0])
        trainer.end_epoch()
        torch.cuda.synchronize(device)
        rec = {"epoch": epoch, "loss": None if loss is None else float(loss), "frames_per_s": frames_seen / (time.time() - t0)}
        if val is not None:
            rec["dice"] = [round(float(d), 4) for d in validate(trainer, val, cfg)["dice"]]
        if int(os.environ.get("RANK", "0")) == 0:
            rec["checkpoint"] = trainer.save(t["save_dir"], epoch)
            log(json.dumps(rec))
        history.append(rec)
    return trainer, history


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--batches", type=int, default=4)
    ap.add_argument("--batch-size", type=int, default=8)
    ap.add_argument("--save-dir", default="./result/model/seg/synthetic")
    ap.add_argument("--fpn-only", action="store_true")
    ap.add_argument("--graphs", action="store_true", help="replay the FPN / discriminator passes from HIP graphs")
    ap.add_argument("--camus", default=None, help="CAMUS root (contains training/<patient>/*.mhd): train on it instead "
                    "of synthetic frames (FPN only, 1 input channel, LV/LA planes)")
    ap.add_argument("--camus-view", default="4CH_ED")
    ap.add_argument("--echonet", default=None, help="EchoNet-Dynamic root (FileList.csv, VolumeTracings.csv, Videos/): "
                    "with --camus, the target domain of the CAMUS -> EchoNet adaptation (graph matching + discriminators "
                    "on; validation Dice on EchoNet's LV tracings)")
    ap.add_argument("--uda-infos", default=None, help="CardiacUDA infos.npy: source Site_G -> target Site_R, view 4")
    a = ap.parse_args()
    cfg = {"train": {"num_epochs": a.epochs, "batch_size": a.batch_size, "save_dir": a.save_dir,
                     "graph_matching": not a.fpn_only, "discriminator": not a.fpn_only, "hip_graphs": a.graphs}}
    rk, ws, dev = init_distributed()
    if a.camus:
        from .datasets import CamusSet, RawBatches
        tr, va = (CamusSet(a.camus, a.camus_view, a.camus_view + "_gt", s) for s in ("train", "valid"))
        cfg["train"].update(in_channel=1, class_values=tr.class_values, graph_matching=False, discriminator=False,
                            spatial_size=272, crop_size=256)      # camus.py:42 img_res / img_crop
        tgt, val = None, RawBatches(va, a.batch_size, dev)
        if a.echonet:                                              # train_camus_echo.py:146-176: source CAMUS, target EchoNet
            from .datasets import EchoFrames, EchoSet
            cfg["train"].update(graph_matching=not a.fpn_only, discriminator=not a.fpn_only, spatial_size=124, crop_size=112,
                                val_planes=1)     # EchoNet labels the LV plane only (train_camus_echo.py:358)
            tgt = RawBatches(EchoFrames(EchoSet(a.echonet, "train")), a.batch_size, dev, shuffle=True, drop_last=True,
                             rank=rk, world=ws)
            val = RawBatches(EchoFrames(EchoSet(a.echonet, "val")), a.batch_size, dev)
        run(cfg, RawBatches(tr, a.batch_size, dev, shuffle=True, drop_last=True, rank=rk, world=ws), tgt, val,
            distributed=ws > 1)
        return
    if a.uda_infos:
        import numpy as np
        from .datasets import CardiacUDASet, RawBatches
        infos = np.load(a.uda_infos, allow_pickle=True).item()       # train_cardiac_uda.py:49
        root = os.path.dirname(a.uda_infos)
        src_set = CardiacUDASet(infos, root, True, set_select=("Site_G",), view_num=("4",))
        tgt_set = CardiacUDASet(infos, root, True, set_select=("Site_R",), view_num=("4",))
        val_set = CardiacUDASet(infos, root, False, data_list=tgt_set.test_list, set_select=("Site_R",), view_num=("4",))
        cfg["train"].update(in_channel=1, class_values=src_set.class_values)
        mk = lambda d, sh: RawBatches(d, a.batch_size, dev, shuffle=sh, drop_last=sh, rank=rk if sh else 0,
                                      world=ws if sh else 1)
        run(cfg, mk(src_set, True), None if a.fpn_only else mk(tgt_set, True), mk(val_set, False), distributed=ws > 1)
        return
    src = SyntheticRawSet(a.batches, a.batch_size, 3, 4, seed=1 + 1000 * rk, device=dev)
    tgt = None if a.fpn_only else SyntheticRawSet(a.batches, a.batch_size, 3, 4, seed=2 + 1000 * rk, device=dev)
    run(cfg, src, tgt, SyntheticRawSet(2, a.batch_size, 3, 4, seed=3, device=dev), device=dev, distributed=ws > 1)


if __name__ == "__main__":
    main()
