"""Training-step driver: the build's counterpart of ``Trainer.train``'s inner loop
(reference train_camus_echo.py:183-303, train_cardiac_uda.py:200-320), driving the HIP modules.

One ``step()`` = what the reference does per iteration: FPN on the source batch, segmentation loss
``0.1 * (Dice + BCE) / 2`` (CAMUS form) or ``Dice + BCE`` (CardiacUDA form); optionally FPN on the target batch,
pseudo-labels ``sigmoid > 0.5``, GModule, four Discriminators (x0.1); optionally the temporal branch (clips folded
into the batch, second GModule call, TGCN); one backward; Adam for the FPN and SGD-momentum for every other
module; the LR schedule is the reference's WarmupMultiStepLR stepped per epoch (constant lr/3 in practice).
Workload "fpn_grapher" is BASELINE.json's config 2: ViG ``Grapher`` blocks (k=9, 'mr', gelu, batch-norm,
r = 4/2/1/1) on the four pyramid levels, trained through an auxiliary activation loss (the reference has no
wiring of Grapher into FPN; this harness is defined in DESIGN.md).
"""
import contextlib
import os

import torch
import torch.nn as nn

from . import functional as GF
from . import nn as gnn
from .ddp import GradSynchronizer, broadcast_parameters
from .graphs import GraphedModule
from .models.fpnseg import FPN, Discriminator
from .models.graph_matching import GModule
from .models.TGCN import TGCN
from .models.vig import Grapher
from .optim import FlatAdam, FlatSGD
from .streams import concurrent_stream
from .utils.lr_scheduler import WarmupMultiStepLR
from .utils.sinkhorn_distance import SinkhornDistance

NET_OPT = dict(lr=3e-4, weight_decay=1e-4)                       # train_camus_echo.py:565-572 (Adam)
AUX_OPT = dict(lr=0.0025, momentum=0.9, weight_decay=1e-4)       # :583-626 (SGD for gmn / tgcn / dis)
SCHED = dict(milestones=(90000,), gamma=0.1, warmup_factor=1 / 3, warmup_iters=1000, warmup_method="constant")


class PyramidGraphers(nn.Module):
    """Grapher(256, k=9, conv='mr', act='gelu', norm='batch', r) on p2..p5 with r = (4, 2, 1, 1)."""

    def __init__(self, channels=256, sizes=(64, 32, 16, 8), ratios=(4, 2, 1, 1)):
        super().__init__()
        self.blocks = nn.ModuleList([Grapher(channels, 9, 1, "mr", "gelu", "batch", True, False, 0.0, r, n=s * s)
                                     for s, r in zip(sizes, ratios)])

    def forward(self, pyramid):
        return [blk(p) for blk, p in zip(self.blocks, pyramid)]


# The autograd engine hands CUDA nodes to a per-device worker thread; every Python-defined Function of this package then
# takes the GIL from that thread.  GE_AUTOGRAD_MT=0 runs the backward passes in the calling thread instead.
_AUTOGRAD_MT = os.environ.get("GE_AUTOGRAD_MT", "1") != "0"


class _NetPart(nn.Module):
    """A slice of the FPN's forward as a module of its own (graphs.GraphedModule captures modules); train / eval state is
    the network's."""

    def __init__(self, net):
        super().__init__()
        self.net = net

    @property
    def training(self):
        return self.net.training

    @training.setter
    def training(self, value):
        pass


class _Pyramid(_NetPart):
    def forward(self, x):
        return tuple(self.net.forward_pyramid(x, smooth=False))


class _Head(_NetPart):
    def forward(self, p2, p3, p4, p5):
        return self.net.forward_head([p2, p3, p4, p5], None)


_GM_FIRST = os.environ.get("GE_GM_FIRST", "1") != "0"


class GraphEchoTrainer:
    GRAPHS_AUTO_MAX_FRAMES = 16
    # ... with the fp16 conv paths the GPU side of a step shrinks by 2 - 3x and the host bounds it up to larger batches: config 5
    # in its stated dtype (16 + 32 frames) 39.4 ms eager, 35.7 ms replayed (profiles/r05_small_steps.txt)
    GRAPHS_AUTO_MAX_FRAMES_F16 = 64

    def __init__(self, device, workload="fpn_grapher", back_bone="resnet", in_channel=3, num_classes=4,
                 image_size=256, seg_loss="camus", clip_len=8, distributed=False, seed=0, conv_precision="f32",
                 transport_method="node_discriminate", graphs=False):
        assert workload in ("fpn", "fpn_grapher", "full", "temporal")
        assert conv_precision in ("f32", "f16", "f16s")
        self.conv_precision = conv_precision   # "f16": BASELINE config 5's fp16-MFMA conv path (fp32 storage/accumulate)
        # source / target / clip FPN passes of a step as ONE backbone + top-down pass with per-pass BatchNorm statistics
        # (GF.bn_segments), the segmentation head per pass (only the source logits carry a gradient).  Default: source and
        # target frames always share a pass (one conv launch per layer instead of two: 56.2 vs 57.1 ms at 16+16 frames,
        # 24.2 vs 31-38 ms at 4+4); the temporal workload's clip frames join it only when the step is data parallel (every
        # pass saved divides the SyncBN exchanges: 100 per step instead of 200-300) -- on one GPU GModule's host read then
        # waits for 16 frames instead of 48, and the 32 clip frames' pass keeps the GPU busy while the host issues GModule and
        # the discriminators (separate 80.8, source+target merged 77.5, all merged 84.7 ms).
        # GE_MERGE_PASSES: 0 = separate passes, 1 = one pass for everything, 2 = source + target only.
        mp = os.environ.get("GE_MERGE_PASSES")
        self.merge_passes = mp != "0"
        self.merge_clips = bool(distributed) if mp is None else mp == "1"
        self.device, self.workload, self.seg_loss_kind = device, workload, seg_loss
        self.distributed = distributed
        if distributed and not (torch.distributed.is_available() and torch.distributed.is_initialized()):
            raise RuntimeError("GraphEchoTrainer(distributed=True) needs an initialised torch.distributed process group "
                               "(graphecho_amd.train.init_distributed or dist.init_process_group('nccl', device_id=...))")
        torch.manual_seed(seed)
        self.network = FPN([2, 4, 23, 3], num_classes=num_classes, in_channel=in_channel, back_bone=back_bone).to(device)
        self.modules = {"Net": self.network}
        if workload == "fpn_grapher":
            s = image_size // 4
            self.graphers = PyramidGraphers(256, (s, s // 2, s // 4, s // 8)).to(device)
            self.modules["Grapher"] = self.graphers
        if workload in ("full", "temporal"):
            self.graph_model = GModule(in_channels=256, num_classes=num_classes, device=device).to(device)
            self.modules["Graph"] = self.graph_model
            self.dis = nn.ModuleDict({f"dis_p{i}": Discriminator(grad_reverse_lambda=0.02) for i in (2, 3, 4, 5)}).to(device)
            for k, d in self.dis.items():
                self.modules["Dis_" + k[-2:].upper()] = d
        if workload == "temporal":
            g = image_size // 32
            # transport_method: the reference trainers build TGCN with its default, 'node_discriminate'
            # (train_camus_echo.py:108); 'sinkhorn_distance' is BASELINE config 5's "fp32 Sinkhorn" transport (TGCN.py:282)
            self.tgcn = TGCN(input_dim=256, hidden_dim=256, clip_shape=(clip_len, g, g), soucre_class=10,
                             target_class=10, transport_method=transport_method).to(device)
            self.modules["tgcn_p5"] = self.tgcn
            self.sinkhorn = SinkhornDistance(eps=0.1, max_iter=5, reduction="mean")
        if distributed:
            # SyncBN on a communicator of its OWN (round 6).  Its exchanges are captured inside the pyramid's HIP graphs; on replay they
            # run on the graph's internal streams, i.e. NOT in the issue order of the process group's collective stream -- while the
            # gradient buckets of the discriminators / GModule ride eagerly under the pyramid's backward.  Two collectives of ONE
            # communicator in flight at once, interleaved differently per rank, is undefined in RCCL; two communicators, each used in
            # one consistent order, is the supported form (the kernels of both are co-resident on 256 CUs).  GE_SYNCBN_GROUP=0: WORLD.
            bn_group = None
            if torch.distributed.is_available() and torch.distributed.is_initialized() and \
                    os.environ.get("GE_SYNCBN_GROUP", "1") != "0":
                bn_group = torch.distributed.new_group()       # collective: every rank builds its trainer at the same point
            self._bn_group = bn_group
            gnn.convert_sync_batchnorm(self.network, bn_group)             # train_camus_echo.py:130
        self.optimizers = {}
        for name, m in self.modules.items():
            self.optimizers[name] = FlatAdam(m, **NET_OPT) if name == "Net" else FlatSGD(m, **AUX_OPT)
        self.schedulers = {n: WarmupMultiStepLR(o, **SCHED) for n, o in self.optimizers.items()}
        if distributed:
            broadcast_parameters([o.fp for o in self.optimizers.values()])
        if distributed:
            # exchange order (ddp.GradSynchronizer): modules above the FPN that always receive gradients (discriminators,
            # Graphers) complete first in backward and go first; the FPN's buckets follow as backward walks down it; the
            # modules with data-dependent graphs (GModule's early return, TGCN) go last
            names = list(self.optimizers)
            late = [n for n in names if n in ("Graph", "tgcn_p5")]
            early = [n for n in reversed(names) if n not in late and n != "Net"]
            # Single backward call: the data-dependent models LAST -- a rank whose GModule returned early has no gradient
            # for it, and its buckets must not be exchanged at a rank-dependent point between the SyncBN collectives of
            # the FPN's backward.  Phased backward (_step_phased): GModule / TGCN finish in an autograd call of their own
            # (no SyncBN collective inside), are declared complete at its end on every rank (sync.mark_complete) and go
            # BEFORE the FPN: their exchange rides under the FPN's backward.
            self._order_single = [names.index(n) for n in early + ["Net"] + late]
            self._order_phased = [names.index(n) for n in early + late + ["Net"]]
            self._late = [self.optimizers[n] for n in late]
            self.sync = GradSynchronizer(self.optimizers.values(), launch_order=self._order_single)
        else:
            self.sync = None
        for m in self.modules.values():
            m.train()
        # HIP-graph replay of the FPN passes (graphs.py): pays when the step is bound by the host issuing launches --
        # small per-GPU batches (config 3, config 4 under data parallelism); GE_GRAPHS=0/1 overrides the argument
        ge = os.environ.get("GE_GRAPHS")
        # graphs="auto" (GE_GRAPHS=auto): replay for the full / temporal workloads whenever a step has at most
        # GRAPHS_AUTO_MAX_FRAMES frames -- where the host bounds the eager step on every box of the pool (8 frames: eager
        # 23.5-30.6 ms depending on the box's host, replayed 18.5-20.9; 16 frames: 27.5-34.4 vs 28.5-30.0; 32 frames: eager
        # wins, 47.8 vs 49.7).  Under data parallelism over RCCL "auto" replays EVERYTHING static, the SyncBN backbone with its
        # exchanges captured inside the graphs (RCCL collectives are stream operations: the per-rank 8-frame step is 14.5 ms
        # replayed against 15.5 - 19 ms with the backbone eager, profiles/r06_per_rank_steps_distributed.txt) -- round 6; rounds
        # 4 - 5 kept the backbone eager at N > 1.  GE_GRAPHS_DP=partial restores that (head + discriminators replayed only); a
        # backend whose collectives are not stream operations (the gloo rehearsal) always gets the partial form.  bench.py falls
        # back to eager steps on every rank if any rank's capture is refused.
        mode = graphs if ge is None else {"0": False, "auto": "auto"}.get(ge, True)
        cuda = torch.device(device).type == "cuda"
        self._graphs_auto = mode == "auto" and workload in ("full", "temporal") and cuda
        stream_collectives = bool(distributed) and torch.distributed.is_initialized() and \
            torch.distributed.get_backend() == "nccl" and os.environ.get("GE_GRAPHS_DP", "all") != "partial"
        self._graphs_partial = self._graphs_auto and bool(distributed) and not stream_collectives
        self.use_graphs = (self._graphs_auto or (mode != "auto" and bool(mode))) and cuda
        if self.use_graphs and not self._graphs_partial and distributed and torch.distributed.get_backend() != "nccl":
            # SyncBN's exchanges are captured inside the graphs: only RCCL collectives are stream operations (a gloo
            # rehearsal moves the tensors through the host)
            raise RuntimeError("GraphEchoTrainer(graphs=True) under data parallelism needs the nccl (RCCL) backend")
        self._net = GraphedModule(self.network, [self.optimizers["Net"].fp])
        # the phased step (full / temporal workloads) replays the FPN in two pieces: backbone + top-down pathway, and the
        # segmentation head (with a tape for the source frames, forward-only for the pseudo-label passes)
        self._pyr = GraphedModule(_Pyramid(self.network), [self.optimizers["Net"].fp])
        self._head = GraphedModule(_Head(self.network), [self.optimizers["Net"].fp])
        self._dis = {}
        for k, d in getattr(self, "dis", {}).items():
            self._dis[k] = GraphedModule(d, [self.optimizers["Dis_" + k[-2:].upper()].fp])
        self._set_graphs(self.use_graphs)
        # TGCN's 16-step recurrence replayed from a HIP graph (static shapes, no collective inside: its BatchNorm is
        # local under data parallelism too); GE_TGCN_GRAPH=0: eager
        if workload == "temporal" and cuda and os.environ.get("GE_TGCN_GRAPH", "1") != "0":
            from .models.TGCN import _RollCore

            self.tgcn.__dict__["_roll_runner"] = GraphedModule(_RollCore(self.tgcn, [8, 4, 2, 1]),
                                                              [self.optimizers["tgcn_p5"].fp])
        self.losses = {}    # persists across steps like the reference's dict (train_camus_echo.py:185)
        # conv weight-gradient kernels run on a side stream beside the data-gradient chain (they only feed the
        # optimizer): co-resident kernels de-phase each other's load / MFMA / store phases, +2.4 % on config 2.

        # (Stream priorities were tried both ways -- side stream low, main stream high; range (0, -1) on this stack --
        # and change neither the step time nor how the two streams' kernels stretch each other: DESIGN.md 7b.)
        on = torch.device(device).type == "cuda" and os.environ.get("GE_WGRAD_STREAM", "1") != "0"
        # (every side stream is PROBED to run beside the streams it must overlap: streams.concurrent_stream)
        main_stream = torch.cuda.current_stream(device) if torch.device(device).type == "cuda" else None
        self._wgrad_stream = concurrent_stream(device, [main_stream]) if on else None
        # GModule on a stream of its own beside the head / discriminator passes (phased step, _step_phased); GE_GM_STREAM=0:
        # everything on the main stream
        gm_on = torch.device(device).type == "cuda" and workload in ("full", "temporal") and \
            os.environ.get("GE_GM_STREAM", "1") != "0"
        self._gm_stream = concurrent_stream(device, [main_stream, self._wgrad_stream],
                                            priority=0) if gm_on else None      # (a high-priority GModule stream lost: 530 -> 250 frames/s, docs/HISTORY.md)
        # The discriminators of p3 / p4 / p5 on ONE more stream beside p2's (round 5; GE_DIS_STREAM=0: all four on the main stream).
        # The four are independent of each other; p2's convolutions fill the chip, the three small levels' do not (128 - 512
        # workgroups at 8 + 8 frames) -- their forward and, because autograd runs a node's backward on the stream of its forward,
        # their backward run beside p2's.  One stream, not three: HIP multiplexes streams onto four hardware queues
        # (streams.py), and main / weight-gradient / GModule streams hold three of them.
        dis_on = gm_on and os.environ.get("GE_DIS_STREAM", "1") != "0"
        self._dis_stream = concurrent_stream(device, [main_stream, self._wgrad_stream, self._gm_stream]) if dis_on else None
        # config 2: the Graphers on a stream of their own beside the segmentation head (GE_GRAPHER_STREAM=1; single GPU)
        gr_on = torch.device(device).type == "cuda" and workload == "fpn_grapher" and not distributed and \
            os.environ.get("GE_GRAPHER_STREAM", "0") != "0"
        self._grapher_stream = concurrent_stream(device, [main_stream, self._wgrad_stream]) if gr_on else None
        if self._gm_stream is not None:
            # GModule's backward on its own stream hands gradients to AccumulateGrad nodes created on the main stream: the
            # engine synchronises the two (intended); its one-time warning about it is noise here
            warn_off = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
            if warn_off is not None:
                warn_off(False)
        if self._gm_stream is not None and self.use_graphs:
            from . import graphs as _graphs

            _graphs.FORK_DEFAULT = "0"     # a forked backward graph keeps other streams' kernels waiting (graphs.py)
            # ... except the pyramid's: its backward is the last thing of the phased step, GModule's stream and the head /
            # discriminator passes have been joined by then, so its weight-gradient kernels run as a side branch beside the
            # data-gradient chain (round 6; GE_PYR_FORK=0: one chain)
            if os.environ.get("GE_PYR_FORK", "1") != "0":
                self._pyr.fork = True
        if self.sync is not None:
            # joined before EVERY bucket exchange (also mark_complete's)
            self.sync.side_streams = [s for s in (self._wgrad_stream, self._gm_stream, self._dis_stream) if s is not None]
        # backward cut at the pyramid into three autograd calls (_step_phased): the head / discriminator backward is in
        # the device queue before the host reaches GModule's blocking read
        # (measured, eager mode: 16+16 frames 34.7 -> 33.2 ms, temporal 75.5 -> 73.8; at 4+4 frames the HOST bounds the step
        # -- ~20 ms of Python per step whatever the batch -- and two more engine calls cost 2.5 ms: "auto" = from 12 frames)
        # slab reduces of the split-K weight gradients batched 16 to a launch (GF.DEFER_SLABS)
        self.defer_slabs = os.environ.get("GE_DEFER_SLABS", "1") != "0"
        sb = os.environ.get("GE_SPLIT_BACKWARD", "auto")
        self.split_backward = None if sb == "auto" else sb != "0"

    def _set_graphs(self, on):
        """Switch graph replay on / off; under data parallelism with graphs="auto" only the collective-free pieces."""
        self.use_graphs = bool(on)
        whole = bool(on) and not self._graphs_partial
        self._net.enabled = self._pyr.enabled = whole
        self._head.enabled = bool(on)
        for gm in self._dis.values():
            gm.enabled = bool(on)

    def graphs_in_use(self):
        """False, "all" or "head+discriminators" (bench / logs)."""
        return False if not self.use_graphs else ("head+discriminators" if self._graphs_partial else "all")

    # ---- losses ----------------------------------------------------------------------------------------------
    def seg_loss(self, pred, masks):
        d, b = GF.dice_loss(pred, masks), GF.bce_with_logits(pred, masks)
        return 0.1 * (d + b) / 2 if self.seg_loss_kind == "camus" else d + b

    # ---- one optimisation step -------------------------------------------------------------------------------
    def step(self, imgs_source, masks, imgs_target=None, clips=None):
        """imgs_*: (B, Cin, H, W); masks: (B, nc, H, W) float one-hot.
        clips (temporal): dict(source=(b,C,H,W,T), target=(b,C,H,W,T), masks=(b,nc,H,W,T))."""
        # read by every conv forward of this step; backward follows forward.  "f16s": the fp16 conv path plus fp16
        # ACTIVATION STORAGE inside the VGG16 backbone's conv stacks (graphecho_amd/half.py)
        prev = (GF.CONV_PRECISION, GF.ACT_STORAGE, GF.H_SCALE_MANAGED)
        GF.CONV_PRECISION = "f16" if self.conv_precision == "f16s" else self.conv_precision
        if self.conv_precision == "f16s":
            GF.ACT_STORAGE = "f16"
        if GF.ACT_STORAGE == "f16":                # ("f16s", or GE_ACT_STORAGE=f16 / functional.ACT_STORAGE set by the caller)
            # The device-resident loss scale is created HERE, outside any HIP-graph capture (created lazily by the first
            # backward it could end up inside a captured graph's pool, with its initialisation as a graph node: every replay would
            # reset it), and updated ONLY here, at the step boundary, when every side stream of the previous step has been joined:
            # forward passes in the middle of a step (the clip pyramid after the first backward) leave it alone.
            if imgs_source.is_cuda:
                GF.h_scale(imgs_source.device)
            GF.h_scale_update(all_devices=True)    # from the last step's recorded magnitudes (device side, no host read)
            GF.H_SCALE_MANAGED = True
        try:
            return self._step(imgs_source, masks, imgs_target, clips)
        finally:
            GF.CONV_PRECISION, GF.ACT_STORAGE, GF.H_SCALE_MANAGED = prev

    def _step(self, imgs_source, masks, imgs_target, clips):
        losses = self.losses
        for o in self.optimizers.values():
            o.zero_grad()
        if self.sync:
            self.sync.reset()
        if self._graphs_auto and imgs_target is not None:
            frames = imgs_source.shape[0] + imgs_target.shape[0]
            if clips is not None:
                frames += sum(clips[k].shape[0] * clips[k].shape[-1] for k in ("source", "target"))
            on = frames <= (self.GRAPHS_AUTO_MAX_FRAMES_F16 if self.conv_precision in ("f16", "f16s") else
                            self.GRAPHS_AUTO_MAX_FRAMES)
            if on != self.use_graphs:
                self._set_graphs(on)
        phased = self.split_backward
        if phased is None and imgs_target is not None:
            # with GModule on its own stream the phased step wins at every batch size (its launches overlap the head /
            # discriminator backward that is already queued); without, only from 12 frames
            phased = self._gm_stream is not None or imgs_source.shape[0] + imgs_target.shape[0] >= 12
        phased = self.workload in ("full", "temporal") and imgs_target is not None and (phased or self.use_graphs) \
            and GF.KERNEL_TIMER is None
        if self.sync:     # (batch sizes, hence the choice, are the same on every rank)
            self.sync.set_launch_order(self._order_phased if phased else self._order_single)
            # phased step: GModule / TGCN are declared complete by mark_complete() at the end of their branch, never by their
            # hooks (in the temporal step GModule's parameters receive gradient in two autograd calls)
            self.sync.defer_fps = {id(o.fp) for o in self._late} if phased else set()
        if phased:
            return self._step_phased(imgs_source, masks, imgs_target, clips)
        if self.workload == "fpn_grapher" and self._grapher_stream is not None and GF.KERNEL_TIMER is None:
            return self._step_grapher_streams(imgs_source, masks)
        clip_out = None
        if self.merge_passes and self.workload in ("full", "temporal") and imgs_target is not None:
            # one FPN pass over [source; target; clip frames]: BatchNorm statistics stay per pass (GF.bn_segments), the
            # convolutions get one launch with the whole batch instead of two or three small ones
            inputs = [imgs_source, imgs_target]
            if self.workload == "temporal" and self.merge_clips:
                folded = self._fold_clips(clips)
                inputs.append(folded[0])
            sizes = [v.shape[0] for v in inputs]
            # backbone + top-down pathway over the merged batch; the segmentation head per pass: only the source
            # logits carry a gradient (target / clip logits become pseudo-label maps), so their head runs without a
            # tape -- in one merged head pass its backward would grind through zeros for every non-source frame.  The
            # head's GroupNorm is per sample: splitting it by pass changes nothing.
            # (the smoothing convs belong to the head: forward_pyramid(smooth=False) leaves them to forward_head)
            with GF.bn_segments(sizes):
                feats = self.network.forward_pyramid(torch.cat(inputs), smooth=False)
            feats = [torch.split(f, sizes) for f in feats]
            head = lambda i: self.network.forward_head([f[i] for f in feats], None)
            preds = [head(0)]
            with torch.no_grad():
                preds += [head(i) for i in range(1, len(sizes))]
            pred_s, feat_s = preds[0], [f[0] for f in feats]
            merged_t = (preds[1], [f[1] for f in feats])
            if self.workload == "temporal" and self.merge_clips:
                clip_out = (folded, preds[2], [f[2] for f in feats])
        else:
            merged_t = None
            pred_s, feat_s = self._net(imgs_source, tag="source")
        losses["seg_loss"] = self.seg_loss(pred_s, masks)
        if self.workload == "fpn_grapher":
            outs = self.graphers(feat_s)
            losses["grapher_loss"] = 0.01 * sum(GF.mean_square(o) for o in outs)
        if self.workload in ("full", "temporal"):
            pred_t, feat_t = merged_t if merged_t is not None else self._net(imgs_target, tag="target")
            score_maps = (torch.sigmoid(pred_t) > 0.5).to(pred_t.dtype)
            # GModule hands the pyramids back untouched (graph_matching.py:258-353), so the discriminators do not depend
            # on it: their forward passes are enqueued between GModule's label kernels and its host-side node planning
            # -- the device works through them while the host sits in that (launch-bound) part of the step.
            prep = self.graph_model.prepare((feat_s, feat_t), masks, score_maps)
            adv = {"loss_adv_" + name: 0.1 * self._dis["dis_" + name]((feat_s[lvl], feat_t[lvl]))
                   for lvl, name in enumerate(("p2", "p3", "p4", "p5"))}
            _, _, gm_loss = self.graph_model((imgs_source, imgs_target), (feat_s, feat_t), targets=masks,
                                             score_maps=score_maps, prepared=prep)
            self._update_graph_losses(losses, gm_loss)
            losses.update(adv)
        if self.workload == "temporal":
            losses["temporal_graph_loss"] = self._temporal(clips, clip_out)
        total = sum(losses.values())
        self._backward(total)
        self._finish_step()
        return total.detach()

    @staticmethod
    def _update_graph_losses(losses, gm_loss):
        # GModule returns an EMPTY dict on its < 6 source nodes early return (graph_matching.py:258-260); the
        # reference's dict persists across iterations (train_camus_echo.py:185) and would then re-sum the previous
        # step's graph losses -- tensors of a freed graph: "backward through the graph a second time", and under data
        # parallelism one rank raising while the others wait in a collective.  The stale entries are dropped here.
        # Keys GModule returned are assigned IN PLACE, so from the second step on they keep their position in the dict
        # (= the summation order of the total) exactly as in the reference's loop; only stale ones are removed.
        for k in GModule.LOSS_KEYS:
            if k not in gm_loss:
                losses.pop(k, None)
        losses.update(gm_loss)

    def _backward(self, loss=None, tensors=None, grads=None):
        """loss.backward() (or autograd.backward(tensors, grads)) with the conv weight gradients accumulated straight into
        the flat gradient buffers, on the side stream."""
        if not _AUTOGRAD_MT:
            with torch.autograd.set_multithreading_enabled(False):
                return self._backward_impl(loss, tensors, grads)
        return self._backward_impl(loss, tensors, grads)

    def _backward_impl(self, loss=None, tensors=None, grads=None):
        GF.DIRECT_GRAD_ACCUM = True     # conv wgrad accumulates straight into the flat gradient buffers
        GF.WGRAD_STREAM = self._wgrad_stream
        GF.DEFER_SLABS = self.defer_slabs
        try:
            if loss is not None:
                loss.backward()
            else:
                torch.autograd.backward(tensors, grads)
        finally:
            GF.flush_slab_reduces()
            GF.DIRECT_GRAD_ACCUM = False
            GF.WGRAD_STREAM = None
            GF.DEFER_SLABS = False

    def _finish_step(self):
        if self._wgrad_stream is not None:
            torch.cuda.current_stream().wait_stream(self._wgrad_stream)
        if self._gm_stream is not None:
            torch.cuda.current_stream().wait_stream(self._gm_stream)
        if getattr(self, "_dis_stream", None) is not None:
            torch.cuda.current_stream().wait_stream(self._dis_stream)
        if self._grapher_stream is not None:
            torch.cuda.current_stream().wait_stream(self._grapher_stream)
        if self.sync:
            self.sync.finish()
            self.sync.step_optimizers()      # full steps, or shard steps + parameter all-gather (mode "rs_ag")
        else:
            for o in self.optimizers.values():
                o.step()

    def _step_grapher_streams(self, imgs, masks):
        """Config 2's step with the four Graphers on a stream of their own: pyramid forward; then the Graphers' forward AND
        backward (k-NN, max-relative gathers, 1x1 convs + BatchNorm on maps down to 8 x 8: many small grids) beside the
        segmentation head's forward / loss / backward on the main stream; the two pyramid gradients are added when the
        streams join and the FPN backbone runs its backward once.  Same losses and gradients as the single backward call
        (the pyramid gradient is the same sum, associated differently)."""
        losses = self.losses
        main, gs = torch.cuda.current_stream(), self._grapher_stream
        pyr = self._pyr(imgs, tag="source")
        leaves = [t.detach().requires_grad_(True) for t in pyr]
        gs.wait_stream(main)
        with torch.cuda.stream(gs):
            g_leaves = [t.detach().requires_grad_(True) for t in pyr]
            outs = self.graphers(g_leaves)
            losses["grapher_loss"] = 0.01 * sum(GF.mean_square(o) for o in outs)
            self._backward(losses["grapher_loss"])
        pred = self._head(*leaves, tag="source")
        seg = self.seg_loss(pred, masks)
        losses["seg_loss"] = seg
        # (dict order = summation order of the reported total: seg first, as in _step)
        losses["grapher_loss"] = losses.pop("grapher_loss")
        self._backward(seg)
        main.wait_stream(gs)
        for d, g in zip(leaves, g_leaves):
            if g.grad is not None:
                g.grad.record_stream(main)
                if d.grad is None:
                    d.grad = g.grad
                else:
                    d.grad.add_(g.grad)
        keep = [(a, d.grad) for a, d in zip(pyr, leaves) if d.grad is not None]
        self._backward(tensors=[a for a, _ in keep], grads=[g for _, g in keep])
        self._finish_step()
        return sum(v.detach() for v in losses.values())

    def _step_phased(self, imgs_source, masks, imgs_target, clips):
        """The full / temporal step with its backward pass cut at the pyramid [p2..p5] into three autograd calls:

          1. FPN backbone + top-down pathway forward (all passes); everything ABOVE the pyramid -- segmentation head,
             discriminators, GModule, TGCN -- reads detached copies of the pyramid maps (leaves);
          2. head + discriminator forward, then the backward of (seg loss + adversarial losses) right away: ~40 % of the
             step's kernel time is in the device queue BEFORE the host reaches GModule's one blocking read (the byte labels,
             GModule.prepare) and its ~400 launch-bound small kernels -- the device works through the queue meanwhile
             instead of idling behind the host (at 4+4 frames the step had 3.8 ms of idle gaps >= 20 us around GModule);
          3. GModule (+ temporal branch) forward and backward: gradients add up in the pyramid leaves;
          4. ONE backward of the FPN from the leaves' summed gradients.

        Same losses, same gradients (the pyramid gradient is the same sum, associated differently).  The FPN's, the head's and
        the discriminators' parameters complete in exactly one autograd call, so their gradient buckets (ddp.GradSynchronizer)
        see each AccumulateGrad node once per step as before.  GModule / TGCN are different: in the temporal step with
        GE_GM_FIRST (the default) GModule's parameters receive gradient in TWO autograd calls (the backward of its first call
        runs before its second call).  Their buckets are therefore never declared ready by hooks: FlatParams.notify reports a
        parameter once per step, the synchroniser ignores hook counts for the flat buffers in `sync.defer_fps`, and
        `sync.mark_complete()` at the end of their branch is what releases them -- on every rank, whatever received a gradient
        (tests/test_host_logic.py::test_deferred_buckets_two_autograd_calls).  GE_SPLIT_BACKWARD=0 restores the single
        backward call."""
        losses = self.losses
        net = self.network
        temporal = self.workload == "temporal"
        inputs = [imgs_source, imgs_target]
        folded = self._fold_clips(clips) if temporal else None
        if temporal and self.merge_clips:
            inputs.append(folded[0])
        attached, leaves = [], []       # pyramid maps inside the FPN's graph / their detached stand-ins

        # with HIP graphs (self.use_graphs) the two FPN pieces and the discriminators replay from captured graphs, one
        # host call per piece and direction; GModule / TGCN (data-dependent shapes) stay eager in between
        def pyramid(x, tag, sizes=None):
            if sizes is not None:
                with GF.bn_segments(sizes):
                    f = self._pyr(x, tag=tag)
            else:
                f = self._pyr(x, tag=tag)
            d = [t.detach().requires_grad_(True) for t in f]
            attached.extend(f)
            leaves.extend(d)
            return d

        if self.merge_passes:
            sizes = [v.shape[0] for v in inputs]
            feats = [torch.split(f, sizes) for f in pyramid(torch.cat(inputs), "merged", sizes)]
            per_pass = [[f[i] for f in feats] for i in range(len(sizes))]
        else:
            per_pass = [pyramid(v, t) for v, t in zip(inputs, ("source", "target", "clips"))]
        feat_s, feat_t = per_pass[0], per_pass[1]
        gs = self._gm_stream
        if gs is not None:
            self._branches_beside_main(losses, imgs_source, imgs_target, masks, clips, folded, per_pass, leaves, pyramid,
                                       sizes if self.merge_passes else None)
        else:
            self._branches_in_line(losses, imgs_source, imgs_target, masks, clips, folded, per_pass, pyramid)
        if self.sync:
            self.sync.mark_complete(self._late)
        keep = [(a, d.grad) for a, d in zip(attached, leaves) if d.grad is not None]
        self._backward(tensors=[a for a, _ in keep], grads=[g for _, g in keep])
        self._finish_step()
        return sum(v.detach() for v in losses.values())

    def _branches_in_line(self, losses, imgs_source, imgs_target, masks, clips, folded, per_pass, pyramid):
        """Everything above the pyramid on the main stream (GE_GM_STREAM=0, CPU): head + discriminators and their backward,
        then GModule (+ the temporal branch) and theirs."""
        temporal = self.workload == "temporal"
        feat_s, feat_t = per_pass[0], per_pass[1]
        with torch.no_grad():     # target / clip logits only become pseudo-label maps: no tape
            pred_t = self._head(*feat_t, tag="target")
        score_maps = (torch.sigmoid(pred_t) > 0.5).to(pred_t.dtype)
        prep = self.graph_model.prepare((feat_s, feat_t), masks, score_maps)   # label kernels + their copy to the host
        pred_s = self._head(*feat_s, tag="source")
        losses["seg_loss"] = self.seg_loss(pred_s, masks)
        adv = {"loss_adv_" + name: 0.1 * self._dis["dis_" + name]((feat_s[lvl], feat_t[lvl]))
               for lvl, name in enumerate(("p2", "p3", "p4", "p5"))}
        self._backward(losses["seg_loss"] + sum(adv.values()))
        _, _, gm_loss = self.graph_model((imgs_source, imgs_target), (feat_s, feat_t), targets=masks,
                                         score_maps=score_maps, prepared=prep)
        self._update_graph_losses(losses, gm_loss)
        losses.update(adv)
        second = list(gm_loss.values())
        if temporal:
            if self.merge_clips:
                clip_feats = per_pass[2]
            else:
                clip_feats = pyramid(folded[0], "clips")
            with torch.no_grad():
                pred_c = self._head(*clip_feats, tag="clips")
            losses["temporal_graph_loss"] = self._temporal(clips, (folded, pred_c, clip_feats))
            second.append(losses["temporal_graph_loss"])
        if second:
            self._backward(sum(second))

    def _branches_beside_main(self, losses, imgs_source, imgs_target, masks, clips, folded, per_pass, leaves, pyramid,
                              sizes):
        """GModule -- and, in the temporal workload, the whole temporal branch (second GModule call, the 16-step TGCN
        recurrence, SinkhornDistance) -- on a stream of their own (self._gm_stream) BESIDE the segmentation head and the
        discriminators: ~600 (temporal: ~1 500) small launches of a dozen workgroups each, issued at the pace of the
        host, whose blocking host reads (byte labels, node rows for the seed-bank clustering) then wait for that stream's
        short queue only.  Order of issue: everything dense first (all pyramid passes, the pseudo-label heads, then source
        head + discriminators + their backward: the main stream's queue is full), then the host-paced branches on the
        side stream.  They read the pyramid through detached leaves of their own; those gradients are added to the main
        leaves' when the streams join, before the ONE backward of the pyramid.  TGCN's parameters complete in one autograd
        call; GModule's in one (full workload, GE_GM_FIRST=0) or two (temporal default: the first call's backward runs
        between the two calls).  Under data parallelism their buckets are HELD while the branch runs and released by
        mark_complete() after the join, never by their hooks (see _step_phased)."""
        temporal = self.workload == "temporal"
        gs, main = self._gm_stream, torch.cuda.current_stream()
        feat_s, feat_t = per_pass[0], per_pass[1]
        nl = len(feat_s)
        gm_first = temporal and _GM_FIRST
        early = gm_first and not self.merge_clips      # the clip frames' pyramid is a pass of its own: it can wait
        clip_feats = pred_c = None

        def clip_pass():      # the clip frames' pyramid + pseudo-label head (main stream; the side stream needs them for call 2)
            feats = per_pass[2] if self.merge_clips else pyramid(folded[0], "clips")
            with torch.no_grad():
                return feats, self._head(*feats, tag="clips")

        if temporal and not early:      # issued before the head / discriminators
            clip_feats, pred_c = clip_pass()
        with torch.no_grad():     # target / clip logits only become pseudo-label maps: no tape
            pred_t = self._head(*feat_t, tag="target")
        score_maps = (torch.sigmoid(pred_t) > 0.5).to(pred_t.dtype)
        gs.wait_stream(main)              # pyramid maps, pseudo-label logits
        score_maps.record_stream(gs)
        if pred_c is not None:
            pred_c.record_stream(gs)
        n_first = len(leaves)             # leaves that exist now (temporal, early: source + target levels only)
        with torch.cuda.stream(gs):
            g_leaves = [d.detach().requires_grad_(True) for d in leaves]
            if self.merge_passes:
                gsplit = [torch.split(f, sizes) for f in g_leaves[:nl]]
                g_pass = [[f[i] for f in gsplit] for i in range(len(sizes))]
                if temporal and not self.merge_clips and not early:
                    g_pass.append(g_leaves[nl:])
            else:
                g_pass = [g_leaves[i * nl:(i + 1) * nl] for i in range(len(g_leaves) // nl)]
            prep = self.graph_model.prepare((g_pass[0], g_pass[1]), masks, score_maps)
        if early:
            # Round 5 (config 5 in its stated dtype is paced by the host-side chain GModule -> GModule -> TGCN -> backward): the
            # label kernels of GModule's FIRST call are in the side stream's queue before the 32 clip frames' pyramid pass is
            # issued, so its one blocking read waits for the 16 source / target frames only (it sat behind 7 ms of GPU work)
            clip_feats, pred_c = clip_pass()
            clips_ready = torch.cuda.Event()
            clips_ready.record(main)      # (the side stream waits for THIS, not for the head / discriminator passes queued behind it)
        # ---- main stream: source head, discriminators, their backward
        adv = {}

        def main_block():
            ds = self._dis_stream
            if ds is not None:      # p3 / p4 / p5 beside the head and p2 (forward here, backward on the same stream by autograd's rule)
                ds.wait_stream(main)
                with torch.cuda.stream(ds):
                    for lvl, name in ((1, "p3"), (2, "p4"), (3, "p5")):
                        adv["loss_adv_" + name] = 0.1 * self._dis["dis_" + name]((feat_s[lvl], feat_t[lvl]))
            pred_s = self._head(*feat_s, tag="source")
            losses["seg_loss"] = self.seg_loss(pred_s, masks)
            for lvl, name in enumerate(("p2", "p3", "p4", "p5")):
                if "loss_adv_" + name not in adv:
                    adv["loss_adv_" + name] = 0.1 * self._dis["dis_" + name]((feat_s[lvl], feat_t[lvl]))
            if ds is not None:
                main.wait_stream(ds)
                ordered = {"loss_adv_" + n: adv["loss_adv_" + n] for n in ("p2", "p3", "p4", "p5")}      # the reference's order
                adv.clear()
                adv.update(ordered)
            total = losses["seg_loss"] + sum(adv["loss_adv_" + n] for n in ("p2", "p3", "p4", "p5"))
            self._backward(total)
            if ds is not None:
                main.wait_stream(ds)      # (the engine joins the streams it used; explicit for the flat gradient buffers)

        def side_backward(terms):
            if not terms:
                return
            if self.sync:
                self.sync.hold = True       # buckets of these models are exchanged after the join (mark_complete)
            try:
                self._backward(sum(terms))
            finally:
                if self.sync:
                    self.sync.hold = False

        # Temporal workload (GE_GM_FIRST=0: as the other workloads): GModule's first call goes FIRST -- it submits the seed
        # bank's spectral-clustering fits to the worker processes (6 - 9 ms each), and the second call (on the clip frames) has to
        # wait for them when it completes a missing class from the bank (synthetic data: every step).  Everything the host can
        # do meanwhile is put between the two calls: the BACKWARD of the first call's losses (round 5: its own autograd call --
        # the second call then found the fits done instead of waiting 4 ms for them), the head / discriminator passes and theirs.
        if not gm_first:
            main_block()        # the main stream's queue is full before the host turns to the side stream
        # ---- side stream: GModule (+ temporal branch), forward and backward
        with torch.cuda.stream(gs):
            _, _, gm_loss = self.graph_model((imgs_source, imgs_target), (g_pass[0], g_pass[1]), targets=masks,
                                             score_maps=score_maps, prepared=prep)
            if gm_first:
                side_backward(list(gm_loss.values()))
        if gm_first:
            main_block()
        if early:
            gs.wait_event(clips_ready)    # the clip frames' pyramid and logits
            pred_c.record_stream(gs)
            with torch.cuda.stream(gs):
                g_clip = [d.detach().requires_grad_(True) for d in leaves[n_first:]]
                g_leaves += g_clip
                g_pass.append(g_clip)
        with torch.cuda.stream(gs):
            second = [] if gm_first else list(gm_loss.values())
            t_loss = None
            if temporal:
                t_loss = self._temporal(clips, (folded, pred_c, g_pass[2]))
                second.append(t_loss)
            side_backward(second)
        self._update_graph_losses(losses, gm_loss)
        losses.update(adv)
        if t_loss is not None:
            losses["temporal_graph_loss"] = t_loss
        main.wait_stream(gs)
        for d, g in zip(leaves, g_leaves):        # the side stream's share of the pyramid gradient
            if g.grad is not None:
                g.grad.record_stream(main)
                if d.grad is None:
                    d.grad = g.grad
                else:
                    d.grad.add_(g.grad)

    @staticmethod
    def _fold_clips(clips):
        """(b, C, H, W, T) source + target clips -> frames (b*T, C, H, W), source first; masks likewise."""
        src, tgt, cm = clips["source"], clips["target"], clips["masks"]
        x = torch.cat([src, tgt], dim=0)
        b, c, h, w, t = x.shape
        x = x.permute(0, 4, 1, 2, 3).reshape(-1, c, h, w)
        cm = cm.permute(0, 4, 1, 2, 3).reshape(b * t // 2, -1, h, w).to(x.dtype)
        return x, cm, b, t

    def _temporal(self, clips, done=None):
        """Temporal branch (train_camus_echo.py:232-290): frames folded into the batch, GModule on the clip
        features, TGCN over (b, t) pyramids.  `done`: (folded clips, logits, pyramid) when the FPN pass over the clip
        frames already ran as a segment of the step's merged pass."""
        if done is not None:
            (x, cm, b, t), preds, feats = done
        else:
            x, cm, b, t = self._fold_clips(clips)
            preds, feats = self._net(x, tag="clips")
        half = b * t // 2
        pred_src = preds[:half]
        # frames whose label map is (nearly) empty -- sparsely annotated clips -- hand the PREDICTION to GModule as
        # the target (train_camus_echo.py:253-264, train_cardiac_uda.py:279-290).  The reference also accumulates a
        # Dice+BCE `temp_seg_loss` over the labelled frames there but never adds it to any loss (:286): not computed.
        labelled = cm.sum(dim=(1, 2, 3)) > 100
        src_masks = torch.where(labelled.view(-1, 1, 1, 1), cm, pred_src.detach())
        src_f = [f[:f.shape[0] // 2] for f in feats]
        tgt_f = [f[f.shape[0] // 2:] for f in feats]
        (_, _), (s_nodes, t_nodes), gm_loss = self.graph_model((x[:half], x[half:]), (src_f, tgt_f), targets=src_masks,
                                                                score_maps=preds[half:])
        graph_feats = [f.reshape(b, -1, f.shape[1], f.shape[2], f.shape[3]) for f in feats]
        idx = (torch.zeros(b // 2, dtype=torch.long, device=x.device),) * 2
        # TGCN's own convolutions stay exact fp32 under the fp16 conv modes ("fp16 MFMA conv path + fp32 Sinkhorn" is the FPN's
        # path): its max-relative features grow through the 16-step recurrence -- 7e4 at the sixth step of a config-5 soak,
        # past fp16's 65504, i.e. inf operands and a NaN transport loss (tools/soak.py) -- and 64-node graphs have no use for
        # the fp16 matrix rate.  The backward of a conv follows the precision its forward used.
        prec = GF.CONV_PRECISION
        GF.CONV_PRECISION = "f32"
        try:
            tg_loss = self.tgcn(graph_feats, (s_nodes.clone().detach(), t_nodes.clone().detach()), self.sinkhorn,
                                nn.CrossEntropyLoss(), idx, r=[8, 4, 2, 1])
        finally:
            GF.CONV_PRECISION = prec
        self.last_temporal = {"tgcn": tg_loss, "graph": gm_loss}  # the branch's own terms (logging / tests)
        return sum(tg_loss.values()) + sum(gm_loss.values())     # train_camus_echo.py:286

    def end_epoch(self):
        for s in self.schedulers.values():
            s.step()

    # ---- validation metric (train_camus_echo.py:350-417) -----------------------------------------------------
    @torch.no_grad()
    def overlap_metrics(self, gt, pred, eps=1e-5):
        o, t = pred.reshape(-1).float(), gt.reshape(-1).float()
        tp, fp = (o * t).sum(), (o * (1 - t)).sum()
        fn, tn = ((1 - o) * t).sum(), ((1 - o) * (1 - t)).sum()
        return ((tp + tn + eps) / (tp + tn + fp + fn + eps), (2 * tp + eps) / (2 * tp + fp + fn + eps),
                (tp + eps) / (tp + fp + eps), (tn + eps) / (tn + fp + eps), (tp + eps) / (tp + fn + eps))

    # ---- checkpoint format of the reference: {'network': state_dict} -> net_%05d.pth + latest.ckpt ----------
    def save(self, save_dir, epoch):
        """``net_<id>.pth`` holding {'network': state_dict}; ``latest.ckpt`` holds the zero-padded id only -- the
        reference's load() rebuilds the file name from it (train_camus_echo.py:449-459, 472-489)."""
        os.makedirs(save_dir, exist_ok=True)
        ckpt_id = str(epoch).zfill(5)
        path = os.path.join(save_dir, "net_" + ckpt_id + ".pth")
        torch.save({"network": {k: v.cpu() for k, v in self.network.state_dict().items()}}, path)
        with open(os.path.join(save_dir, "latest.ckpt"), "w") as f:
            f.write(ckpt_id + "\n")
        return path

    def load_states(self, states):
        """{model name as in self.modules ("Net", "Graph", "Dis_P2", ...): state_dict} -> in-place load of those models
        (the values land in the flat parameter buffers), packed conv operands refreshed."""
        for name, sd in states.items():
            self.modules[name].load_state_dict(sd)
        GF.bump_param_epoch()
        for name in states:
            o = self.optimizers[name]
            o.fp.version += 1
            if o.packer is not None:
                o.packer.repack()

    def load(self, path):
        """`path`: a ``net_<id>.pth`` file, or a checkpoint directory (resolved through its ``latest.ckpt``)."""
        if os.path.isdir(path):
            with open(os.path.join(path, "latest.ckpt")) as f:
                path = os.path.join(path, "net_" + f.read().splitlines()[-1].strip() + ".pth")
        sd = torch.load(path, map_location="cpu")["network"]
        # as the reference does (train_camus_echo.py:467): every "module." removed, entries the network does not have are
        # dropped (checkpoints with extra buffers load), and what is left goes through a strict load_state_dict -- it
        # copies in place, so the values land in the flat buffer the parameters alias; it reports missing / mis-shaped
        # keys and runs BatchNorm2d._load_from_state_dict (pending-count reset)
        own = self.network.state_dict()
        sd = {k.replace("module.", ""): v for k, v in sd.items() if k.replace("module.", "") in own}
        self.network.load_state_dict(sd)
        GF.bump_param_epoch()
        for o in self.optimizers.values():
            o.fp.version += 1
            if o.packer is not None:
                o.packer.repack()


def synthetic_batch(batch, in_channel, num_classes, size, device, seed):
    """Seeded synthetic frames in [0,1] and one-hot-ish masks with every class present."""
    g = torch.Generator().manual_seed(seed)
    imgs = torch.rand(batch, in_channel, size, size, generator=g)
    masks = torch.zeros(batch, num_classes, size, size)
    boxes = [(0.16, 0.47, 0.2, 0.55), (0.39, 0.78, 0.39, 0.86), (0.59, 0.97, 0.12, 0.35), (0.08, 0.35, 0.63, 0.94)]
    jit = torch.randint(-size // 16, size // 16 + 1, (batch, num_classes, 2), generator=g)
    for b in range(batch):
        for c in range(num_classes):
            y0, y1, x0, x1 = boxes[c % 4]
            dy, dx = int(jit[b, c, 0]), int(jit[b, c, 1])
            ya, yb = max(0, int(y0 * size) + dy), min(size, int(y1 * size) + dy)
            xa, xb = max(0, int(x0 * size) + dx), min(size, int(x1 * size) + dx)
            masks[b, c, ya:yb, xa:xb] = 1.0
    return imgs.to(device), masks.to(device)
