"""Data-parallel gradient exchange over RCCL/xGMI (one process per GPU, torch.distributed backend "nccl").

The reference wraps its networks in DistributedDataParallel + SyncBatchNorm (train_camus_echo.py:129-142) but
ships that path disabled and mis-wired.  Here the exchange is built on the flat gradient
buffers of graphecho_amd.optim.FlatParams:

  * the flat buffer is cut into a few large buckets (default 32 MiB: xGMI is point-to-point, 7 links x ~153 GB/s
    per GPU, so a ring step is per-link bound and wants large messages);
  * a bucket's SUM all-reduce is launched asynchronously from the parameters' AccumulateGrad hooks as soon as its
    last gradient has landed, so it overlaps the rest of backward -- but always in ONE fixed order (model by model --
    the trainer puts the discriminators / Graphers first, then the FPN, then the data-dependent GModule / TGCN --, each
    model's buckets from the end of its buffer to the start, which is the order backward completes them in):
    a bucket that becomes ready early waits for its predecessors.  Ranks whose data-dependent graphs (GModule)
    finish parameters in a different order would otherwise pair different buckets in the same collective;
  * the mean is folded into the optimizer kernel (grad_scale = 1/world), no extra pass over the gradients;
  * parameters that received no gradient (TGCN.prediction.*, GModule's early return) contribute zeros, i.e.
    find_unused_parameters=True semantics without a graph walk; buckets still pending at the end of backward
    are flushed by ``finish()``;
  * which parameters the optimizers step is agreed EVERY step: the ranks OR their "received a gradient" bit maps
    (data-dependent: GModule returns early without losses on a rank whose batch yields < 6 nodes,
    graph_matching.py:258-260) over a host-side gloo group -- CPU tensors, so the exchange never touches the GPU
    streams and costs no device synchronisation.  A parameter that got a gradient on any rank is stepped on all of
    them (with the averaged gradient); one that got none anywhere is skipped like torch's ``grad is None``.
GModule is synchronised too (the reference forgets to wrap it, which would let replicas diverge).

Two exchange modes (``GradSynchronizer(mode=...)``, env ``GE_DDP_MODE``):
  * ``"allreduce"`` (default): every bucket is SUM-all-reduced, every rank runs the whole optimizer step;
  * ``"rs_ag"``: every bucket is SUM-reduce-scattered -- rank r receives the r-th 1/world of each bucket --, the
    rank runs the optimizer on its shards only (1/world of the Adam / SGD traffic) and the updated shards are
    all-gathered back into every rank's parameter buffer: the same bytes on the xGMI links as a ring all-reduce,
    split into its two halves with the optimizer in between.  Buckets are cut at multiples of the world size
    (FlatParams pads its buffers), so a parameter may straddle two buckets and counts towards both.
``comm_stats`` holds what the last step exchanged (collective count, bytes) for bench.py's report.
"""
import os

import torch
import torch.distributed as dist


class GradSynchronizer:
    def __init__(self, optimizers, bucket_bytes=32 << 20, group=None, mode=None, launch_order=None):
        self.group = group
        self.mode = mode or os.environ.get("GE_DDP_MODE", "allreduce")
        if self.mode not in ("allreduce", "rs_ag"):
            raise ValueError(f"GradSynchronizer: unknown mode {self.mode!r}")
        self.comm_stats = {"collectives": 0, "bytes": 0}
        self.force = False                       # run the collectives even at world size 1 (single-GPU self-test)
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self._host_group = None                  # gloo group for the per-step "used" bit maps (host tensors)
        self._device_agree = False               # fallback: bit maps through the device group + a host read
        self.opts = list(optimizers)
        # CUDA models: the agreement rides on the DEVICE -- the local flags go up in one small pinned copy, are MAX-all-
        # reduced on the gradient group behind the last bucket (no host read, no second transport) and the optimizer
        # kernels read them (ge_*_step_masked).  The host exchange is kept for CPU tensors (host-logic tests) and as
        # GE_DDP_USED=host (every rank must set it alike: opening the host group is a collective).  With the device path
        # the host-side views -- FlatParams.used, the optimizers' `started` lists -- stay RANK-LOCAL; agreed_used() reads
        # the agreed map.
        self._dev_used = all(o.fp.flat.is_cuda for o in self.opts) and os.environ.get("GE_DDP_USED", "device") != "host"
        if dist.is_available() and dist.is_initialized() and not self._dev_used:
            if dist.get_backend(group) == "gloo":
                self._host_group = group if group is not None else dist.group.WORLD
            else:   # collective call: every rank constructs its synchroniser at the same point
                try:
                    self._host_group = dist.new_group(ranks=dist.get_process_group_ranks(group) if group is not None
                                                      else None, backend="gloo")
                except Exception:    # no usable host transport on THIS rank
                    self._host_group = None
                # the ranks must take the same path: one that fell back alone would wait in a device all-reduce while
                # the others sit in the gloo one.  Agree over the device group (it exists): gloo only if everyone has it.
                ok = torch.tensor([1 if self._host_group is not None else 0], dtype=torch.int32,
                                  device=torch.device("cuda", torch.cuda.current_device()))
                dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
                if int(ok.item()) == 0:      # agree through the device group instead (one host read per step)
                    self._host_group = None
                    self._device_agree = True
        self.used_syncs = 0                      # number of bit-map agreements made (tests)
        if self._dev_used:
            n = sum(len(o.fp.params) for o in self.opts)
            dev = self.opts[0].fp.flat.device
            self._flags_dev = torch.zeros(n, device=dev)
            self._started_dev = torch.zeros(n, device=dev)
            self._flags_host = [torch.zeros(n).pin_memory(), torch.zeros(n).pin_memory()]   # alternate: a copy may be in flight
            self._flags_events = [None, None]    # recorded behind each buffer's upload, waited for before its reuse
            self._flag_slices, lo = [], 0
            for o in self.opts:
                k = len(o.fp.params)
                self._flag_slices.append((lo, lo + k))
                lo += k
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        self.buckets = []     # (flat_params, start, end, [param indices])
        self._of_param = {}   # (id(fp), param index) -> [bucket ids the parameter overlaps]
        shard_w = self.world if self.mode == "rs_ag" else 1
        for opt in self.opts:
            opt.grad_scale = 1.0 / self.world
            fp = opt.fp
            per = max(1, bucket_bytes // 4)
            # the bucket at the START of the buffer holds the first layers: its gradients land last and its all-reduce
            # is the only one nothing can hide, so it is kept small (4 MiB)
            first = max(1, min(per, (4 << 20) // 4))
            if shard_w > fp.PAD:
                raise ValueError("rs_ag: world size exceeds FlatParams.PAD")
            cur_start = 0
            for i, (p, o) in enumerate(zip(fp.params, fp.offsets)):
                end = o + p.numel()
                last = i == len(fp.params) - 1
                if end - cur_start >= (first if cur_start == 0 else per) or last:
                    # sharded mode: bucket sizes are multiples of the world size; the cut may fall inside parameter i
                    # (which then also belongs to the next bucket) or, for the last bucket, in the zero padding
                    cut = end if shard_w == 1 else cur_start + -(-(end - cur_start) // shard_w) * shard_w if last \
                        else cur_start + ((end - cur_start) // shard_w) * shard_w
                    if cut <= cur_start:
                        continue
                    idx = [j for j, (q, oq) in enumerate(zip(fp.params, fp.offsets))
                           if oq < cut and oq + q.numel() > cur_start]
                    bid = len(self.buckets)
                    self.buckets.append((fp, cur_start, cut, idx))
                    for j in idx:
                        self._of_param.setdefault((id(fp), j), []).append(bid)
                    cur_start = cut
            fp.listeners.append(self._make_listener(fp))
        self._shard_buf = {}  # bucket id -> this rank's reduced shard (rs_ag)
        # streams of the owner that produce gradients beside the main one (the trainer sets them: conv weight-gradient
        # stream, GModule's stream); all are joined before every exchange
        self.side_stream = None
        self.side_streams = []
        cuda = all(o.fp.flat.is_cuda for o in self.opts)
        self.home_stream = torch.cuda.current_stream(self.opts[0].fp.flat.device) if cuda else None
        # hold = True: gradient hooks only mark buckets ready; nothing is exchanged until mark_complete() / finish()
        # drains them (the trainer sets it around a backward pass that runs on a side stream)
        self.hold = False
        # flat buffers whose buckets become ready ONLY through mark_complete(): models that receive gradient in more than one
        # autograd call of a step (GModule in the temporal step: the backward of its first call runs before its second call) --
        # a bucket whose parameters have all been seen once must not be exchanged while a later call still adds to it
        self.defer_fps = set()
        # fixed launch order: optimizers in `launch_order` (indices into `optimizers`; default: as given), each one's
        # buckets last-to-first.  A bucket waits for its predecessors in this order, so models whose gradients are
        # complete early in backward (and on every rank, every step) belong in front, a model whose graph is
        # data-dependent (GModule: may get no gradient at all on a rank) at the end, where it cannot hold anyone up.
        self.set_launch_order(launch_order)
        self._pending = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._ready = [False] * len(self.buckets)
        self._next = 0
        self._works = []
        self.reset()

    def set_launch_order(self, launch_order=None):
        """Fix the order the buckets are exchanged in (between steps only): optimizers in `launch_order`, each one's
        buckets last-to-first.  EVERY rank must use the same order in the same step, and the order must not let a
        data-dependent model (GModule: no gradient at all on a rank that returned early) be exchanged at a
        rank-dependent point between other collectives of the group (SyncBN's) -- see trainer._step / _step_phased."""
        first, lo = {}, 0
        for k, opt in enumerate(self.opts):
            n = sum(1 for b in self.buckets if b[0] is opt.fp)
            first[k] = (lo, n)
            lo += n
        seq = list(launch_order) if launch_order is not None else list(range(len(self.opts)))
        if sorted(seq) != list(range(len(self.opts))):
            raise ValueError("GradSynchronizer: launch_order must be a permutation of the optimizer indices")
        self._order = []
        for k in seq:
            lo, n = first[k]
            self._order += list(range(lo + n - 1, lo - 1, -1))

    def _make_listener(self, fp):
        def on_grad(i):
            if self.world == 1 and not self.force:
                return
            if id(fp) in self.defer_fps:
                # a model that receives gradient in several autograd calls of the step (GModule in the temporal step): its
                # buckets are declared ready by mark_complete() alone; hook counts mean nothing here and are not kept
                if not self.hold:
                    self._drain()
                return
            for bid in self._of_param[(id(fp), i)]:
                if self._pending[bid] > 0:
                    self._pending[bid] -= 1
                    if self._pending[bid] == 0:
                        self._ready[bid] = True
            if not self.hold:
                self._drain()
        return on_grad

    def mark_complete(self, optimizers):
        """No further gradient will reach these optimizers' parameters in this step (the caller finished the autograd
        call(s) that could produce them -- trainer._step_phased): their buckets count as ready, whatever received a
        gradient, and are exchanged now, in the fixed order, instead of at finish()."""
        if self.world == 1 and not self.force:
            return
        fps = {id(o.fp) for o in optimizers}
        for bid, b in enumerate(self.buckets):
            if id(b[0]) in fps:
                self._ready[bid] = True
        self._drain()

    def _drain(self, everything=False):
        """Launch, in the fixed order, every bucket up to the first one that is not ready yet."""
        while self._next < len(self._order) and (everything or self._ready[self._order[self._next]]):
            self._launch(self._order[self._next])
            self._next += 1

    def _launch(self, bid):
        if self._launched[bid]:
            return
        fp, a, b, _ = self.buckets[bid]
        self._launched[bid] = True
        from . import functional as GF

        GF.flush_slab_reduces()           # deferred slab reduces: the bucket's gradients must be complete
        # weight gradients of this bucket (and the slab reduces just flushed) may still be in flight on the conv
        # weight-gradient side stream: wait for it ALWAYS, not only while a backward call has GF.WGRAD_STREAM set --
        # mark_complete() launches buckets between autograd calls, when the trainer has already reset it
        sides = list(self.side_streams)
        if self.side_stream is not None:
            sides.append(self.side_stream)
        if GF.WGRAD_STREAM is not None and GF.WGRAD_STREAM not in sides:
            sides.append(GF.WGRAD_STREAM)
        if sides or self.home_stream is not None:
            # A bucket can be launched from inside an autograd hook, where the current stream is whatever stream the node
            # runs on: a backward pass on a side stream (GModule's) hands some gradients to AccumulateGrad nodes that add
            # them on the HOME stream (the one the flat buffers were built on) -- the exchange must see those adds too.
            cur = torch.cuda.current_stream()
            for side in sides + ([self.home_stream] if self.home_stream is not None else []):
                if side != cur:
                    cur.wait_stream(side)
        self.comm_stats["collectives"] += 1
        self.comm_stats["bytes"] += 4 * (b - a)
        if self.mode == "allreduce":
            self._works.append(dist.all_reduce(fp.grad[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            return
        n = (b - a) // self.world
        buf = self._shard_buf.get(bid)
        if buf is None:
            buf = self._shard_buf[bid] = torch.empty(n, device=fp.grad.device, dtype=fp.grad.dtype)
        self._works.append(dist.reduce_scatter_tensor(buf, fp.grad_padded[a:b], op=dist.ReduceOp.SUM, group=self.group,
                                                      async_op=True))

    def owned_ranges(self, fp):
        """[lo, hi) element ranges of `fp`'s buffers this rank reduces, updates and publishes (rs_ag); None otherwise."""
        if self.mode != "rs_ag":
            return None
        out = []
        for f, a, b, _ in self.buckets:
            if f is fp:
                n = (b - a) // self.world
                out.append((a + self.rank * n, a + (self.rank + 1) * n))
        return out

    def step_optimizers(self):
        """Optimizer steps of every model.  allreduce mode: plain full steps.  rs_ag mode: each rank steps its shards,
        then the shards are all-gathered into every rank's parameter buffer and the packed conv operands refreshed."""
        if self.mode != "rs_ag" or not (self.world > 1 or self.force):
            for opt in self.opts:
                opt.step()
            self._mark_started()
            return
        works = []
        for opt in self.opts:
            fp = opt.fp
            opt.step(within=self.owned_ranges(fp), finish=False)
            for f, a, b, _ in self.buckets:
                if f is not fp:
                    continue
                n = (b - a) // self.world
                mine = fp.flat_padded[a + self.rank * n:a + (self.rank + 1) * n].clone()
                works.append(dist.all_gather_into_tensor(fp.flat_padded[a:b], mine, group=self.group, async_op=True))
                self.comm_stats["collectives"] += 1
                self.comm_stats["bytes"] += 4 * (b - a)
        for w in works:
            w.wait()
        for opt in self.opts:
            opt.finish_step()
        self._mark_started()

    def _mark_started(self):
        """Device-side map: the parameters stepped this time have a momentum buffer from now on (one launch, all models)."""
        if self._dev_used and self.opts[0].device_flags is not None:
            from . import functional as GF

            GF.flags_max_(self._started_dev, self._flags_dev)

    def agreed_used(self):
        """{optimizer position: [bool per parameter]} of the last step as the ranks agreed on it (reads the device flags:
        a synchronising call, for tests and logging)."""
        if not self._dev_used:
            return {k: list(o.fp.used) for k, o in enumerate(self.opts)}
        bits = self._flags_dev.tolist()
        return {k: [v > 0 for v in bits[lo:hi]] for k, (lo, hi) in enumerate(self._flag_slices)}

    def reset(self):
        """Call after zero_grad, before the next backward."""
        self._pending = [len(b[3]) for b in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._ready = [False] * len(self.buckets)
        self._next = 0
        self._works = []
        self.comm_stats = {"collectives": 0, "bytes": 0}

    def finish(self):
        """Call after backward: flush buckets that hold unused parameters and wait for every all-reduce."""
        if self.world > 1 or self.force:
            self._drain(everything=True)
            # the "received a gradient" maps: on the device behind the last bucket (CUDA models), else over the host group
            # WHILE the device works through the buckets (posted here, waited for behind the stream-level bucket waits)
            pending = None
            if self._dev_used:
                slot = self.used_syncs & 1
                host = self._flags_host[slot]
                # the upload of two steps ago read this pinned buffer asynchronously: workloads without a per-step host
                # synchronisation (fpn) could get here before it has run
                if self._flags_events[slot] is not None:
                    self._flags_events[slot].synchronize()
                host.copy_(torch.tensor([1.0 if u else 0.0 for o in self.opts for u in o.fp.used]))
                self._flags_dev.copy_(host, non_blocking=True)
                if self._flags_dev.is_cuda:
                    ev = self._flags_events[slot] = self._flags_events[slot] or torch.cuda.Event()
                    ev.record()
                self._works.append(dist.all_reduce(self._flags_dev, op=dist.ReduceOp.MAX, group=self.group, async_op=True))
                for o, (lo, hi) in zip(self.opts, self._flag_slices):
                    o.device_flags = (self._flags_dev[lo:hi], self._started_dev[lo:hi])
                self.used_syncs += 1
            else:
                pending = self._post_used()
            for w in self._works:
                w.wait()
            if self.mode == "rs_ag":      # the reduced shards land in the gradient buffer's owned ranges
                for bid, (fp, a, b, _) in enumerate(self.buckets):
                    n = (b - a) // self.world
                    fp.grad_padded[a + self.rank * n:a + (self.rank + 1) * n].copy_(self._shard_buf[bid])
            # every rank must step the same parameters: a parameter used on any rank is used everywhere
            if not self._dev_used:
                self._sync_used(pending)
        elif self._dev_used:
            for o in self.opts:
                o.device_flags = None
        self._works = []

    def _post_used(self):
        """Post the OR of every rank's per-parameter "received a gradient" bits (all optimizers in one message)."""
        if self._host_group is None and not self._device_agree:
            return None
        t = torch.tensor([u for opt in self.opts for u in opt.fp.used], dtype=torch.uint8)
        if self._host_group is not None:
            return t, dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self._host_group, async_op=True)
        t = t.to(self.opts[0].fp.flat.device, dtype=torch.int32)
        return t, dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group, async_op=True)

    def _sync_used(self, pending=None):
        """Wait for the posted exchange and adopt the agreed map, every step."""
        pending = pending if pending is not None else self._post_used()
        if pending is None:
            return
        t, work = pending
        work.wait()
        bits, lo = t.tolist(), 0
        for opt in self.opts:
            n = len(opt.fp.used)
            opt.fp.used = [bool(v) for v in bits[lo:lo + n]]
            lo += n
        self.used_syncs += 1


def broadcast_parameters(flat_params_list, src=0, group=None):
    """Make all replicas start from rank `src`'s weights (what DDP does at construction)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    for fp in flat_params_list:
        dist.broadcast(fp.flat, src=src, group=group)
        fp.version += 1   # packed conv operands made from the old values are stale
