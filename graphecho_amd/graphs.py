"""HIP-graph replay of the static-shape segments of a training step.

A step of the reference's loop (train_camus_echo.py:183-303) spends most of its launches in networks whose shapes never
change from one iteration to the next -- the FPN passes (fpnseg.py:405-444) and the four Discriminators -- and at the
per-GPU batch sizes of data-parallel training (BASELINE config 4: 64 frames over 8 GPUs) those passes are bound by the
host issuing ~1500 launches through Python, not by the kernels.  ``GraphedModule`` captures such a module's forward and
its backward once each into a ``hipGraph`` (torch.cuda.CUDAGraph drives hipStreamBeginCapture/hipGraphLaunch) and
replays them afterwards: one host call per pass and direction.

What is captured is exactly the eager path: the same library entry points on the capture stream, the BatchNorm running
statistics updated in place by the same kernels, SyncBN's exchanges (RCCL collectives are capturable) and the weight
gradients accumulated straight into the flat gradient buffers (functional.DIRECT_GRAD_ACCUM).  The data-dependent parts
of the step (GModule's node sampling, spectral clustering, matching) stay eager.

Mechanics:
  * the first ``warmup`` calls of a (tag, input shapes, BatchNorm segmentation, conv precision) combination run eagerly
    (one-time lazy setup inside the library, packed-operand caches); the next call captures the forward, the backward is
    captured lazily inside the first backward pass that reaches it, with the set of outputs that actually receive a
    gradient;
  * forward and backward share one private memory pool: saved activations stay where the forward graph wrote them until
    the backward graph has consumed them, so a slot is replayed strictly forward -> backward -> forward ... -- call sites
    that run the module twice before one backward (source pass + target pass) use different ``tag``s;
  * parameters that received gradient inside the captured backward are reported to their FlatParams when the replay has
    been enqueued, so the optimizers' "used" maps and the gradient synchroniser's buckets see what they see in eager
    mode (all of the segment's parameters at once rather than layer by layer);
  * host-side counters the eager code keeps (BatchNorm.num_batches_tracked increments, SyncBN exchange counts) advance
    per replay by what one eager pass advances them.
"""
import os
import weakref

import torch
from torch.autograd import Function

from . import functional as GF
from . import nn as gnn


# Default of GE_GRAPH_FORK: "1" captures the weight-gradient kernels of a backward graph as a side branch (fork / join
# events), "0" as links of one chain.  A forked graph is executed over several internal streams of the runtime, and
# kernels of OTHER streams submitted after its launch then wait for the whole graph (measured: GModule's stream started
# only when the head / discriminator backward graph had finished, 23.2 vs 18.5 ms per 8-frame step) -- the trainer sets
# "0" when it runs GModule on a stream of its own.
FORK_DEFAULT = "1"

# untyped-storage addresses of every LIVE captured graph's static outputs: memory owned by this module (graph pools are never
# returned to the allocator while their graph lives), the only inputs a later capture may read in place.  A slot removes its
# addresses when it dies (weakref.finalize): once its pool is released the allocator may hand the same address to a caller's
# tensor, which a later capture must NOT treat as framework-owned (it would alias it and write later batches into it).
_POOL_STORAGES = set()


def _flatten(obj, out):
    """Nested tuples / lists of tensors -> structure descriptor; tensors appended to `out`."""
    if torch.is_tensor(obj):
        out.append(obj)
        return None
    if isinstance(obj, (tuple, list)):
        return (type(obj), [_flatten(o, out) for o in obj])
    raise TypeError(f"GraphedModule: unsupported value of type {type(obj).__name__} (tensors in tuples / lists only)")


def _rebuild(spec, it):
    if spec is None:
        return next(it)
    kind, subs = spec
    return kind(_rebuild(s, it) for s in subs)


def _reachable_leaves(roots):
    """ids of the leaf tensors (AccumulateGrad variables) reachable from `roots` in the autograd graph."""
    seen, leaves = set(), set()
    stack = [r.grad_fn for r in roots if r.grad_fn is not None]
    while stack:
        fn = stack.pop()
        if fn in seen:
            continue
        seen.add(fn)
        var = getattr(fn, "variable", None)
        if var is not None:
            leaves.add(id(var))
            continue
        stack.extend(nf for nf, _ in fn.next_functions if nf is not None)
    return leaves


class _Replay(Function):
    """Autograd node standing for one replayed forward: its backward replays the captured backward graph."""

    @staticmethod
    def forward(ctx, slot, _anchor, *inputs):
        ctx.slot = slot
        ctx.generation = slot.generation
        outs = tuple(o.detach() for o in slot.static_outs)
        nd = [o for o, s in zip(outs, slot.static_outs) if not s.requires_grad]
        if nd:
            ctx.mark_non_differentiable(*nd)
        ctx.set_materialize_grads(False)
        return outs

    @staticmethod
    def backward(ctx, *gouts):
        if ctx.generation != ctx.slot.generation:
            raise RuntimeError("GraphedModule: backward of a forward pass whose saved activations a later replay of the "
                               "same call site has overwritten; give call sites that run before one backward their own "
                               "tag")
        gin = ctx.slot.backward(gouts)
        return (None, None) + tuple(gin)


class _Slot:
    def __init__(self, owner, grad=True):
        self.owner = owner
        self.grad = grad            # False: forward-only slot (called under torch.no_grad())
        self.calls = 0
        self.fwd_graph = self.bwd_graph = None
        self.pool = None
        self.static_in = self.static_outs = self.static_gouts = self.static_gin = None
        self.in_spec = self.out_spec = None
        self.mask = None
        self.proxies = {}           # id(stand-in leaf) -> (stand-in, FlatParams, parameter index)
        self.used = []              # [(FlatParams, [parameter indices])] touched by the captured backward
        self.bn_counts = []         # [(BatchNorm2d, num_batches_tracked increments per forward)]
        self.sync_fwd = self.sync_bwd = (0, 0, 0)
        self.generation = 0         # forward replays so far: a backward must belong to the latest one
        self._owned = []            # storage addresses this slot put into _POOL_STORAGES
        weakref.finalize(self, _POOL_STORAGES.difference_update, self._owned)

    # ---- forward ---------------------------------------------------------------------------------------------
    def _capture_forward(self, inputs):
        own = self.owner
        # Inputs that live in memory THIS FRAMEWORK owns -- the static outputs of another captured graph (the pyramid
        # maps a head / discriminator graph consumes, or views of them) -- are read where they are: they sit at the same
        # address every step and nobody else can hold them.  Anything else (the caller's frames, a preloaded batch list)
        # is copied into a private static buffer before every replay: writing a later batch INTO the tensor the caller
        # passed at capture time would silently destroy the caller's data.
        self.static_in, self.aliased = [], []
        for x in inputs:
            own_mem = x.untyped_storage().data_ptr() in _POOL_STORAGES
            s = x.detach() if own_mem else torch.empty_strided(x.shape, x.stride(), dtype=x.dtype, device=x.device)
            if not own_mem:
                s.copy_(x.detach())
            self.static_in.append(s.requires_grad_(x.requires_grad and self.grad))
            self.aliased.append(own_mem)
        bns = [m for m in own.module.modules() if isinstance(m, gnn.BatchNorm2d)]
        before = [m._pending_batches for m in bns]
        sync0 = list(GF.SYNC_BN_STATS)
        self.pool = torch.cuda.graph_pool_handle()
        g = torch.cuda.CUDAGraph()
        args = _rebuild(self.in_spec, iter(self.static_in))
        # The module runs on stand-ins of its parameters: fresh leaves on the same storage (and the same flat gradient
        # views).  A parameter's own AccumulateGrad node lives on the stream FlatParams was built on and is kept alive
        # by its hooks; autograd would synchronise that stream with the capture stream when a captured gradient reaches
        # it, which is illegal inside a capture.  The stand-ins' nodes are created here, on the capture stream.
        self.proxies, by_name = {}, {}
        table = {id(p): (fp, i) for fp in own.fps for i, p in enumerate(fp.params)}
        if self.grad:
            for name, p in own.module.named_parameters():
                if id(p) in table:
                    q = p.detach().requires_grad_(True)
                    q._ge_flat, q.grad = p._ge_flat, p.grad
                    self.proxies[id(q)] = (q,) + table[id(p)]
                    by_name[name] = q
        with torch.cuda.graph(g, pool=self.pool, stream=own.capture_stream, capture_error_mode="thread_local"):
            if self.grad:
                with torch.enable_grad():
                    result = torch.func.functional_call(own.module, by_name, tuple(args), strict=False)
            else:      # forward-only replay (pseudo-label passes): no tape, no stand-ins
                with torch.no_grad():
                    result = own.module(*args)
        outs = []
        self.out_spec = _flatten(result, outs)
        self.static_outs = outs
        for o in outs:
            addr = o.untyped_storage().data_ptr()
            _POOL_STORAGES.add(addr)
            self._owned.append(addr)
        self.fwd_graph = g
        # what one eager pass adds to the host-side counters (the capture itself executed nothing)
        self.bn_counts = [(m, m._pending_batches - b) for m, b in zip(bns, before) if m._pending_batches != b]
        for m, b in zip(bns, before):
            m._pending_batches = b
        self.sync_fwd = tuple(a - b for a, b in zip(GF.SYNC_BN_STATS, sync0))
        GF.SYNC_BN_STATS[:] = sync0

    def run(self, inputs):
        if self.fwd_graph is None:
            self._capture_forward(inputs)
        else:
            with torch.no_grad():
                for s, x, own_mem in zip(self.static_in, inputs, self.aliased):
                    # an aliased input that arrives at the captured address with the captured layout (strides are part
                    # of the slot key) needs nothing; everything else is copied to where the graph reads it -- for an
                    # aliased slot that is pool memory of another graph, rewritten by that graph's next replay anyway
                    if not (own_mem and s.data_ptr() == x.data_ptr()):
                        s.copy_(x.detach())
        self.fwd_graph.replay()
        self.generation += 1
        for m, n in self.bn_counts:
            m._pending_batches += n
        for k in range(3):
            GF.SYNC_BN_STATS[k] += self.sync_fwd[k]
        if not self.grad:
            return _rebuild(self.out_spec, iter([o.detach() for o in self.static_outs]))
        outs = _Replay.apply(self, self.owner._anchor, *inputs)
        return _rebuild(self.out_spec, iter(outs))

    # ---- backward --------------------------------------------------------------------------------------------
    def _capture_backward(self, gouts):
        own = self.owner
        self.mask = tuple(g is not None for g in gouts)
        self.static_gouts = [torch.empty_like(o) if m else None for o, m in zip(self.static_outs, self.mask)]
        roots = [o for o, m in zip(self.static_outs, self.mask) if m]
        # Leaves the captured backward reaches.  It is run with torch.autograd.grad on exactly these, so no
        # AccumulateGrad node executes inside the capture (those nodes were created on another stream and carry the
        # FlatParams hooks); gradients the kernels do not accumulate in place are added to the flat buffers by hand.
        reach = _reachable_leaves(roots)
        wanted = [(fp, i, q) for q, fp, i in self.proxies.values() if id(q) in reach]
        known = {id(q) for _fp, _i, q in wanted} | {id(s) for s in self.static_in}
        if reach - known:
            raise RuntimeError("GraphedModule: the module has trainable parameters outside the FlatParams it was given")
        leaf_in = [s for s in self.static_in if s.requires_grad]
        sync0 = list(GF.SYNC_BN_STATS)
        saved = (GF.DIRECT_GRAD_ACCUM, GF.WGRAD_STREAM)
        # Weight-gradient kernels only feed the optimizer.  Captured on a second stream (forked from the capture stream
        # by an event per layer, joined once at the end) they become a side branch of the graph instead of links of its
        # one chain of nodes: at small batches, where a single kernel cannot fill the chip, the data-gradient chain and
        # the weight-gradient kernels then run side by side.  GE_GRAPH_FORK=0 captures one chain.
        fork = own.fork_stream if own.fork_enabled() else None
        saved_defer = GF.DEFER_SLABS
        GF.flush_slab_reduces()      # nothing queued by eager layers may end up inside the capture
        GF.DIRECT_GRAD_ACCUM, GF.WGRAD_STREAM = True, fork
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g, pool=self.pool, stream=own.capture_stream, capture_error_mode="thread_local"):
                grads = torch.autograd.grad(roots, [p for _fp, _i, p in wanted] + leaf_in,
                                            [s for s in self.static_gouts if s is not None], allow_unused=True)
                GF.flush_slab_reduces()      # batched slab reduces of this segment: inside the graph, on their branch
                if fork is not None:
                    torch.cuda.current_stream().wait_stream(fork)        # join: the capture ends on one stream
                with torch.no_grad():
                    for (_fp, _i, p), gp in zip(wanted, grads):
                        if gp is not None:
                            p.grad.add_(gp)
        finally:
            GF.DIRECT_GRAD_ACCUM, GF.WGRAD_STREAM = saved
            GF.DEFER_SLABS = saved_defer
        by_fp = {}
        for fp, i, _p in wanted:
            by_fp.setdefault(id(fp), (fp, []))[1].append(i)
        self.used = list(by_fp.values())
        gin = iter(grads[len(wanted):])
        self.static_gin = [next(gin) if s.requires_grad else None for s in self.static_in]
        self.sync_bwd = tuple(a - b for a, b in zip(GF.SYNC_BN_STATS, sync0))
        GF.SYNC_BN_STATS[:] = sync0
        self.bwd_graph = g

    def backward(self, gouts):
        if self.bwd_graph is None:
            self._capture_backward(gouts)
        mask = tuple(g is not None for g in gouts)
        if any(m and not c for m, c in zip(mask, self.mask)):
            raise RuntimeError("GraphedModule: an output that had no gradient when the backward was captured has one "
                               "now; give this call site its own tag")
        for s, g_ in zip(self.static_gouts, gouts):
            if s is None:
                continue
            if g_ is None:
                s.zero_()
            else:
                s.copy_(g_)
        self.bwd_graph.replay()
        for k in range(3):
            GF.SYNC_BN_STATS[k] += self.sync_bwd[k]
        for fp, idx in self.used:
            for i in idx:
                fp.notify(i)
        return [None if g_ is None else g_.detach() for g_ in self.static_gin]


class GraphedModule:
    """``GraphedModule(module, flat_params)(*inputs, tag=...)`` == ``module(*inputs)`` in train mode, replayed from
    HIP graphs (under ``torch.no_grad()`` a forward-only graph); anything else (eval mode, live kernel timing,
    ``enabled = False``) goes to the module directly."""

    def __init__(self, module, flat_params=(), warmup=2):
        self.module = module
        self.fps = list(flat_params)
        self.warmup = warmup
        self.enabled = True
        self.slots = {}
        self._anchor = None
        self.fork_stream = None         # side branch of the backward graphs (weight-gradient kernels)
        # None: GE_GRAPH_FORK / FORK_DEFAULT decide; True / False: this module's backward graphs are (not) forked whatever the
        # default -- the trainer forks the pyramid's backward (nothing else of the step is in flight while it runs) and keeps the
        # head / discriminator graphs one chain (they run beside GModule's stream, which a forked graph would hold up)
        self.fork = None
        self.capture_stream = None      # fwd and bwd captures of every slot use one stream: autograd runs a node's
        #                                 backward on the stream its forward ran on

    def _fingerprint(self):
        """Module state that changes WHAT a capture records without changing any input shape: SyncBN on / off and its
        group, world size, BatchNorm momentum / running-statistics mode, the fork switch of the backward capture.  Part
        of the slot key, so flipping any of them (convert_sync_batchnorm, force_sync in tests) captures anew instead of
        silently replaying the old graph."""
        import torch.distributed as dist

        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        h = [world, self.fork_enabled()]
        for m in self.module.modules():
            if isinstance(m, gnn.BatchNorm2d):
                h.append((m.sync, m.force_sync, m.momentum, m.track_running_stats, id(m.process_group)))
        return hash(tuple(h))

    def fork_enabled(self):
        return (os.environ.get("GE_GRAPH_FORK", FORK_DEFAULT) != "0") if self.fork is None else bool(self.fork)

    def _eligible(self, flat):
        return (self.enabled and self.module.training and GF.KERNEL_TIMER is None and all(t.is_cuda for t in flat))

    def __call__(self, *inputs, tag=0):
        flat = []
        spec = _flatten(tuple(inputs), flat)
        if not self._eligible(flat):
            return self.module(*inputs)
        grad = torch.is_grad_enabled()
        key = (tag, grad, tuple((tuple(t.shape), tuple(t.stride()), t.dtype, t.requires_grad and grad) for t in flat),
               GF.BN_SEGMENTS,
               GF.CONV_PRECISION, GF.ACT_STORAGE, repr(spec), self._fingerprint())
        slot = self.slots.get(key)
        if slot is None:
            slot = self.slots[key] = _Slot(self, grad)
        slot.calls += 1
        if slot.calls <= self.warmup:
            return self.module(*inputs)
        if self._anchor is None:
            # gives the replay node a differentiable input when no tensor input needs a gradient (the FPN's images)
            self._anchor = torch.zeros(1, device=flat[0].device, requires_grad=True)
            self.capture_stream = torch.cuda.Stream(device=flat[0].device)
            self.fork_stream = torch.cuda.Stream(device=flat[0].device)
        slot.in_spec = spec
        return slot.run(flat)

    def graphs(self):
        """Number of captured (forward, backward) graphs (tests / bench report)."""
        return (sum(1 for s in self.slots.values() if s.fwd_graph is not None),
                sum(1 for s in self.slots.values() if s.bwd_graph is not None))
