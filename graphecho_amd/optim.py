"""Flat-buffer optimizers: every parameter of a model lives in one contiguous fp32 HBM buffer (and its
gradient in another), so the optimizer step is ONE fused kernel and the data-parallel gradient exchange is a
handful of large contiguous RCCL all-reduces instead of hundreds of small ones.

Semantics are those of torch.optim.Adam / torch.optim.SGD as the reference configures them
(train_camus_echo.py:425-435: Adam(lr, weight_decay) for the FPN, SGD(lr, momentum 0.9, weight_decay) for the
rest), including "a parameter that received no gradient this step is left untouched" (torch skips ``grad is
None``) -- tracked with post-accumulate hooks, no host sync.
"""
import torch

from . import functional as GF


class FlatParams:
    """Re-homes the parameters of `modules` into one flat buffer; ``p.grad`` become views of a flat grad buffer."""

    PAD = 64

    def __init__(self, modules):
        if isinstance(modules, torch.nn.Module):
            modules = [modules]
        seen, params = set(), []
        for m in modules:
            for p in m.parameters():
                if p.requires_grad and id(p) not in seen:
                    seen.add(id(p))
                    params.append(p)
        if not params:
            raise ValueError("FlatParams: no trainable parameters")
        self.params = params
        dev, dt = params[0].device, params[0].dtype
        self.offsets, total = [], 0
        for p in params:
            self.offsets.append(total)
            total += p.numel()
        self.numel = total
        # PAD zero elements behind the last parameter: the sharded gradient exchange (ddp.GradSynchronizer mode
        # "rs_ag") cuts the buffers into pieces whose sizes are multiples of the world size
        self.flat_padded = torch.zeros(total + self.PAD, device=dev, dtype=dt)
        self.grad_padded = torch.zeros(total + self.PAD, device=dev, dtype=dt)
        self.flat = self.flat_padded[:total]
        self.grad = self.grad_padded[:total]
        self.used = [False] * len(params)
        self.version = 0   # bumped whenever the flat parameter buffer is rewritten (optimizer step, broadcast, load)
        self._hooks, self._nodes = [], []
        for i, (p, o) in enumerate(zip(params, self.offsets)):
            n = p.numel()
            self.flat[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.flat[o:o + n].view(p.shape)
            p.grad = self.grad[o:o + n].view(p.shape)
            p._ge_flat = (self, i)   # lets the conv wgrad kernel accumulate straight into the flat buffer
            # "Parameter i has all of its gradient" = its AccumulateGrad node has run.  The engine runs that node once
            # per backward, after EVERY reachable use of the parameter (a module applied to the source and the target
            # batch, weight sharing across pyramid levels) has executed its backward -- also when those backwards
            # accumulated straight into the flat buffer and handed autograd no gradient at all (a tensor-level
            # post-accumulate hook would never fire then).  The node is kept alive here so the hook stays attached.
            node = p.view_as(p).grad_fn.next_functions[0][0]
            self._nodes.append(node)
            self._hooks.append(node.register_hook(self._make_hook(i)))
        self.listeners = []  # called as fn(index) when parameter `index` has its gradient accumulated
        self._seg_end = None

    def seg_end_dev(self):
        """int32 device tensor of the parameters' exclusive end offsets in the flat buffer (masked optimizer kernels)."""
        if self._seg_end is None:
            self._seg_end = torch.tensor([o + p.numel() for p, o in zip(self.params, self.offsets)], dtype=torch.int32,
                                         device=self.flat.device)
        return self._seg_end

    def notify(self, i):
        """Parameter i received (all of) its gradient for this backward pass."""
        if not self.used[i]:
            self.used[i] = True
            for fn in self.listeners:
                fn(i)

    def _make_hook(self, i):
        def hook(_grad_inputs, _grad_outputs):
            self.notify(i)
        return hook

    def zero_grad(self):
        self.grad.zero_()
        self.used = [False] * len(self.params)
        for p, o in zip(self.params, self.offsets):   # user code may have set p.grad = None
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + o * 4:
                p.grad = self.grad[o:o + p.numel()].view(p.shape)

    def used_ranges(self, within=None):
        """Contiguous [start, end) element ranges covering the parameters that received a gradient; `within`: list of
        [lo, hi) ranges (this rank's shards under the sharded exchange) the result is clipped to."""
        if within is not None:
            return clip_ranges(self.used_ranges(), within)
        ranges, start, end = [], None, None
        for p, o, u in zip(self.params, self.offsets, self.used):
            if u:
                if start is None:
                    start = o
                end = o + p.numel()
            elif start is not None:
                ranges.append((start, end))
                start = None
        if start is not None:
            ranges.append((start, end))
        return ranges


def clip_ranges(ranges, within):
    """Intersection of two lists of [lo, hi) ranges (each sorted, non-overlapping); extra tuple fields of `ranges`
    are carried over."""
    out = []
    for r in ranges:
        for lo, hi in within:
            a, b = max(r[0], lo), min(r[1], hi)
            if a < b:
                out.append((a, b) + tuple(r[2:]))
    return out


class WeightPacker:
    """Re-packs every conv weight of a FlatParams model into its K-major operand layouts (forward, and
    data-gradient where it differs from OIHW) with ONE launch right after the optimizer step, instead of one lazy
    `ge_conv2d_pack_weight` launch per layer and layout on first use (~140 launches per step on FPN-ResNet).
    The per-layer PackCache keeps validating (pointer, version, parameter epoch), so any other weight change
    simply falls back to the lazy path."""

    def __init__(self, fp, modules):
        from . import nn as gnn
        from ._lib import check, lib

        self._check, self._lib = check, lib
        if isinstance(modules, torch.nn.Module):
            modules = [modules]
        index = {id(p): i for i, p in enumerate(fp.params)}
        rows, self.slots, total = [], [], 0
        for top in modules:
            for m in top.modules():
                if not isinstance(m, gnn.Conv2d) or id(m.weight) not in index:
                    continue
                src = fp.offsets[index[id(m.weight)]]
                cout, cin_g, kh, kw = m.weight.shape
                n = m.weight.numel()
                for tr in (0, 1):
                    if tr and m.groups == 1 and kh == 1 and kw == 1:
                        continue      # PackCache hands out the OIHW weight itself
                    rows.append([src, total, n, m.groups, cout // m.groups, cin_g, kh * kw, tr])
                    self.slots.append((m, tr, total, n))
                    total += n
        self.fp = fp
        # Winograd operands (ge_wino.hip) of the 3x3 layers that run on them: registered at a layer's first use
        # (PackCache.get_wino -> add_wino), one persistent buffer per (layer, direction), all refreshed by one more batched launch
        self._by_cache = {}
        for top in modules:
            for m in top.modules():
                if isinstance(m, gnn.Conv2d) and id(m.weight) in index:
                    m._pack.owner = self
                    self._by_cache[id(m._pack)] = (m, fp.offsets[index[id(m.weight)]])
        self.wino_rows, self.wino_slots, self.wino_table = [], [], None
        self.n = len(rows)
        if self.n:
            self.table = torch.tensor(rows, dtype=torch.int64, device=fp.flat.device)
            self.packed = torch.empty(total, device=fp.flat.device, dtype=fp.flat.dtype)
            self.views = [self.packed[o:o + n] for (_m, _tr, o, n) in self.slots]

    @torch.no_grad()
    def add_wino(self, cache, weight, transposed):
        """First Winograd use of a layer (outside any capture, weights at the version the static views hold): allocate its
        persistent operand buffer, pack it now, and enrol it in the per-step batched launch."""
        m, src = self._by_cache[id(cache)]
        cout, cin = weight.shape[0], weight.shape[1]
        M, C = (cin, cout) if transposed else (cout, cin)
        buf = torch.empty(16 * cout * cin, device=weight.device, dtype=weight.dtype)
        self._check(self._lib.ge_wino3x3_pack_weight(weight.data_ptr(), buf.data_ptr(), M, C, int(transposed),
                                                     torch.cuda.current_stream().cuda_stream), "wino3x3_pack_weight")
        self.wino_rows.append([src, buf.data_ptr(), M, C, int(transposed)])
        self.wino_slots.append((m, bool(transposed), buf))
        self.wino_table = None
        cache.static[("wino", bool(transposed))] = buf
        return buf

    @torch.no_grad()
    def repack(self):
        if not self.fp.flat.is_cuda:
            return
        if self.wino_rows:
            if self.wino_table is None:      # (re)built when a layer joined: the first steps of a run only
                self.wino_table = torch.tensor(self.wino_rows, dtype=torch.int64).pin_memory().to(self.fp.flat.device,
                                                                                                 non_blocking=True)
            self._check(self._lib.ge_wino3x3_pack_weights_batched(self.fp.flat.data_ptr(), self.wino_table.data_ptr(),
                                                                  len(self.wino_rows),
                                                                  torch.cuda.current_stream().cuda_stream),
                        "wino3x3_pack_weights_batched")
        if not self.n:
            return
        self._check(self._lib.ge_conv2d_pack_weights_batched(self.fp.flat.data_ptr(), self.packed.data_ptr(),
                                                             self.table.data_ptr(), self.n,
                                                             torch.cuda.current_stream().cuda_stream),
                    "pack_weights_batched")
        ver = self.fp.version
        for (m, tr, _o, _n), view in zip(self.slots, self.views):
            cache = m._pack
            cache.static[tr] = view
            cache.static_key = (m.weight.data_ptr(), m.weight._version, ver)


class _FlatOptimizer(torch.optim.Optimizer):
    """torch.optim.Optimizer subclass (so LR schedulers accept it) whose step is a fused flat-buffer kernel."""

    def __init__(self, modules, lr):
        self.fp = modules if isinstance(modules, FlatParams) else FlatParams(modules)
        self.grad_scale = 1.0   # set to 1/world_size by the gradient synchroniser (sum all-reduce -> mean)
        # data parallelism: (used, started) device flag views over this model's parameters, set by the gradient
        # synchroniser for the step at hand -- the "which parameters are stepped" decision is then read by the kernels
        # (ge_*_step_masked) instead of being planned on the host from fp.used
        self.device_flags = None
        self.packer = WeightPacker(self.fp, modules) if not isinstance(modules, FlatParams) else None
        super().__init__(self.fp.params, dict(lr=lr))
        if self.packer is not None:
            self.packer.repack()

    def zero_grad(self, set_to_none=False):
        self.fp.zero_grad()

    def _lr(self):
        return self.param_groups[0]["lr"]


class FlatAdam(_FlatOptimizer):
    def __init__(self, modules, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(modules, lr)
        self.betas, self.eps, self.weight_decay = betas, eps, weight_decay
        self.m = torch.zeros_like(self.fp.flat)
        self.v = torch.zeros_like(self.fp.flat)
        self.step_count = 0

    @torch.no_grad()
    def step(self, within=None, finish=True):
        """within: [lo, hi) element ranges this rank owns (sharded exchange: the owner updates, then the shards are
        all-gathered); finish=False leaves the version bump + conv-operand repack to the caller (after the gather)."""
        self.step_count += 1
        fp = self.fp
        if self.device_flags is not None:
            used, _started = self.device_flags
            for a, b in (within if within is not None else [(0, fp.numel)]):
                GF.adam_step_masked_(fp.flat[a:b], fp.grad[a:b], self.m[a:b], self.v[a:b], a, fp.seg_end_dev(), used,
                                     self._lr(), self.betas[0], self.betas[1], self.eps, self.weight_decay,
                                     self.step_count, self.grad_scale)
            if finish:
                self.finish_step()
            return
        for a, b in fp.used_ranges(within):
            GF.adam_step_(fp.flat[a:b], fp.grad[a:b], self.m[a:b], self.v[a:b], self._lr(), self.betas[0],
                          self.betas[1], self.eps, self.weight_decay, self.step_count, self.grad_scale)
        if finish:
            self.finish_step()

    def finish_step(self):
        self.fp.version += 1
        if self.packer is not None:
            self.packer.repack()


class FlatSGD(_FlatOptimizer):
    def __init__(self, modules, lr=1e-3, momentum=0.0, weight_decay=0.0):
        super().__init__(modules, lr)
        self.momentum, self.weight_decay = momentum, weight_decay
        self.buf = torch.zeros_like(self.fp.flat) if momentum != 0 else None
        self.started = torch.zeros(len(self.fp.params), dtype=torch.bool).tolist()

    @torch.no_grad()
    def step(self, within=None, finish=True):
        fp = self.fp
        if self.device_flags is not None:       # (the "first step of a parameter" rule is read from the device flags too)
            used, started = self.device_flags
            for a, b in (within if within is not None else [(0, fp.numel)]):
                GF.sgd_step_masked_(fp.flat[a:b], fp.grad[a:b], None if self.buf is None else self.buf[a:b], a,
                                    fp.seg_end_dev(), used, started, self._lr(), self.momentum, self.weight_decay,
                                    self.grad_scale)
            if finish:
                self.finish_step()
            return
        # momentum buffers start as "buf = grad" the first time a parameter is stepped (torch.optim.SGD)
        first = [u and not s for u, s in zip(fp.used, self.started)]
        ranges = []
        for is_first in (True, False):
            start = end = None
            for p, o, u, f in zip(fp.params, fp.offsets, fp.used, first):
                if u and f == is_first:
                    if start is None:
                        start = o
                    end = o + p.numel()
                elif start is not None:
                    ranges.append((start, end, is_first))
                    start = None
            if start is not None:
                ranges.append((start, end, is_first))
        if within is not None:
            ranges = clip_ranges(ranges, within)
        for a, b, is_first in ranges:
            GF.sgd_step_(fp.flat[a:b], fp.grad[a:b], None if self.buf is None else self.buf[a:b], self._lr(),
                         self.momentum, self.weight_decay, is_first, self.grad_scale)
        self.started = [s or u for s, u in zip(self.started, fp.used)]
        if finish:
            self.finish_step()

    def finish_step(self):
        self.fp.version += 1
        if self.packer is not None:
            self.packer.repack()
