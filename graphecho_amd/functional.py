"""autograd bindings of the HIP kernels (torch tensors in, ctypes calls on the current HIP stream).

Every function here runs on the hand-written gfx950 kernels of libgraphecho_hip.so; there is no ATen/CPU
fallback (a missing library or a non-CUDA tensor raises).  Shapes follow PyTorch's conventions for the ops
the reference uses (see include/graphecho_hip.h for the reference call sites).
"""

import os
import re

import torch
from torch.autograd import Function

from ._lib import lib, check

_f32 = torch.float32


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """hipStream_t of torch's current stream as an int (the raw getter is ~10x cheaper than building a Stream object;
    a training step asks for it ~2000 times)."""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


def _c(t):
    """Contiguous fp32 CUDA tensor or a loud failure."""
    if not t.is_cuda:
        raise RuntimeError("graphecho_amd ops need tensors on the HIP device (no CPU fallback)")
    if t.dtype != _f32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


# --------------------------------------------------------------------------------------------------
# optional live kernel timing (bench.py): HIP events recorded on the stream the kernels are launched on
# --------------------------------------------------------------------------------------------------
class KernelTimer:
    """Collects (kernel family, rocprof kernel name, algorithmic FLOPs, start/end events) per conv launch.

    One record per kernel launch: a weight-gradient call is two launches (the split-K kernel and its slab reduce), told
    apart by an event the C side records between them (ge_set_wgrad_split_event), so the split-K kernel's own duration
    is what its record holds and `slab_reduce_kernel` gets records of its own (FLOPs 0, bytes = slabs read + dW)."""

    def __init__(self):
        self.records = []

    def begin(self):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    def begin_wgrad(self):
        """-> (start event, split event); the split event is handed to the library, which re-records it on the launch
        stream right after the split-K kernel."""
        mid = torch.cuda.Event(enable_timing=True)
        mid.record()                      # creates the underlying hipEvent_t
        lib.ge_set_wgrad_split_event(mid.cuda_event)
        return self.begin(), mid

    def end(self, start, kind, flops, nbytes=0, split=None, slab_bytes=0, executed=None):
        """flops: ALGORITHMIC FLOPs of the launch (direct convolution: 2 M N K); executed: FLOPs the matrix pipe
        actually performs when that differs (Winograd F(2x2, 3x3): 16 / 36 of them) -- the roofline fraction of a kernel is
        executed FLOPs over its time over the pipe's peak, never above 1; nbytes: compulsory HBM bytes of the launch (each
        operand and the result once)."""
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        name = lib.ge_last_conv_kernel().decode()   # instantiation the C side just launched, as rocprofv3 names it
        ex = flops if executed is None else executed
        if split is not None:
            lib.ge_set_wgrad_split_event(None)
            self.records.append((kind, name, flops, start, split, nbytes, ex))
            self.records.append(("slab_reduce", "slab_reduce_kernel", 0.0, split, ev, slab_bytes, 0.0))
        else:
            self.records.append((kind, name, flops, start, ev, nbytes, ex))

    def summary(self, peak_tflops):
        fam, inst = {}, {}
        for kind, name, flops, s, e, nbytes, ex in self.records:
            dt = s.elapsed_time(e) * 1e-3
            for agg, key in ((fam, kind), (inst, name)):
                a = agg.setdefault(key, [0.0, 0.0, 0, 0.0, 0.0])
                a[0] += flops
                a[1] += dt
                a[2] += 1
                a[3] += nbytes
                a[4] += ex
        if not fam:
            return None
        total_t = sum(a[1] for a in fam.values())
        total_f = sum(a[0] for a in fam.values())
        total_x = sum(a[4] for a in fam.values())
        # The dominant kernel = the instantiation with the largest summed time, whatever it is (forward, data or weight
        # gradient).  Every record brackets exactly one launch, and this pass runs without the weight-gradient side
        # stream, so avg_launch_ms is the kernel's own duration (a rocprofv3 run of the normal two-stream step sees
        # kernels of both streams stretched by their co-runners; profile with GE_WGRAD_STREAM=0 to compare).
        name, (f, t, n, nb, fx) = max(((k, v) for k, v in inst.items() if v[0] > 0), key=lambda kv: kv[1][1])
        ach = fx / t / 1e12

        def rnd(v):
            d = {"tflops": round(v[0] / v[1] / 1e12, 2), "ms": round(1e3 * v[1], 3), "n": v[2],
                 "avg_launch_ms": round(1e3 * v[1] / v[2], 4)}
            if v[4] != v[0]:      # "tflops" = algorithmic (direct-convolution) FLOPs over time; what the matrix pipe executed:
                d["mfma_tflops"] = round(v[4] / v[1] / 1e12, 2)
            return d

        out = {"bound": "mfma", "kernel": name, "achieved": round(ach, 2), "peak": peak_tflops, "unit": "TFLOP/s",
               "frac": round(ach / peak_tflops, 4), "traffic": None, "launches": n,
               "avg_launch_ms": round(1e3 * t / n, 4), "algorithmic_gflop_per_launch": round(f / n / 1e9, 2),
               "algorithmic_bytes_per_launch": round(nb / n)}
        if fx != f:
            # a Winograd instantiation: `achieved` / `frac` count the FLOPs the matrix pipe executes (16 multiplications per 2 x 2
            # outputs); against the DIRECT algorithm's FLOPs (SURVEY 8d's figure) the same launches run at:
            out["executed_gflop_per_launch"] = round(fx / n / 1e9, 2)
            out["algorithmic_tflops"] = round(f / t / 1e12, 2)
            out["algorithmic_over_peak"] = round(f / t / 1e12 / peak_tflops, 4)
        return {**out,
                "all_conv_kernels": {"achieved": round(total_x / total_t / 1e12, 2),
                                     "frac": round(total_x / total_t / 1e12 / peak_tflops, 4),
                                     "algorithmic_tflops": round(total_f / total_t / 1e12, 2),
                                     "time_s": round(total_t, 4)},
                "per_kernel": {k: rnd(v) for k, v in sorted(fam.items())},
                "per_instance": {k: rnd(v) for k, v in sorted(inst.items(), key=lambda kv: -kv[1][1])}}


KERNEL_TIMER = None


TIMER_DETAIL = False


def _conv_kind(prefix, kh, stride, M, N, K=0):
    """Label of a timing record: kernel family, or family + GEMM extents (M x N x K) when TIMER_DETAIL is set."""
    if TIMER_DETAIL:
        return f"{prefix}_k{kh}s{stride}_M{M}_N{N}_K{K}"
    return f"{prefix}_k{kh}s{stride}"


# --------------------------------------------------------------------------------------------------
# conv2d
# --------------------------------------------------------------------------------------------------
_param_epoch = 0
# When True (set by the trainer around backward()), conv weight/bias gradients of parameters that live in a
# FlatParams buffer are accumulated by the wgrad kernel straight into that buffer and the autograd return is
# None: this removes one ATen add + one allocation per parameter per step.
DIRECT_GRAD_ACCUM = False
# When set (to a torch.cuda.Stream) together with DIRECT_GRAD_ACCUM, conv weight-gradient kernels are enqueued on that
# side stream: they only feed the optimizer, so they can run beside the data-gradient / normalisation kernels of the
# layers below instead of in front of them.  The owner (trainer) joins the stream before the optimizer step.
WGRAD_STREAM = None
# "f32" (exact fp32 MFMA) or "f16" (operands rounded to fp16, fp32 accumulation and storage -- ge_mfma_f16.hip: BASELINE.json
# config 5's conv path).  The fp16-operand kernels cover layers whose per-group channel counts are multiples of 32,
# the others stay on the fp32 kernels.  Read at forward time; the backward of a layer follows the precision its
# forward used.
CONV_PRECISION = "f32"
# "f32" or "f16": with "f16" the conv3x3 -> BatchNorm -> ReLU (-> max-pool) stacks of the VGG16 backbone keep their
# activations and activation gradients in HBM as channel-blocked fp16 (graphecho_amd/half.py, csrc/ge_half.hip);
# everything outside those stacks is untouched (and follows CONV_PRECISION).  Read at forward time.
ACT_STORAGE = os.environ.get("GE_ACT_STORAGE", "f32")
# fp32 3x3 / stride 1 / pad 1 convolutions (forward and data gradient) as Winograd F(2x2, 3x3) on the layers ge_wino.hip covers
# (csrc/ge_wino.hip: 16 multiplications per 2x2 outputs instead of 36; error against fp64 below the direct kernels').  0: direct
WINOGRAD = os.environ.get("GE_WINOGRAD", "1") != "0"
# Routing of a covered layer: None = the library's own plan (ge_wino3x3_splits: unsplit when the grid fills the chip, split over the
# input channels for the 32 x 32 / 16 x 16 / 8 x 8 maps of small per-GPU batches, 0 = the direct kernels keep the layer); an integer
# = every covered layer with at least that many workgroups (tests, bench.py's two-frame parity probe: they exercise on small inputs
# the kernels the timed batch-32 step runs)
WINOGRAD_MIN_BLOCKS = None
_WINO_PLAN = {}
# the same layers' WEIGHT gradient as Winograd F(3x3, 2x2) (csrc/ge_wino_wgrad.hip); GE_WINOGRAD_WGRAD=0: the direct kernel
WINOGRAD_WGRAD = os.environ.get("GE_WINOGRAD_WGRAD", "1") != "0"
_WINO_WGRAD = {}


def _wino_wgrad_ws(B, Cin, Cout, H, W):
    """Workspace floats of the Winograd weight gradient for the layer, 0 when it stays on the direct kernel."""
    if not (WINOGRAD and WINOGRAD_WGRAD):
        return 0
    key = (B, Cin, Cout, H, W)
    n = _WINO_WGRAD.get(key)
    if n is None:
        n = _WINO_WGRAD[key] = lib.ge_wino3x3_wgrad_workspace(B, Cin, Cout, H, W) if \
            lib.ge_wino3x3_wgrad_supported(B, Cin, Cout, H, W) else 0
    return n


def _wino_plan(B, C, M, H, W):
    """(K splits, workspace floats) of the Winograd route for a 3x3 / s1 / p1 pass with C reduction and M output channels;
    splits == 0: the pass stays on the direct kernels."""
    if not WINOGRAD:
        return 0, 0
    key = (B, C, M, H, W, WINOGRAD_MIN_BLOCKS)
    plan = _WINO_PLAN.get(key)
    if plan is None:
        if WINOGRAD_MIN_BLOCKS is None:
            splits = lib.ge_wino3x3_splits(B, C, M, H, W)
        else:
            ok = lib.ge_wino3x3_covered(B, C, M, H, W) and B * (H * W // 128) * (M // 64) >= WINOGRAD_MIN_BLOCKS
            splits = max(1, lib.ge_wino3x3_splits(B, C, M, H, W)) if ok else 0
        plan = _WINO_PLAN[key] = (splits, lib.ge_wino3x3_workspace(B, C, M, H, W) if splits > 1 else 0)
    return plan


# Loss scale of the gradients stored as fp16 (half.py; 3x3 convs below): multiplied in where a gradient is cast to fp16,
# divided out by the kernels that leave the fp16 domain (data gradient to fp32, weight / bias / affine gradients).
# H_DYNAMIC_SCALE (default): the scale lives in DEVICE memory (h_scale(): {scale, 1/scale, largest |gradient| cast since the last
# update}) and follows the gradients -- every cast records its tensor's largest magnitude, and once per step (h_scale_update():
# the trainer at the start of a step, anyone else's next forward after a backward) the scale becomes the power of two that puts
# that magnitude at 4096, 16x below fp16's maximum (stores saturate anyway).  torch.cuda.amp.GradScaler's job without a host
# read; H_GRAD_SCALE is the initial value.  Measured on config 5 (tools/h_grad_range.py, profiles/r04_half_grad_range.txt): the
# largest gradient element of a step is 4e-4 .. 8e-3; with a fixed 4096, 40 - 56 % of the non-zero elements of the big casts
# fall below fp16's smallest normal number, with the dynamic scale under 1 %.  GE_H_DYNAMIC_SCALE=0: fixed H_GRAD_SCALE.
H_GRAD_SCALE = float(os.environ.get("GE_H_GRAD_SCALE", "1024"))      # initial: the first backward of a run knows no magnitude yet (config 5's reaches 1e0)
H_DYNAMIC_SCALE = os.environ.get("GE_H_DYNAMIC_SCALE", "1") != "0"
H_SCALE_TARGET = 4096.0
H_SCALE_MIN, H_SCALE_MAX = float(2.0 ** -14), float(2 ** 24)     # gradients of 1e7 down to 1e-4 land on the target
_H_SCALE = {}       # device index -> the 4-float device tensor
_H_DIRTY = set()    # devices whose casts recorded magnitudes since the last update


def h_scale(device):
    """The device-resident loss scale of `device` (None with H_DYNAMIC_SCALE off)."""
    if not H_DYNAMIC_SCALE:
        return None
    t = _H_SCALE.get(device.index)
    if t is None:
        if torch.cuda.is_current_stream_capturing():
            # the tensor would land in that graph's private pool and its initialisation would become a graph node: every replay
            # would reset the scale.  GraphEchoTrainer.step / half.to_blocked create it before any capture.
            raise RuntimeError("functional.h_scale: the device-resident loss scale must exist before a HIP-graph capture "
                               "(call functional.h_scale(device) first)")
        t = torch.zeros(4, device=device, dtype=_f32)
        check(lib.ge_h_scale_init(_p(t), H_GRAD_SCALE, _stream()), "h_scale_init")
        _H_SCALE[device.index] = t
    return t


def h_scale_args(device, cast=False):
    """(host factor to multiply in, host factor to divide out, device pointer or None) for the kernels' scale arguments;
    cast=True marks the device: a gradient cast is about to record its magnitude."""
    t = h_scale(device)
    if t is None:
        return H_GRAD_SCALE, 1.0 / H_GRAD_SCALE, None
    if cast:
        _H_DIRTY.add(device.index)
    return 1.0, 1.0, t.data_ptr()


def h_scale_update(all_devices=False):
    """Step boundary: bring the scale of every device that saw gradient casts up to date (one tiny launch).  all_devices: also
    devices whose casts the host did not see -- a backward replayed from a HIP graph records magnitudes without running the
    Python that marks the device (the trainer passes True; the kernel leaves the scale alone when nothing was recorded)."""
    for idx in (list(_H_SCALE) if all_devices else list(_H_DIRTY)):
        check(lib.ge_h_scale_update(_p(_H_SCALE[idx]), H_SCALE_TARGET, H_SCALE_MIN, H_SCALE_MAX, _stream()), "h_scale_update")
    _H_DIRTY.clear()


# True while a trainer manages the step boundary (GraphEchoTrainer.step): forward passes then leave the scale alone.  An update
# triggered from a forward in the middle of a step -- the clip pyramid of the temporal workload runs after the first backward --
# could change the scale between a gradient cast and the weight-gradient kernels (side stream) that divide it out again.
H_SCALE_MANAGED = False


def h_scale_forward_update(device):
    """Forward paths entering the fp16 domain: make sure the scale exists (outside any capture) and, when nobody manages the step
    boundary, bring it up to date after a backward that recorded magnitudes."""
    if not H_DYNAMIC_SCALE:
        return
    if device.index not in _H_SCALE and not torch.cuda.is_current_stream_capturing():
        h_scale(device)
    if _H_DIRTY and not H_SCALE_MANAGED:
        h_scale_update()


def h_scale_value(device):
    """Current scale as a Python float (one host read: tests / diagnostics)."""
    t = h_scale(device)
    return H_GRAD_SCALE if t is None else float(t[0].item())


# With ACT_STORAGE == "f16", every OTHER 3x3 / stride 1 / pad 1 conv the blocked-fp16 kernels cover (FPN smoothing and head
# convs, discriminator towers, Bottleneck.conv2) runs on them too: its input (and, in backward, the incoming gradient) is
# cast to channel-blocked fp16 once, the kernels' epilogues write fp32 NCHW.  GE_H_GENERIC=0: only the VGG stacks.
H_GENERIC = os.environ.get("GE_H_GENERIC", "1") != "0"


def _lp_fns(mode):
    """Entry points of the fp16-operand conv path (mode "f16")."""
    if mode != "f16":
        raise RuntimeError(f"conv precision {mode!r}: only 'f32' and 'f16' exist (the parked bf16x3 family was removed in round 6)")
    return {k: getattr(lib, f"ge_conv2d_f16_{k}") for k in
            ("supported", "pack_weight", "fwd_stat_parts", "fwd", "dgrad", "wgrad_workspace", "wgrad")}


_LP = {}


def lp_fns(mode):
    fns = _LP.get(mode)
    if fns is None:
        fns = _LP[mode] = _lp_fns(mode)
    return fns


# Deferred slab reduces (set by the trainer around backward()): a direct-accumulating weight-gradient call leaves its K-split
# slabs in the workspace and up to 16 of them are reduced by ONE launch (ge_slab_reduce_batched) -- at small per-GPU
# batches the per-layer reduce launches (146 of ~1500 launches at 4 + 4 frames) cost more than the sums they compute.
# A batch is flushed when it is full, when its slabs exceed SLAB_CAP bytes (they should still sit in the Infinity Cache
# when they are read back), before a gradient bucket is exchanged and at the end of every backward call.
DEFER_SLABS = False
SLAB_CAP = 96 << 20
SLAB_DEFER_MAX = int(float(os.environ.get("GE_SLAB_DEFER_MAX_MB", "4")) * (1 << 20))    # only layers whose slabs are small: their reduce launch is pure latency; large ones reduce at once
_PENDING_SLABS = {}       # stream handle -> [entries (workspace, dw, n, splits)], bytes
_WGRAD_SPLITS = {}
_WGRAD_FUSES_BIAS = {}
# bias gradient inside the weight-gradient pass where the layer's kernel supports it (GE_WGRAD_BIAS=0: always ge_channel_sum)
WGRAD_BIAS = os.environ.get("GE_WGRAD_BIAS", "1") != "0"


def _push_slabs(stream, ws, dw, n, splits, stride=None):
    stride = n if stride is None else stride
    ent = _PENDING_SLABS.get(stream)
    if ent is not None and any(it[1].data_ptr() == dw.data_ptr() for it in ent[0]):
        _flush_stream(stream)     # a weight used twice (conv2 / semantic_branch across pyramid levels): its two reduces
        ent = None                # accumulate into the same gradient and must not share a launch
    if ent is None:
        ent = _PENDING_SLABS[stream] = [[], 0]
    ent[0].append((ws, dw, n, splits, stride))
    ent[1] += 4 * n * splits
    if len(ent[0]) >= 16 or ent[1] >= SLAB_CAP:
        _flush_stream(stream)


def _flush_stream(stream):
    import ctypes

    ent = _PENDING_SLABS.pop(stream, None)
    if not ent or not ent[0]:
        return
    items = ent[0]
    k = len(items)
    slabs = (ctypes.c_void_p * k)(*[it[0].data_ptr() for it in items])
    outs = (ctypes.c_void_p * k)(*[it[1].data_ptr() for it in items])
    ns = (ctypes.c_longlong * k)(*[it[2] for it in items])
    strides = (ctypes.c_longlong * k)(*[it[4] for it in items])
    sp = (ctypes.c_int * k)(*[it[3] for it in items])
    acc = (ctypes.c_int * k)(*([1] * k))
    check(lib.ge_slab_reduce_batched(slabs, outs, ns, strides, sp, acc, k, stream), "slab_reduce_batched")


def flush_slab_reduces():
    """Reduce every pending slab batch (each on the stream its weight-gradient kernels ran on)."""
    for stream in list(_PENDING_SLABS):
        _flush_stream(stream)


def bump_param_epoch():
    """Called by the fused optimizers: parameters changed behind autograd's version counters."""
    global _param_epoch
    _param_epoch += 1


def _weight_key(weight):
    """What a packed copy of `weight` is valid for: its storage, its autograd version and the version of the flat
    parameter buffer it lives in (bumped by that model's optimizer only -- another model's step does not invalidate
    it); weights outside a flat buffer fall back to the global parameter epoch."""
    flat = getattr(weight, "_ge_flat", None)
    return (weight.data_ptr(), weight._version, flat[0].version if flat is not None else _param_epoch)


class PackCache:
    """Per-layer cache of the K-major packed weights (forward and data-gradient layouts)."""

    __slots__ = ("entries", "static", "static_key", "owner")

    def __init__(self):
        self.entries = {}
        self.static = {}          # {transposed: view into a model-wide packed buffer} kept fresh by optim.WeightPacker
        self.static_key = None    # _weight_key(weight) the static views were packed at
        self.owner = None         # the optim.WeightPacker that refreshes `static` after every optimizer step

    def get(self, weight, groups, transposed):
        if transposed and groups == 1 and weight.shape[2] == 1 and weight.shape[3] == 1:
            return weight     # [K=co][M=ci] is exactly the OIHW layout of a 1x1 filter
        key = _weight_key(weight)
        if self.static_key == key and transposed in self.static:
            return self.static[transposed]
        ent = self.entries.get(transposed)
        if ent is not None and ent[0] == key:
            return ent[1]
        out = _pack_weight(weight, groups, transposed)
        self.entries[transposed] = (key, out)
        return out

    def get_wino(self, weight, transposed):
        """Winograd-transformed filters G g G^T in ge_wino.hip's operand order (same invalidation rule).  A layer of a model
        with a WeightPacker gets a PERSISTENT operand buffer at its first use, refreshed by the packer's batched launch after
        every optimizer step -- a HIP graph that reads it sees current weights at a fixed address whichever graph or eager pass
        ran first.  Inside a capture nothing is taken from (or put into) the lazy cache: such a buffer would live in one
        graph's private pool, or in allocator memory that can be freed, while other graphs keep reading it."""
        key = _weight_key(weight)
        slot = ("wino", transposed)
        if self.static_key == key and slot in self.static:
            return self.static[slot]
        capturing = weight.is_cuda and torch.cuda.is_current_stream_capturing()
        if self.owner is not None and self.static_key == key and not capturing:
            return self.owner.add_wino(self, weight, transposed)
        if capturing:
            return _pack_weight_wino(weight, transposed)      # a node of THIS graph, in its own pool
        ent = self.entries.get(slot)
        if ent is not None and ent[0] == key:
            return ent[1]
        out = _pack_weight_wino(weight, transposed)
        self.entries[slot] = (key, out)
        return out

    def get_lp(self, weight, groups, transposed, mode):
        """fp16 operand Wp[g][tap][m][c] for the "f16" kernels (same invalidation rule as `get`)."""
        key = _weight_key(weight)
        slot = (mode, transposed)
        if self.static_key == key and slot in self.static:
            return self.static[slot]
        ent = self.entries.get(slot)
        if ent is not None and ent[0] == key:
            return ent[1]
        out = _pack_weight_lp(weight, groups, transposed, mode)
        self.entries[slot] = (key, out)
        return out


def _pack_weight(weight, groups, transposed):
    Cout, Cin_g, kh, kw = weight.shape
    out = torch.empty(weight.numel(), device=weight.device, dtype=_f32)
    check(lib.ge_conv2d_pack_weight(_p(weight), _p(out), Cout, Cin_g, kh, kw, groups, int(transposed), _stream()),
          "conv2d_pack_weight")
    return out


def _pack_weight_wino(weight, transposed):
    Cout, Cin, _, _ = weight.shape
    M, C = (Cin, Cout) if transposed else (Cout, Cin)
    out = torch.empty(16 * weight.shape[0] * weight.shape[1], device=weight.device, dtype=_f32)
    check(lib.ge_wino3x3_pack_weight(_p(weight), _p(out), M, C, int(transposed), _stream()), "wino3x3_pack_weight")
    return out


def _pack_weight_lp(weight, groups, transposed, mode):
    Cout, Cin_g, kh, kw = weight.shape
    out = torch.empty(weight.numel(), device=weight.device, dtype=torch.int16)
    check(lp_fns(mode)["pack_weight"](_p(weight), _p(out), Cout, Cin_g, kh, kw, groups, int(transposed), _stream()),
          "conv2d_lp_pack_weight")
    return out


_FWD_SPLIT, _DGRAD_SPLIT = {}, {}    # layer geometry -> floats of split-K workspace (0: plain kernel), asked once per shape


def _conv_out(h, k, s, p):
    return (h + 2 * p - k) // s + 1


class _Conv2dFn(Function):
    """conv2d; with `with_skip` the op also returns an alias of x ("skip") whose incoming gradient is added to the
    data gradient inside the dgrad kernel's epilogue (one pass instead of dgrad + a separate tensor add)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, groups, cache, with_skip=False, want_stats=False):
        ctx.set_materialize_grads(False)   # no zero-fill launch for the (non-differentiable) stats output
        x = _c(x)
        weight = _c(weight)
        B, Cin, Hi, Wi = x.shape
        Cout, Cin_g, kh, kw = weight.shape
        if Cin != Cin_g * groups:
            raise RuntimeError(f"conv2d: input has {Cin} channels, weight expects {Cin_g * groups}")
        Ho, Wo = _conv_out(Hi, kh, stride, padding), _conv_out(Wi, kw, stride, padding)
        mode = CONV_PRECISION
        lp = lp_fns(mode) if mode != "f32" else None
        if lp is not None and not lp["supported"](Cin, Cout, groups):
            lp = None
        ctx.lp_dgrad = ctx.lp_wgrad = mode if lp is not None else None
        y = torch.empty((B, Cout, Ho, Wo), device=x.device, dtype=_f32)
        stats = None
        wino = False
        kt = KERNEL_TIMER
        ctx.hs = ACT_STORAGE == "f16" and H_GENERIC and kh == 3 and kw == 3 and stride == 1 and padding == 1 and \
            groups == 1 and bool(lib.ge_h_conv3x3_supported(B, Cin, Cout, Hi, Wi))
        if ctx.hs:
            # blocked-fp16 operand copy of x (kept for the weight gradient instead of x), fp32 NCHW result
            h_scale_forward_update(x.device)
            xh = torch.empty((B, Cin // 32, Hi, Wi, 32), device=x.device, dtype=torch.float16)
            check(lib.ge_h_from_f32(_p(x), _p(xh), B, Cin, Hi * Wi, 1.0, None, _stream()), "h_from_f32")
            wp = cache.get_lp(weight, 1, False, "f16") if cache is not None else _pack_weight_lp(weight, 1, False, "f16")
            if want_stats:
                stats = torch.empty((Cout, lib.ge_h_conv3x3_stat_parts(B, Hi, Wi), 3), device=x.device, dtype=_f32)
            t0 = kt.begin() if kt else None
            check(lib.ge_h_conv3x3_fwd_f32(_p(xh), _p(wp), _p(bias), _p(y), _p(stats), B, Cin, Cout, Hi, Wi, _stream()),
                  "h_conv3x3_fwd_f32")
            if kt:
                kt.end(t0, _conv_kind("convh_fwd", 3, 1, Cout, B * Ho * Wo, Cin * 9), 2.0 * B * Ho * Wo * Cout * Cin * 9,
                       2 * x.numel() + 2 * weight.numel() + 4 * y.numel())
            ctx.save_for_backward(xh, weight)
            ctx.xshape = (B, Cin, Hi, Wi)
            ctx.cfg = (stride, padding, groups, bias is not None, cache)
            ctx.params = (weight, bias)
            ctx.with_skip = with_skip
            outs = (y,)
            if with_skip:
                outs += (x.view_as(x),)
            if want_stats:
                ctx.mark_non_differentiable(stats)
                outs += (stats,)
            return outs if len(outs) > 1 else y
        if lp is not None:
            wp = cache.get_lp(weight, groups, False, mode) if cache is not None else \
                _pack_weight_lp(weight, groups, False, mode)
            if want_stats:
                parts = lp["fwd_stat_parts"](B, Cout, Ho, Wo, groups)
                stats = torch.empty((Cout, parts, 3), device=x.device, dtype=_f32)
            t0 = kt.begin() if kt else None
            check(lp["fwd"](_p(x), _p(wp), _p(bias), _p(y), _p(stats), B, Cin, Hi, Wi, Cout, Ho, Wo, kh, kw,
                            stride, padding, groups, 0, _stream()), "conv2d_lp_fwd")
        elif kh == 3 and kw == 3 and stride == 1 and padding == 1 and groups == 1 and _wino_plan(B, Cin, Cout, Hi, Wi)[0]:
            splits, ws_n = _wino_plan(B, Cin, Cout, Hi, Wi)
            u = cache.get_wino(weight, False) if cache is not None else _pack_weight_wino(weight, False)
            if want_stats and splits == 1:   # BatchNorm moments of y from the epilogue: one triple per workgroup and channel
                stats = torch.empty((Cout, lib.ge_wino3x3_stat_parts(B, Hi, Wi), 3), device=x.device, dtype=_f32)
            ws = torch.empty(ws_n, device=x.device, dtype=_f32) if ws_n else None
            t0 = kt.begin() if kt else None
            check(lib.ge_wino3x3_fwd(_p(x), _p(u), _p(bias), None, _p(y), _p(stats), _p(ws), B, Cin, Cout, Hi, Wi, _stream()),
                  "wino3x3_fwd")
            wino = True
        else:
            wp = cache.get(weight, groups, False) if cache is not None else _pack_weight(weight, groups, False)
            # layers whose tile grid cannot fill the chip (B*Ho*Wo of a few thousand) run split over K; that path has
            # no statistics epilogue, the BatchNorm behind it takes its moments from the (small) activation itself
            key = (B, Cin, Cout, Ho, Wo, kh, kw, groups)
            ws_n = _FWD_SPLIT.get(key)
            if ws_n is None:
                ws_n = _FWD_SPLIT[key] = lib.ge_conv2d_fwd_workspace(*key)
            if ws_n:
                ws = torch.empty(ws_n, device=x.device, dtype=_f32)
                t0 = kt.begin() if kt else None
                check(lib.ge_conv2d_fwd_splitk(_p(x), _p(wp), _p(bias), _p(y), B, Cin, Hi, Wi, Cout, Ho, Wo, kh, kw,
                                               stride, padding, groups, _p(ws), _stream()), "conv2d_fwd_splitk")
            else:
                if want_stats:   # BatchNorm moments of y, produced by the conv epilogue: [Cout][parts][3]
                    parts = lib.ge_conv2d_fwd_stat_parts(B, Cin, Cout, Ho, Wo, kh, kw, groups)
                    stats = torch.empty((Cout, parts, 3), device=x.device, dtype=_f32)
                t0 = kt.begin() if kt else None
                check(lib.ge_conv2d_fwd(_p(x), _p(wp), _p(bias), _p(y), _p(stats), B, Cin, Hi, Wi, Cout, Ho, Wo, kh, kw,
                                        stride, padding, groups, 0, _stream()), "conv2d_fwd")
        if kt:
            fl = 2.0 * B * Ho * Wo * Cout * Cin_g * kh * kw
            kt.end(t0, _conv_kind("conv_fwd", kh, stride, Cout, B * Ho * Wo, Cin_g * kh * kw), fl,
                   4 * (x.numel() + weight.numel() + y.numel()), executed=fl * 16.0 / 36.0 if wino else None)
        ctx.save_for_backward(x, weight)
        ctx.cfg = (stride, padding, groups, bias is not None, cache)
        ctx.params = (weight, bias)
        ctx.with_skip = with_skip
        outs = (y,)
        if with_skip:
            outs += (x.view_as(x),)
        if want_stats:
            if stats is not None:
                ctx.mark_non_differentiable(stats)
            outs += (stats,)
        return outs if len(outs) > 1 else y

    @staticmethod
    def backward(ctx, dy, *rest):
        dskip = rest[0] if (ctx.with_skip and rest) else None
        if dy is None:   # only the skip alias was consumed
            return dskip, None, None, None, None, None, None, None, None
        x, weight = ctx.saved_tensors
        stride, padding, groups, has_bias, cache = ctx.cfg
        dy = _c(dy)
        if ctx.hs:
            return _Conv2dFn._backward_h(ctx, x, weight, dy, dskip)
        B, Cin, Hi, Wi = x.shape
        Cout, Cin_g, kh, kw = weight.shape
        Ho, Wo = dy.shape[2], dy.shape[3]
        st = _stream()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            kt = KERNEL_TIMER
            wino = False
            add = _c(dskip) if dskip is not None else None
            if ctx.lp_dgrad:
                wp = cache.get_lp(weight, groups, True, ctx.lp_dgrad) if cache is not None else \
                    _pack_weight_lp(weight, groups, True, ctx.lp_dgrad)
                t0 = kt.begin() if kt else None
                check(lp_fns(ctx.lp_dgrad)["dgrad"](_p(dy), _p(wp), _p(add), _p(dx), B, Cin, Hi, Wi, Cout, Ho, Wo, kh, kw,
                                              stride, padding, groups, st), "conv2d_lp_dgrad")
            elif kh == 3 and kw == 3 and stride == 1 and padding == 1 and groups == 1 and _wino_plan(B, Cout, Cin, Hi, Wi)[0]:
                ws_n = _wino_plan(B, Cout, Cin, Hi, Wi)[1]
                ut = cache.get_wino(weight, True) if cache is not None else _pack_weight_wino(weight, True)
                ws = torch.empty(ws_n, device=x.device, dtype=_f32) if ws_n else None
                t0 = kt.begin() if kt else None
                check(lib.ge_wino3x3_fwd(_p(dy), _p(ut), None, _p(add), _p(dx), None, _p(ws), B, Cout, Cin, Hi, Wi, st),
                      "wino3x3_dgrad")
                wino = True
            else:
                wp = cache.get(weight, groups, True) if cache is not None else _pack_weight(weight, groups, True)
                key = (B, Cin, Hi, Wi, Cout, kh, kw, stride, groups)
                ws_n = _DGRAD_SPLIT.get(key)
                if ws_n is None:
                    ws_n = _DGRAD_SPLIT[key] = lib.ge_conv2d_dgrad_workspace(*key)
                t0 = kt.begin() if kt else None
                if ws_n:
                    ws = torch.empty(ws_n, device=x.device, dtype=_f32)
                    check(lib.ge_conv2d_dgrad_splitk(_p(dy), _p(wp), _p(add), _p(dx), B, Cin, Hi, Wi, Cout, Ho, Wo, kh,
                                                     kw, stride, padding, groups, _p(ws), st), "conv2d_dgrad_splitk")
                else:
                    check(lib.ge_conv2d_dgrad(_p(dy), _p(wp), _p(add), _p(dx), B, Cin, Hi, Wi, Cout, Ho, Wo, kh, kw,
                                              stride, padding, groups, st), "conv2d_dgrad")
            if kt:
                fl = 2.0 * B * Ho * Wo * Cout * Cin_g * kh * kw
                kt.end(t0, _conv_kind("conv_dgrad", kh, stride, Cin, B * Hi * Wi, Cout // groups * kh * kw), fl,
                       4 * (dy.numel() + weight.numel() + dx.numel()), executed=fl * 16.0 / 36.0 if wino else None)
        wparam, bparam = ctx.params
        db_fused = None
        if ctx.needs_input_grad[1]:
            wg_ws, wg_fn = (lp_fns(ctx.lp_wgrad)["wgrad_workspace"], lp_fns(ctx.lp_wgrad)["wgrad"]) if ctx.lp_wgrad else \
                (lib.ge_conv2d_wgrad_workspace, lib.ge_conv2d_wgrad)
            wino_w = 0
            if not ctx.lp_wgrad and kh == 3 and kw == 3 and stride == 1 and padding == 1 and groups == 1:
                wino_w = _wino_wgrad_ws(B, Cin, Cout, Hi, Wi)
            ws_n = wino_w if wino_w else wg_ws(B, Cin, Cout, Ho, Wo, kh, kw, groups)
            direct = DIRECT_GRAD_ACCUM and getattr(wparam, "_ge_flat", None) is not None and wparam.grad is not None
            dw = wparam.grad if direct else torch.empty_like(weight)
            kt = KERNEL_TIMER
            t0, t_mid = kt.begin_wgrad() if kt else (None, None)
            side = WGRAD_STREAM if (direct and kt is None) else None
            nsplit = 0
            if DEFER_SLABS and direct and kt is None and not ctx.lp_wgrad and not wino_w:      # leave the slabs to a batched reduce
                key = (B, Cin, Cout, Hi, Wi, Ho, Wo, kh, kw, stride, padding, groups)
                nsplit = _WGRAD_SPLITS.get(key)
                if nsplit is None:
                    nsplit = lib.ge_conv2d_wgrad_splits(*key)
                    if 4 * weight.numel() * nsplit > SLAB_DEFER_MAX:
                        nsplit = 0
                    _WGRAD_SPLITS[key] = nsplit
            mode = 3 if nsplit else int(direct)
            # bias gradient inside the weight-gradient pass (row sums of the dY tile the kernel stages anyway) where the
            # layer's kernel supports it: no separate read of dy on the main stream (ge_channel_sum)
            if has_bias and ctx.needs_input_grad[2] and not ctx.lp_wgrad and WGRAD_BIAS and not wino_w:
                bdirect = DIRECT_GRAD_ACCUM and getattr(bparam, "_ge_flat", None) is not None and bparam.grad is not None
                bkey = (B, Cin, Cout, Hi, Wi, Ho, Wo, kh, kw, stride, padding, groups)
                fuses = _WGRAD_FUSES_BIAS.get(bkey)
                if fuses is None:
                    fuses = _WGRAD_FUSES_BIAS[bkey] = bool(lib.ge_conv2d_wgrad_fuses_bias(*bkey))
                if fuses and bdirect == direct:
                    db_fused = bparam.grad if bdirect else torch.empty(Cout, device=x.device, dtype=_f32)
                    _dbp = _p(db_fused)
                    wg_call = lambda xs, dys, dws, wss, stream: lib.ge_conv2d_wgrad_bias(
                        xs, dys, dws, _dbp, wss, B, Cin, Hi, Wi, Cout, Ho, Wo, kh, kw, stride, padding, groups, mode, stream)
            if wino_w and has_bias and ctx.needs_input_grad[2] and WGRAD_BIAS:
                # the Winograd weight gradient loads every dy element: its c-tile-0 workgroups add up the bias gradient as well
                bdirect = DIRECT_GRAD_ACCUM and getattr(bparam, "_ge_flat", None) is not None and bparam.grad is not None
                if bdirect == direct:
                    db_fused = bparam.grad if bdirect else torch.empty(Cout, device=x.device, dtype=_f32)
            if wino_w and db_fused is not None:
                _dbw = _p(db_fused)
                wg_call = lambda xs, dys, dws, wss, stream: lib.ge_wino3x3_wgrad_bias(xs, dys, dws, _dbw, wss, B, Cin, Cout, Hi, Wi,
                                                                                      mode, stream)
            elif wino_w:
                wg_call = lambda xs, dys, dws, wss, stream: lib.ge_wino3x3_wgrad(xs, dys, dws, wss, B, Cin, Cout, Hi, Wi, mode,
                                                                                 stream)
            elif db_fused is None:
                wg_call = lambda xs, dys, dws, wss, stream: wg_fn(xs, dys, dws, wss, B, Cin, Hi, Wi, Cout, Ho, Wo, kh, kw,
                                                                 stride, padding, groups, mode, stream)
            ws = torch.empty(ws_n, device=x.device, dtype=_f32) if side is None else None
            if side is not None:
                side.wait_stream(torch.cuda.current_stream())     # dy and x are ready
                with torch.cuda.stream(side):
                    ws = torch.empty(ws_n, device=x.device, dtype=_f32)
                    check(wg_call(_p(x), _p(dy), _p(dw), _p(ws), side.cuda_stream), "conv2d_wgrad")
                x.record_stream(side)
                dy.record_stream(side)
            else:
                check(wg_call(_p(x), _p(dy), _p(dw), _p(ws), st), "conv2d_wgrad")
            if nsplit and db_fused is not None:      # slabs carry the bias row sums behind the weights: two entries, one stride
                sst, n_w = (side.cuda_stream if side is not None else st), weight.numel()
                _push_slabs(sst, ws, dw, n_w, nsplit, n_w + Cout)
                _push_slabs(sst, ws[n_w:], db_fused, Cout, nsplit, n_w + Cout)
            elif nsplit:
                _push_slabs(side.cuda_stream if side is not None else st, ws, dw, weight.numel(), nsplit)
            if direct:   # FlatParams learns about it from the parameter's AccumulateGrad node
                dw = None
            if kt:
                fl = 2.0 * B * Ho * Wo * Cout * Cin_g * kh * kw
                kt.end(t0, _conv_kind("conv_wgrad", kh, stride, Cout, Cin_g * kh * kw, B * Ho * Wo), fl,
                       4 * (x.numel() + dy.numel() + weight.numel()), split=t_mid, slab_bytes=4 * (ws_n + weight.numel()),
                       executed=fl * 16.0 / 36.0 if wino_w else None)
        if db_fused is not None:
            db = None if db_fused is bparam.grad else db_fused
        elif has_bias and ctx.needs_input_grad[2]:
            direct = DIRECT_GRAD_ACCUM and getattr(bparam, "_ge_flat", None) is not None and bparam.grad is not None
            db = bparam.grad if direct else torch.empty(Cout, device=x.device, dtype=_f32)
            part = torch.empty(B * Cout, device=x.device, dtype=_f32)
            check(lib.ge_channel_sum(_p(dy), _p(db), _p(part), B, Cout, Ho * Wo, int(direct), st), "channel_sum")
            if direct:
                db = None
        if dx is None and dskip is not None and ctx.needs_input_grad[0]:
            dx = dskip
        return dx, dw, db, None, None, None, None, None, None


def _conv2d_backward_h(ctx, xh, weight, dy, dskip):
    """Backward of a 3x3 / s1 / p1 conv that ran on the blocked-fp16 kernels (forward branch `ctx.hs`): the incoming fp32
    gradient is cast to blocked fp16 ONCE (times the loss scale) and feeds both the data- and the weight-gradient kernel."""
    stride, padding, groups, has_bias, cache = ctx.cfg
    B, Cin, Hi, Wi = ctx.xshape
    Cout = weight.shape[0]
    st = _stream()
    S, invS, hsp = h_scale_args(dy.device, cast=True)
    kt = KERNEL_TIMER
    flops = 2.0 * B * Hi * Wi * Cout * Cin * 9
    dyh = torch.empty((B, Cout // 32, Hi, Wi, 32), device=dy.device, dtype=torch.float16)
    check(lib.ge_h_from_f32(_p(dy), _p(dyh), B, Cout, Hi * Wi, S, hsp, st), "h_from_f32")
    dx = dw = db = None
    if ctx.needs_input_grad[0]:
        dx = torch.empty((B, Cin, Hi, Wi), device=dy.device, dtype=_f32)
        add = _c(dskip) if dskip is not None else None
        wpt = cache.get_lp(weight, 1, True, "f16") if cache is not None else _pack_weight_lp(weight, 1, True, "f16")
        t0 = kt.begin() if kt else None
        check(lib.ge_h_conv3x3_dgrad_f32(_p(dyh), _p(wpt), _p(add), _p(dx), invS, hsp, B, Cin, Cout, Hi, Wi, st),
              "h_conv3x3_dgrad_f32")
        if kt:
            kt.end(t0, _conv_kind("convh_dgrad", 3, 1, Cin, B * Hi * Wi, Cout * 9), flops,
                   2 * dyh.numel() + 2 * weight.numel() + 4 * dx.numel())
    wparam, bparam = ctx.params
    if ctx.needs_input_grad[1]:
        direct = DIRECT_GRAD_ACCUM and getattr(wparam, "_ge_flat", None) is not None and wparam.grad is not None
        dw = wparam.grad if direct else torch.empty_like(weight)
        ws_n = lib.ge_h_conv3x3_wgrad_workspace(B, Cin, Cout, Hi, Wi)
        side = WGRAD_STREAM if (direct and kt is None) else None
        if side is not None:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                ws = torch.empty(ws_n, device=dy.device, dtype=_f32)
                check(lib.ge_h_conv3x3_wgrad(_p(xh), _p(dyh), _p(dw), _p(ws), B, Cin, Cout, Hi, Wi, invS, hsp, int(direct),
                                             side.cuda_stream), "h_conv3x3_wgrad")
            xh.record_stream(side)
            dyh.record_stream(side)
        else:
            ws = torch.empty(ws_n, device=dy.device, dtype=_f32)
            t0, t_mid = kt.begin_wgrad() if kt else (None, None)
            check(lib.ge_h_conv3x3_wgrad(_p(xh), _p(dyh), _p(dw), _p(ws), B, Cin, Cout, Hi, Wi, invS, hsp, int(direct), st),
                  "h_conv3x3_wgrad")
            if kt:
                kt.end(t0, _conv_kind("convh_wgrad", 3, 1, Cout, Cin * 9, B * Hi * Wi), flops,
                       2 * (xh.numel() + dyh.numel()) + 4 * weight.numel(), split=t_mid,
                       slab_bytes=4 * (ws_n + weight.numel()))
        if direct:
            dw = None
    if has_bias and ctx.needs_input_grad[2]:
        direct = DIRECT_GRAD_ACCUM and getattr(bparam, "_ge_flat", None) is not None and bparam.grad is not None
        db = bparam.grad if direct else torch.empty(Cout, device=dy.device, dtype=_f32)
        part = torch.empty(B * Cout, device=dy.device, dtype=_f32)
        check(lib.ge_channel_sum(_p(dy), _p(db), _p(part), B, Cout, Hi * Wi, int(direct), st), "channel_sum")
        if direct:
            db = None
    if dx is None and dskip is not None and ctx.needs_input_grad[0]:
        dx = dskip
    return dx, dw, db, None, None, None, None, None, None


_Conv2dFn._backward_h = staticmethod(_conv2d_backward_h)


def _norm_sp(stride, padding):
    if isinstance(stride, (tuple, list)):
        stride = stride[0]
    if isinstance(padding, (tuple, list)):
        padding = padding[0]
    return int(stride), int(padding)


def conv2d(x, weight, bias=None, stride=1, padding=0, groups=1, cache=None, bn_stats=False):
    """conv2d; with bn_stats=True returns (y, stats) where stats are the per-tile BatchNorm moments of y computed in
    the conv epilogue (pass them to batch_norm(..., partial=stats))."""
    stride, padding = _norm_sp(stride, padding)
    return _Conv2dFn.apply(x, weight, bias, stride, padding, int(groups), cache, False, bool(bn_stats))


def conv2d_with_skip(x, weight, bias=None, stride=1, padding=0, groups=1, cache=None, bn_stats=False):
    """-> (conv2d(x), skip[, stats]) where skip aliases x.  Route every other use of x through `skip`: its gradient is
    then merged into this conv's data gradient in the kernel epilogue instead of by a separate add."""
    stride, padding = _norm_sp(stride, padding)
    return _Conv2dFn.apply(x, weight, bias, stride, padding, int(groups), cache, True, bool(bn_stats))


# --------------------------------------------------------------------------------------------------
# GEMM family
# --------------------------------------------------------------------------------------------------
def _gemm_raw(a, b, ta, tb, alpha=1.0, bias=None, bias_mode=0, relu=False, out=None, accumulate=False):
    """out[M,N] = alpha * op(a) @ op(b) (+bias); a, b are contiguous 2-D."""
    if ta:
        K, M = a.shape
        sam, sak = 1, M
    else:
        M, K = a.shape
        sam, sak = K, 1
    if tb:
        N, K2 = b.shape
        sbk, sbn = 1, K2
    else:
        K2, N = b.shape
        sbk, sbn = N, 1
    if K != K2:
        raise RuntimeError(f"gemm: inner dimensions differ ({K} vs {K2})")
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=_f32)
    check(lib.ge_gemm(_p(a), _p(b), _p(bias), _p(out), M, N, K, sam, sak, sbk, sbn, N, 1, 1, 0, 0, 0, float(alpha),
                      bias_mode, int(relu), int(accumulate), _stream()), "gemm")
    return out


class _MatMulFn(Function):
    @staticmethod
    def forward(ctx, a, b, ta, tb, alpha):
        a, b = _c(a), _c(b)
        ctx.save_for_backward(a, b)
        ctx.cfg = (ta, tb, alpha)
        return _gemm_raw(a, b, ta, tb, alpha)

    @staticmethod
    def backward(ctx, dc):
        a, b = ctx.saved_tensors
        ta, tb, alpha = ctx.cfg
        dc = _c(dc)
        da = db = None
        if ctx.needs_input_grad[0]:
            if not ta:
                da = _gemm_raw(dc, b, False, not tb, alpha)
            else:
                da = _gemm_raw(b, dc, tb, True, alpha)
        if ctx.needs_input_grad[1]:
            if not tb:
                db = _gemm_raw(a, dc, not ta, False, alpha)
            else:
                db = _gemm_raw(dc, a, True, ta, alpha)
        return da, db, None, None, None


def matmul(a, b, transpose_a=False, transpose_b=False, alpha=1.0):
    """alpha * op(a) @ op(b) for 2-D operands."""
    return _MatMulFn.apply(a, b, bool(transpose_a), bool(transpose_b), float(alpha))


def _direct(param):
    """True when `param`'s gradient may be accumulated straight into its flat gradient buffer (trainer's backward)."""
    return DIRECT_GRAD_ACCUM and getattr(param, "_ge_flat", None) is not None and param.grad is not None


class _LinearFn(Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        x2 = _c(x).reshape(-1, x.shape[-1])
        ctx.params = (weight, bias)
        weight = _c(weight)
        y = _gemm_raw(x2, weight, False, True, 1.0, bias, 2 if bias is not None else 0)
        ctx.save_for_backward(x2, weight)
        ctx.has_bias = bias is not None
        ctx.in_shape = x.shape
        return y.reshape(*x.shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, weight = ctx.saved_tensors
        wparam, bparam = ctx.params
        dy2 = _c(dy).reshape(-1, weight.shape[0])
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = _gemm_raw(dy2, weight, False, False).reshape(ctx.in_shape)
        bias_done = False
        if ctx.needs_input_grad[1]:
            wd = _direct(wparam) and wparam.grad.is_contiguous()
            K, M = dy2.shape
            N = x2.shape[1]
            if ctx.has_bias and ctx.needs_input_grad[2] and lib.ge_gemm_rowsum_ok(M, N, K, 1):
                # dW = dY^T X and db = column sums of dY in ONE launch (the kernel's A operand IS dY^T)
                bd = _direct(bparam)
                out = wparam.grad if wd else torch.empty((M, N), device=dy2.device, dtype=_f32)
                db_t = bparam.grad if bd else torch.empty(M, device=dy2.device, dtype=_f32)
                check(lib.ge_gemm_rowsum(_p(dy2), _p(x2), _p(out), M, N, K, 1, M, N, 1, N, 1, 1, 0, 0, 0, 1.0, int(wd),
                                         _p(db_t), int(bd), _stream()), "gemm_rowsum")
                dw = None if wd else out
                db = None if bd else db_t
                bias_done = True
            elif wd:
                # weight gradient accumulated by the GEMM epilogue into the flat buffer: no ATen add, no allocation
                _gemm_raw(dy2, x2, True, False, out=wparam.grad, accumulate=True)
            else:
                dw = _gemm_raw(dy2, x2, True, False)
        if ctx.has_bias and ctx.needs_input_grad[2] and not bias_done:
            if _direct(bparam):
                check(lib.ge_colsum_accumulate(_p(dy2), _p(bparam.grad), dy2.shape[0], dy2.shape[1], _stream()),
                      "colsum_accumulate")
            else:
                db = torch.empty(weight.shape[0], device=dy2.device, dtype=_f32)
                check(lib.ge_colsum(_p(dy2), _p(db), dy2.shape[0], dy2.shape[1], _stream()), "colsum")
        return dx, dw, db


def linear(x, weight, bias=None):
    return _LinearFn.apply(x, weight, bias)


# --------------------------------------------------------------------------------------------------
# normalisation
# --------------------------------------------------------------------------------------------------
# Batch sizes of the independent forward passes that were concatenated along the batch dimension (see bn_segments).
BN_SEGMENTS = None
# SyncBN exchanges issued since the counter was last reset: [forward all-gathers, backward all-reduces, bytes sent]
SYNC_BN_STATS = [0, 0, 0]
# SyncBN over several concatenated passes (source + target frames): all segments in each of the layer's launches (GE_SYNCBN_SEGS=0:
# one launch per segment on either side of the exchange, the round-5 form; the results are the same bits)
SYNC_BN_SEGS = os.environ.get("GE_SYNCBN_SEGS", "1") != "0"


class bn_segments:
    """`with bn_segments([8, 8, 32]): net(torch.cat([a, b, c]))` computes what `net(a); net(b); net(c)` compute -- every
    train-mode BatchNorm takes its statistics (and updates its running statistics, in order) per segment -- while the
    convolutions see one batch: one launch with a 3x larger N instead of three small ones.  GroupNorm / LayerNorm are
    per-sample and need no help.  Used by the trainer for the source / target / clip passes of one step."""

    def __init__(self, sizes):
        self.sizes = tuple(int(v) for v in sizes)

    def __enter__(self):
        global BN_SEGMENTS
        self.prev = BN_SEGMENTS
        BN_SEGMENTS = self.sizes if len(self.sizes) > 1 else None
        return self

    def __exit__(self, *exc):
        global BN_SEGMENTS
        BN_SEGMENTS = self.prev
        return False


def _segment_bounds(B, segments):
    if not segments or len(segments) < 2:
        return [(0, B)]
    if sum(segments) != B:
        raise RuntimeError(f"batch_norm: segments {tuple(segments)} do not add up to the batch size {B}")
    out, b0 = [], 0
    for n in segments:
        out.append((b0, n))
        b0 += n
    return out


class _BatchNormFn(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, residual, training, momentum, eps, relu, group,
                partial=None, segments=None):
        x = _c(x)
        B, C, H, W = x.shape
        HW = H * W
        st = _stream()
        dev = x.device
        world = 1
        relu = int(relu)            # fused activation code: 0 none, 1 ReLU, 2 GELU (erf)
        if relu == 2 and residual is not None:
            raise RuntimeError("batch_norm: the fused GELU has no residual form")
        bounds = _segment_bounds(B, segments if training else None)
        S = len(bounds)
        plane = C * HW * 4      # bytes per sample
        if training:
            mean = torch.empty((S, C), device=dev, dtype=_f32)
            invstd = torch.empty((S, C), device=dev, dtype=_f32)
            # per segment: (pointer to its [C][parts][3] moments, parts, channel stride in floats)
            parts, keep = [], []
            width = 0
            if partial is not None and S > 1:
                # moments from the conv epilogue cover fixed runs of `width` output positions (ge_conv2d_fwd_stat_parts);
                # they can be split by segment only if every boundary falls between two runs
                nb = partial.numel() // (C * 3)
                n128, n64 = 2 * ((B * HW + 127) // 128), 2 * ((B * HW + 63) // 64)
                width = 64 if (nb == n128 and nb != n64) else (32 if (nb == n64 and nb != n128) else 0)
                if nb * 128 == B * HW:      # the Winograd kernels' epilogue: one triple per 128 positions (4 x 32 / 8 x 16 pixels of ONE image)
                    width = 128
                if width and any((b0 * HW) % width or (bs * HW) % width for b0, bs in bounds):
                    width = 0
            # small layers without SyncBN: the whole forward of a segment is ONE launch (moments from the conv-epilogue
            # partials or from x, running statistics, apply): ge_bn_fwd_channel
            fused = [group is None and bool(lib.ge_bn_channel_ok(bs, HW)) for _b0, bs in bounds]
            # SyncBN, small layers, no usable conv-epilogue moments: the segments' local moments come from x in one launch
            # (ge_bn_stats_channel_segs) instead of a statistics pass + finalize per segment
            sync_x_segs = SYNC_BN_SEGS and group is not None and 1 < S <= 16 and not (partial is not None and width) and \
                all(bool(lib.ge_bn_channel_ok(bs, HW)) for _b0, bs in bounds)
            for (b0, bs), fz in zip(bounds, fused):
                if sync_x_segs:
                    parts.append((None, 0, 0))
                elif partial is not None and S == 1:
                    nb = partial.numel() // (C * 3)
                    parts.append((_p(partial), nb, nb * 3))
                elif partial is not None and width:
                    nb = partial.numel() // (C * 3)
                    parts.append((_p(partial) + (b0 * HW // width) * 12, bs * HW // width, nb * 3))
                elif fz:
                    parts.append((None, 0, 0))
                else:
                    nbs = lib.ge_bn_num_partials(bs, HW)
                    own = torch.empty(C * nbs * 3, device=dev, dtype=_f32)
                    keep.append(own)
                    check(lib.ge_bn_stats_partial(_p(x) + b0 * plane, _p(own), bs, C, HW, st), "bn_stats_partial")
                    parts.append((_p(own), nbs, nbs * 3))
            if group is None:
                pass      # finalized per segment below, in segment order (the running statistics are updated in that order)
            else:
                import torch.distributed as dist

                world = dist.get_world_size(group)
                stats = torch.empty((S, C * 3), device=dev, dtype=_f32)
                # small layers: merging the ranks' moments, the running statistics and the apply pass are one launch per
                # segment (ge_bn_fwd_channel reading the gathered [world] triples of its channel)
                # (all segments or none: the running statistics must be updated in segment order)
                sync_fused = [all(bool(lib.ge_bn_channel_ok(bs, HW)) for _b0, bs in bounds)] * S
                # ... and, with the segments' moments side by side in the conv epilogue's buffer, ONE launch for all of them on
                # either side of the exchange (ge_bn_finalize_segs, ge_bn_fwd_channel_segs_sync): 3 launches per layer, not 2 S + 1
                sync_segs = SYNC_BN_SEGS and 1 < S <= 16 and sync_fused[0] and partial is not None and bool(width)
                # big layers: the local finalize still takes all segments at once; behind the exchange every segment is one
                # launch (ge_bn_fwd_merge_apply_sync) instead of finalize + apply
                sync_merge = SYNC_BN_SEGS and not sync_fused[0] and HW % 4 == 0
                local_segs = sync_segs or (sync_merge and 1 < S <= 16 and partial is not None and bool(width))
                if sync_x_segs:
                    import ctypes

                    seg_arr = (ctypes.c_int * (4 * S))(*[int(v) for b0, bs in bounds for v in (b0, bs, 0, 0)])
                    check(lib.ge_bn_stats_channel_segs(_p(x), seg_arr, S, C, HW, _p(stats), st), "bn_stats_channel_segs")
                    sync_segs = True      # behind the exchange: all segments in one launch as well
                elif local_segs:
                    import ctypes

                    nb = partial.numel() // (C * 3)
                    seg_arr = (ctypes.c_int * (4 * S))(*[int(v) for b0, bs in bounds
                                                         for v in (b0, bs, b0 * HW // width, bs * HW // width)])
                    check(lib.ge_bn_finalize_segs(_p(partial), nb * 3, 3, seg_arr, S, C, HW, _p(stats), st), "bn_finalize_segs")
                else:
                    for s, (ptr, nbs, cstride) in enumerate(parts):
                        check(lib.ge_bn_finalize(ptr, cstride, 3, nbs, C, eps, momentum, _p(stats[s]), None, None, None,
                                                 None, st), "bn_finalize_local")
                gathered = torch.empty((world, S, C * 3), device=dev, dtype=_f32)   # one exchange for all segments
                dist.all_gather_into_tensor(gathered.view(-1), stats.view(-1), group=group)
                SYNC_BN_STATS[0] += 1
                SYNC_BN_STATS[2] += 4 * stats.numel()
                for s in range(S):
                    if sync_fused[s] or sync_merge:
                        continue
                    check(lib.ge_bn_finalize(_p(gathered) + s * C * 12, 3, S * C * 3, world, C, eps, momentum, None,
                                             _p(mean[s]), _p(invstd[s]), _p(running_mean), _p(running_var), st),
                          "bn_finalize_sync")
        else:
            mean = running_mean.reshape(1, C)
            invstd = torch.rsqrt(running_var + eps).reshape(1, C)
        y = torch.empty_like(x)
        res = _c(residual) if residual is not None else None
        # every segment a small layer (and no SyncBN): ALL segments in one launch -- the channel's workgroup walks them in
        # order (ge_bn_fwd_channel_segs): half the BatchNorm launches of a merged source + target pass
        multi = training and group is None and 1 < S <= 16 and all(fused)
        if multi:
            if all(pp[0] is None for pp in parts):
                base, cstride, segs = None, 0, [(b0, bs, 0, 0) for b0, bs in bounds]
            elif partial is not None and width:
                nb = partial.numel() // (C * 3)
                base, cstride = _p(partial), nb * 3
                segs = [(b0, bs, b0 * HW // width, bs * HW // width) for b0, bs in bounds]
            else:
                multi = False
        if multi:
            import ctypes

            seg = (ctypes.c_int * (4 * S))(*[int(v) for sg in segs for v in sg])
            check(lib.ge_bn_fwd_channel_segs(_p(x), base, cstride, 3, seg, S, _p(gamma), _p(beta), _p(res), _p(y),
                                             _p(mean), _p(invstd), _p(running_mean), _p(running_var), C, HW, eps,
                                             momentum, int(relu), st), "bn_fwd_channel_segs")
        if training and group is not None and sync_segs:
            check(lib.ge_bn_fwd_channel_segs_sync(_p(x), _p(gathered), world, seg_arr, S, _p(gamma), _p(beta), _p(res), _p(y),
                                                  _p(mean), _p(invstd), _p(running_mean), _p(running_var), C, HW, eps,
                                                  momentum, int(relu), st), "bn_fwd_channel_segs_sync")
            multi = True      # (nothing left for the per-segment loop)
        for s, (b0, bs) in enumerate(bounds):
            if multi:
                break
            off = b0 * plane
            if training and group is not None and sync_merge:
                check(lib.ge_bn_fwd_merge_apply_sync(_p(x) + off, _p(gathered) + s * C * 12, 3, S * C * 3, world, _p(gamma),
                                                     _p(beta), None if res is None else _p(res) + off, _p(y) + off,
                                                     _p(mean[s]), _p(invstd[s]), _p(running_mean), _p(running_var), bs, C, HW,
                                                     eps, momentum, int(relu), st), "bn_fwd_merge_apply_sync")
                continue
            if training and group is not None and sync_fused[s]:
                check(lib.ge_bn_fwd_channel(_p(x) + off, _p(gathered) + s * C * 12, 3, S * C * 3, world, _p(gamma),
                                            _p(beta), None if res is None else _p(res) + off, _p(y) + off, _p(mean[s]),
                                            _p(invstd[s]), _p(running_mean), _p(running_var), bs, C, HW, eps, momentum,
                                            int(relu), st), "bn_fwd_channel_sync")
                continue
            if training and fused[s]:
                ptr, nbs, cstride = parts[s]
                check(lib.ge_bn_fwd_channel(_p(x) + off, ptr, cstride, 3, nbs, _p(gamma), _p(beta),
                                            None if res is None else _p(res) + off, _p(y) + off, _p(mean[s]),
                                            _p(invstd[s]), _p(running_mean), _p(running_var), bs, C, HW, eps, momentum,
                                            int(relu), st), "bn_fwd_channel")
                continue
            if training and group is None:
                ptr, nbs, cstride = parts[s]
                if ptr is not None and lib.ge_bn_fwd_merge_apply_ok(nbs, HW):      # finalize + apply in one launch
                    check(lib.ge_bn_fwd_merge_apply(_p(x) + off, ptr, cstride, 3, nbs, _p(gamma), _p(beta),
                                                    None if res is None else _p(res) + off, _p(y) + off, _p(mean[s]),
                                                    _p(invstd[s]), _p(running_mean), _p(running_var), bs, C, HW, eps,
                                                    momentum, int(relu), st), "bn_fwd_merge_apply")
                    continue
                check(lib.ge_bn_finalize(ptr, cstride, 3, nbs, C, eps, momentum, None, _p(mean[s]), _p(invstd[s]),
                                         _p(running_mean), _p(running_var), st), "bn_finalize")
            check(lib.ge_bn_apply(_p(x) + off, _p(mean[s]), _p(invstd[s]), _p(gamma), _p(beta),
                                  None if res is None else _p(res) + off, _p(y) + off, bs, C, HW, int(relu), st),
                  "bn_apply")
        # ReLU mask for backward: recomputed from x when there is no residual, else read from the saved output
        ctx.save_for_backward(x, gamma, mean, invstd, y if (relu == 1 and residual is not None) else None, beta)
        ctx.cfg = (training, relu, residual is not None, group, world, gamma is not None, bounds)
        ctx.params = (gamma, beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, invstd, out, beta = ctx.saved_tensors
        training, relu, has_res, group, world, affine, bounds = ctx.cfg
        recompute = relu if (relu and not has_res) else 0      # 1: ReLU mask, 2: GELU derivative, both from fma(x, sc, sh)
        dy = _c(dy)
        B, C, H, W = x.shape
        HW = H * W
        st = _stream()
        dev = x.device
        S = len(bounds)
        plane = C * HW * 4
        sums = torch.empty((S, C, 2), device=dev, dtype=_f32)
        gparam, bparam = ctx.params
        dgamma = dbeta = None
        direct = False
        if affine:
            direct = DIRECT_GRAD_ACCUM and getattr(gparam, "_ge_flat", None) is not None and gparam.grad is not None \
                and getattr(bparam, "_ge_flat", None) is not None and bparam.grad is not None
            dgamma = gparam.grad if direct else torch.empty(C, device=dev, dtype=_f32)
            dbeta = bparam.grad if direct else torch.empty(C, device=dev, dtype=_f32)
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if (has_res and ctx.needs_input_grad[5]) else None
        # small layers without SyncBN (and train mode): reduce + finalize + apply of a segment in ONE launch
        fused = [training and group is None and bool(lib.ge_bn_channel_ok(bs, HW)) for _b0, bs in bounds]
        # segments on the two-launch path write the affine gradients in the APPLY loop, i.e. after the other segments'
        # reduce calls: with mixed segments everything accumulates into zero-initialised buffers
        acc_all = training and group is None and any(not f and lib.ge_bn_bwd_two_launch_ok(bs, HW)
                                                       for f, (_b0, bs) in zip(fused, bounds))
        if acc_all and affine and not direct and S > 1:
            dgamma.zero_()
            dbeta.zero_()
        acc_all = acc_all and S > 1
        multi = 1 < S <= 4 and all(fused)
        if multi:       # all segments in one launch (see forward)
            import ctypes

            seg = (ctypes.c_int * (4 * S))(*[int(v) for b0, bs in bounds for v in (b0, bs, 0, 0)])
            check(lib.ge_bn_bwd_channel_segs(_p(dy), _p(x), _p(out), _p(mean), _p(invstd), _p(gamma), _p(beta), recompute,
                                             _p(dgamma), _p(dbeta), int(direct), seg, S, _p(dx), _p(dres), C, HW, st),
                  "bn_bwd_channel_segs")
        # SyncBN, every segment a small layer: both halves take all segments per launch (forward: sync_segs)
        sync_segs = SYNC_BN_SEGS and training and group is not None and 1 < S <= 16 and \
            all(bool(lib.ge_bn_channel_ok(bs, HW)) for _b0, bs in bounds)
        if sync_segs:
            import ctypes

            seg_arr = (ctypes.c_int * (4 * S))(*[int(v) for b0, bs in bounds for v in (b0, bs, 0, 0)])
            check(lib.ge_bn_bwd_reduce_channel_segs(_p(dy), _p(x), _p(out), _p(mean), _p(invstd), _p(gamma), _p(beta), recompute,
                                                    _p(sums), _p(dgamma), _p(dbeta), int(direct), seg_arr, S, C, HW, st),
                  "bn_bwd_reduce_channel_segs")
        two = {}
        for s, (b0, bs) in enumerate(bounds):
            if multi or sync_segs:
                break
            off = b0 * plane
            if fused[s]:
                check(lib.ge_bn_bwd_channel(_p(dy) + off, _p(x) + off, None if out is None else _p(out) + off, _p(mean[s]),
                                            _p(invstd[s]), _p(gamma), _p(beta), recompute, _p(dgamma), _p(dbeta),
                                            int(direct or acc_all or s > 0), 1.0 / (bs * HW), _p(dx) + off,
                                            None if dres is None else _p(dres) + off, bs, C, HW, st), "bn_bwd_channel")
                continue
            if training and group is not None and lib.ge_bn_channel_ok(bs, HW):     # SyncBN, small layer: one launch
                check(lib.ge_bn_bwd_reduce_channel(_p(dy) + off, _p(x) + off, None if out is None else _p(out) + off,
                                                   _p(mean[s]), _p(invstd[s]), _p(gamma), _p(beta), recompute,
                                                   _p(sums[s]), _p(dgamma), _p(dbeta), int(direct or acc_all or s > 0), bs, C, HW,
                                                   st), "bn_bwd_reduce_channel")
                continue
            partial = torch.empty(C * lib.ge_bn_num_partials(bs, HW) * 2, device=dev, dtype=_f32)
            if training and group is None and lib.ge_bn_bwd_two_launch_ok(bs, HW):
                # no SyncBN: the apply pass folds the per-slice sums itself (no finalize launch on the critical stream)
                check(lib.ge_bn_bwd_partials(_p(dy) + off, _p(x) + off, None if out is None else _p(out) + off, _p(mean[s]),
                                             _p(invstd[s]), _p(gamma), _p(beta), recompute, _p(partial), bs, C, HW, st),
                      "bn_bwd_partials")
                two[s] = partial
                continue
            check(lib.ge_bn_bwd_reduce(_p(dy) + off, _p(x) + off, None if out is None else _p(out) + off, _p(mean[s]),
                                       _p(invstd[s]), _p(gamma), _p(beta), recompute, _p(partial), _p(sums[s]),
                                       _p(dgamma), _p(dbeta), int(direct or acc_all or s > 0), bs, C, HW, st), "bn_bwd_reduce")
        dgamma_t, dbeta_t = dgamma, dbeta       # the two-launch apply pass writes the affine gradients
        if direct:
            dgamma = dbeta = None
        scale = 1
        if not training:
            sums = torch.zeros_like(sums)
        elif group is not None:
            import torch.distributed as dist

            dist.all_reduce(sums, group=group)      # all segments in one exchange
            SYNC_BN_STATS[1] += 1
            SYNC_BN_STATS[2] += 4 * sums.numel()
            scale = world
        if sync_segs:
            inv = (ctypes.c_float * S)(*[1.0 / (bs * HW * scale) for _b0, bs in bounds])
            check(lib.ge_bn_bwd_apply_channel_segs(_p(dy), _p(x), _p(out), _p(mean), _p(invstd), _p(gamma), _p(beta), recompute,
                                                   _p(sums), inv, seg_arr, S, _p(dx), _p(dres), C, HW, st),
                  "bn_bwd_apply_channel_segs")
        for s, (b0, bs) in enumerate(bounds):
            if fused[s] or sync_segs:
                continue
            off = b0 * plane
            if s in two:
                check(lib.ge_bn_bwd_apply_partials(_p(dy) + off, _p(x) + off, None if out is None else _p(out) + off,
                                                   _p(mean[s]), _p(invstd[s]), _p(gamma), _p(beta), recompute, _p(two[s]),
                                                   _p(dgamma_t), _p(dbeta_t), int(direct or acc_all or s > 0), 1.0 / (bs * HW),
                                                   _p(dx) + off, None if dres is None else _p(dres) + off, bs, C, HW, st),
                      "bn_bwd_apply_partials")
                continue
            check(lib.ge_bn_bwd_apply(_p(dy) + off, _p(x) + off, None if out is None else _p(out) + off, _p(mean[s]),
                                      _p(invstd[s]), _p(gamma), _p(beta), recompute, _p(sums[s]),
                                      1.0 / (bs * HW * scale), _p(dx) + off, None if dres is None else _p(dres) + off,
                                      bs, C, HW, st), "bn_bwd_apply")
        return dx, dgamma, dbeta, None, None, dres, None, None, None, None, None, None, None


def batch_norm(x, gamma, beta, running_mean, running_var, training, momentum=0.1, eps=1e-5, residual=None, relu=False,
               group=None, partial=None, segments=None):
    """BatchNorm2d (+ optional fused residual add and ReLU; relu="gelu": fused erf-GELU, no residual).  `group`: process
    group for SyncBN statistics;
    `partial`: per-tile moments of x from conv2d(..., bn_stats=True) (skips the statistics pass over x);
    `segments`: batch sizes of independent passes concatenated in x (train mode: statistics per segment)."""
    act = 2 if relu == "gelu" else int(bool(relu))
    return _BatchNormFn.apply(x, gamma, beta, running_mean, running_var, residual, bool(training), float(momentum),
                              float(eps), act, group, partial, segments)


class _GroupNormFn(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, G, eps, relu):
        x = _c(x)
        B, C = x.shape[0], x.shape[1]
        HW = x.numel() // (B * C)
        mean = torch.empty(B * G, device=x.device, dtype=_f32)
        invstd = torch.empty(B * G, device=x.device, dtype=_f32)
        y = torch.empty_like(x)
        check(lib.ge_groupnorm_fwd(_p(x), _p(gamma), _p(beta), _p(y), _p(mean), _p(invstd), B, C, HW, G, eps,
                                   int(relu), _stream()), "groupnorm_fwd")
        ctx.save_for_backward(x, gamma, mean, invstd, y if relu else None)
        ctx.cfg = (G, gamma is not None)
        ctx.params = (gamma, beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, invstd, out = ctx.saved_tensors
        G, affine = ctx.cfg
        gparam, bparam = ctx.params
        dy = _c(dy)
        B, C = x.shape[0], x.shape[1]
        HW = x.numel() // (B * C)
        dev = x.device
        st = _stream()
        dx = torch.empty_like(x)
        part = torch.empty((2, B, C), device=dev, dtype=_f32)
        direct = affine and _direct(gparam) and _direct(bparam)
        dgamma = torch.empty(C, device=dev, dtype=_f32) if (affine and not direct) else None
        dbeta = torch.empty(C, device=dev, dtype=_f32) if (affine and not direct) else None
        check(lib.ge_groupnorm_bwd(_p(dy), _p(x), _p(out), _p(gamma), _p(mean), _p(invstd), _p(dx), _p(part[0]),
                                   _p(part[1]), _p(dgamma), _p(dbeta), B, C, HW, G, st), "groupnorm_bwd")
        if direct:   # per-sample partials summed straight into the flat gradient buffers
            check(lib.ge_colsum_accumulate2(_p(part[0]), _p(gparam.grad), _p(part[1]), _p(bparam.grad), B, C, st),
                  "colsum_accumulate2")
        return dx, dgamma, dbeta, None, None, None


def group_norm(x, num_groups, gamma=None, beta=None, eps=1e-5, relu=False):
    return _GroupNormFn.apply(x, gamma, beta, int(num_groups), float(eps), bool(relu))


class _LayerNormFn(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        x = _c(x)
        D = x.shape[-1]
        R = x.numel() // D
        mean = torch.empty(R, device=x.device, dtype=_f32)
        invstd = torch.empty(R, device=x.device, dtype=_f32)
        y = torch.empty_like(x)
        check(lib.ge_layernorm_fwd(_p(x), _p(gamma), _p(beta), _p(y), _p(mean), _p(invstd), R, D, eps, _stream()),
              "layernorm_fwd")
        ctx.save_for_backward(x, gamma, mean, invstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, invstd = ctx.saved_tensors
        dy = _c(dy)
        D = x.shape[-1]
        R = x.numel() // D
        dev = x.device
        dx = torch.empty_like(x)
        dgamma = dbeta = None
        gp = bp = None
        if gamma is not None:
            nblk = lib.ge_layernorm_bwd_blocks(R)
            part = torch.empty((2, nblk, D), device=dev, dtype=_f32)
            gp, bp = part[0], part[1]
            dgamma = torch.empty(D, device=dev, dtype=_f32)
            dbeta = torch.empty(D, device=dev, dtype=_f32)
        check(lib.ge_layernorm_bwd(_p(dy), _p(x), _p(gamma), _p(mean), _p(invstd), _p(dx), _p(gp), _p(bp), _p(dgamma),
                                   _p(dbeta), R, D, _stream()), "layernorm_bwd")
        return dx, dgamma, dbeta, None


def layer_norm(x, gamma=None, beta=None, eps=1e-5):
    return _LayerNormFn.apply(x, gamma, beta, float(eps))


# --------------------------------------------------------------------------------------------------
# spatial ops
# --------------------------------------------------------------------------------------------------
class _UpsampleFn(Function):
    @staticmethod
    def forward(ctx, x, add, Ho, Wo):
        x = _c(x)
        B, C, Hi, Wi = x.shape
        y = torch.empty((B, C, Ho, Wo), device=x.device, dtype=_f32)
        a = _c(add) if add is not None else None
        check(lib.ge_upsample_bilinear_fwd(_p(x), _p(a), _p(y), B, C, Hi, Wi, Ho, Wo, _stream()), "upsample_fwd")
        ctx.shape = (B, C, Hi, Wi, Ho, Wo)
        ctx.has_add = add is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        B, C, Hi, Wi, Ho, Wo = ctx.shape
        dy = _c(dy)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((B, C, Hi, Wi), device=dy.device, dtype=_f32)
            check(lib.ge_upsample_bilinear_bwd(_p(dy), _p(dx), B, C, Hi, Wi, Ho, Wo, _stream()), "upsample_bwd")
        dadd = dy if (ctx.has_add and ctx.needs_input_grad[1]) else None
        return dx, dadd, None, None


def upsample_bilinear(x, size, add=None):
    """F.interpolate(x, size, mode='bilinear', align_corners=True) (+ add).  Same-size calls are the identity."""
    Ho, Wo = int(size[0]), int(size[1])
    if add is None and x.shape[2] == Ho and x.shape[3] == Wo:
        return x
    return _UpsampleFn.apply(x, add, Ho, Wo)


class _MaxPoolFn(Function):
    @staticmethod
    def forward(ctx, x, k, s, p):
        x = _c(x)
        B, C, Hi, Wi = x.shape
        Ho, Wo = _conv_out(Hi, k, s, p), _conv_out(Wi, k, s, p)
        y = torch.empty((B, C, Ho, Wo), device=x.device, dtype=_f32)
        arg = torch.empty((B, C, Ho, Wo), device=x.device, dtype=torch.uint8)
        check(lib.ge_maxpool2d_fwd(_p(x), _p(y), _p(arg), B, C, Hi, Wi, Ho, Wo, k, s, p, _stream()), "maxpool_fwd")
        ctx.save_for_backward(arg)
        ctx.cfg = (B, C, Hi, Wi, Ho, Wo, k, s, p)
        return y

    @staticmethod
    def backward(ctx, dy):
        (arg,) = ctx.saved_tensors
        B, C, Hi, Wi, Ho, Wo, k, s, p = ctx.cfg
        dy = _c(dy)
        dx = torch.empty((B, C, Hi, Wi), device=dy.device, dtype=_f32)
        check(lib.ge_maxpool2d_bwd(_p(dy), _p(arg), _p(dx), B, C, Hi, Wi, Ho, Wo, k, s, p, _stream()), "maxpool_bwd")
        return dx, None, None, None


def max_pool2d(x, kernel_size, stride=None, padding=0):
    k = kernel_size[0] if isinstance(kernel_size, (tuple, list)) else kernel_size
    s = stride if stride is not None else k
    s = s[0] if isinstance(s, (tuple, list)) else s
    p = padding[0] if isinstance(padding, (tuple, list)) else padding
    return _MaxPoolFn.apply(x, int(k), int(s), int(p))


class _AvgPoolFn(Function):
    @staticmethod
    def forward(ctx, x, r):
        x = _c(x)
        B, C, Hi, Wi = x.shape
        y = torch.empty((B, C, Hi // r, Wi // r), device=x.device, dtype=_f32)
        check(lib.ge_avgpool2d_fwd(_p(x), _p(y), B, C, Hi, Wi, r, _stream()), "avgpool_fwd")
        ctx.cfg = (B, C, Hi, Wi, r)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, C, Hi, Wi, r = ctx.cfg
        dy = _c(dy)
        dx = torch.empty((B, C, Hi, Wi), device=dy.device, dtype=_f32)
        check(lib.ge_avgpool2d_bwd(_p(dy), _p(dx), B, C, Hi, Wi, r, _stream()), "avgpool_bwd")
        return dx, None


def avg_pool2d(x, r):
    """F.avg_pool2d(x, r, r)."""
    return _AvgPoolFn.apply(x, int(r))


class _PlaneMeanFn(Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        B, C, H, W = x.shape
        y = torch.empty((B, C, 1, 1), device=x.device, dtype=_f32)
        check(lib.ge_plane_mean(_p(x), _p(y), B * C, H * W, _stream()), "plane_mean")
        ctx.shape = x.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        B, C, H, W = ctx.shape
        return (dy / float(H * W)).expand(B, C, H, W).contiguous()


def adaptive_avg_pool2d_1(x):
    return _PlaneMeanFn.apply(x)


class _ActFn(Function):
    @staticmethod
    def forward(ctx, x, mode, slope=0.0):
        x = _c(x)
        y = torch.empty_like(x)
        check(lib.ge_act_fwd(_p(x), _p(y), x.numel(), mode, float(slope), _stream()), "act_fwd")
        ctx.save_for_backward(y if mode == 0 else x)
        ctx.mode, ctx.slope = mode, float(slope)
        return y

    @staticmethod
    def backward(ctx, dy):
        (ref,) = ctx.saved_tensors
        dy = _c(dy)
        dx = torch.empty_like(dy)
        check(lib.ge_act_bwd(_p(dy), _p(ref), _p(dx), dy.numel(), ctx.mode, ctx.slope, _stream()), "act_bwd")
        return dx, None, None


def relu(x):
    return _ActFn.apply(x, 0) if x.numel() else x


def gelu(x):
    return _ActFn.apply(x, 1) if x.numel() else x


def leaky_relu(x, negative_slope=0.2):
    return _ActFn.apply(x, 2, negative_slope) if x.numel() else x


def hardswish(x):
    return _ActFn.apply(x, 3) if x.numel() else x


class _PReLUFn(Function):
    @staticmethod
    def forward(ctx, x, slope):
        x = _c(x)
        if slope.numel() != 1:
            raise NotImplementedError("prelu: one shared slope (num_parameters=1) only -- what act_layer('prelu') builds")
        y = torch.empty_like(x)
        check(lib.ge_prelu_fwd(_p(x), _p(slope), _p(y), x.numel(), _stream()), "prelu_fwd")
        ctx.save_for_backward(x, slope)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, slope = ctx.saved_tensors
        dy = _c(dy)
        dx = torch.empty_like(x)
        part = torch.empty(lib.ge_prelu_num_partials(x.numel()), device=x.device, dtype=_f32)
        dslope = torch.empty_like(slope)
        check(lib.ge_prelu_bwd(_p(dy), _p(x), _p(slope), _p(dx), _p(part), _p(dslope), x.numel(), _stream()),
              "prelu_bwd")
        return dx, dslope


def prelu(x, slope):
    return _PReLUFn.apply(x, slope) if x.numel() else x


class _LastDimMaxFn(Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        K = x.shape[-1]
        rows = x.numel() // K
        y = torch.empty(x.shape[:-1] + (1,), device=x.device, dtype=_f32)
        arg = torch.empty(rows, device=x.device, dtype=torch.uint8)
        check(lib.ge_lastdim_max_fwd(_p(x), _p(y), _p(arg), rows, K, _stream()), "lastdim_max_fwd")
        ctx.save_for_backward(arg)
        ctx.shape = tuple(x.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        (arg,) = ctx.saved_tensors
        dx = torch.empty(ctx.shape, device=dy.device, dtype=_f32)
        check(lib.ge_lastdim_max_bwd(_p(_c(dy)), _p(arg), _p(dx), arg.numel(), ctx.shape[-1], _stream()),
              "lastdim_max_bwd")
        return dx


class _LastDimSumFn(Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        K = x.shape[-1]
        y = torch.empty(x.shape[:-1] + (1,), device=x.device, dtype=_f32)
        check(lib.ge_lastdim_sum_fwd(_p(x), _p(y), x.numel() // K, K, _stream()), "lastdim_sum_fwd")
        ctx.shape = tuple(x.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        dx = torch.empty(ctx.shape, device=dy.device, dtype=_f32)
        K = ctx.shape[-1]
        check(lib.ge_lastdim_sum_bwd(_p(_c(dy)), _p(dx), dx.numel() // K, K, _stream()), "lastdim_sum_bwd")
        return dx


def neighbour_max(x):
    """max over the last (neighbour) dimension, keepdim (EdgeConv2d / GraphSAGE, vig.py:122,136)."""
    return _LastDimMaxFn.apply(x)


def neighbour_sum(x):
    """sum over the last (neighbour) dimension, keepdim (GINConv2d, vig.py:157)."""
    return _LastDimSumFn.apply(x)


# --------------------------------------------------------------------------------------------------
# Grapher: k-NN graph + max-relative aggregation
# --------------------------------------------------------------------------------------------------
KNN_FUSED = os.environ.get("GE_KNN_FUSED", "1") != "0"      # query-side normalisation inside knn_topk_kernel (round 6); 0: two passes


@torch.no_grad()
def knn_graph(x, y=None, k=9, dilation=1, relative_pos=None, normalize=True):
    """edge_index int64 (2, B, N, k): DenseDilatedKnnGraph.forward of the reference (non-stochastic path).

    x: (B, C, N, 1) or (B, C, N); y: optional (B, C, M[, 1]); relative_pos: optional (1, N, M).
    """
    x3 = _c(x.detach()).reshape(x.shape[0], x.shape[1], -1)
    B, C, N = x3.shape
    st = _stream()
    K = k * dilation
    if y is not None and KNN_FUSED:
        # candidates prepared (M << N), the query side normalised inside the top-k kernel: no pass over x, no copy of it
        y3 = _c(y.detach()).reshape(y.shape[0], y.shape[1], -1)
        M = y3.shape[2]
        yn = torch.empty_like(y3)
        sqy = torch.empty((B, M), device=x3.device, dtype=_f32)
        check(lib.ge_knn_prepare(_p(y3), _p(yn), _p(sqy), B, C, M, int(normalize), st), "knn_prepare")
        rp = _c(relative_pos).reshape(N, M) if relative_pos is not None else None
        edge = torch.empty((2, B, N, (K + dilation - 1) // dilation), device=x3.device, dtype=torch.int64)
        xn = torch.empty_like(x3)         # written and read back by the kernel (its workgroups' own rows)
        check(lib.ge_knn_topk_fused(_p(x3), _p(xn), _p(yn), _p(sqy), _p(rp), _p(edge), B, C, N, M, K, dilation, int(normalize), st),
              "knn_topk_fused")
        edge._ge_centre_is_self = True
        return edge
    xn = torch.empty_like(x3)
    sqx = torch.empty((B, N), device=x3.device, dtype=_f32)
    check(lib.ge_knn_prepare(_p(x3), _p(xn), _p(sqx), B, C, N, int(normalize), st), "knn_prepare")
    if y is not None:
        y3 = _c(y.detach()).reshape(y.shape[0], y.shape[1], -1)
        M = y3.shape[2]
        yn = torch.empty_like(y3)
        sqy = torch.empty((B, M), device=x3.device, dtype=_f32)
        check(lib.ge_knn_prepare(_p(y3), _p(yn), _p(sqy), B, C, M, int(normalize), st), "knn_prepare")
    else:
        M, yn, sqy = N, xn, sqx
    rp = None
    if relative_pos is not None:
        rp = _c(relative_pos).reshape(N, M)
    edge = torch.empty((2, B, N, (K + dilation - 1) // dilation), device=x3.device, dtype=torch.int64)
    check(lib.ge_knn_topk(_p(xn), _p(sqx), _p(yn), _p(sqy), _p(rp), _p(edge), B, C, N, M, K, dilation, st), "knn_topk")
    edge._ge_centre_is_self = True   # edge[1][b][n][k] == n by construction: mr_aggregate may use the tiled kernels
    return edge


# Backward of the max-relative aggregation on k-NN graphs as a deterministic gather over inverse neighbour lists
# (ge_graph.hip: mr_inv_build_kernel, mr_bwd_gather_kernel); GE_MR_BWD_DET=0: the LDS-atomic scatter of rounds 1-3
MR_BWD_DETERMINISTIC = os.environ.get("GE_MR_BWD_DET", "1") != "0"


class _MRGatherFn(Function):
    @staticmethod
    def forward(ctx, x, y, edge, self_centred):
        x3 = _c(x).reshape(x.shape[0], x.shape[1], -1)
        B, C, N = x3.shape
        y3 = x3 if y is None else _c(y).reshape(y.shape[0], y.shape[1], -1)
        M = y3.shape[2]
        edge = edge.contiguous()
        K = edge.shape[3]
        out = torch.empty((B, 2 * C, N, 1), device=x3.device, dtype=_f32)
        argk = torch.empty((B, C, N), device=x3.device, dtype=torch.uint8)
        check(lib.ge_mrconv_gather_fwd(_p(x3), _p(y3), _p(edge), _p(out), _p(argk), B, C, N, M, K, int(self_centred),
                                       _stream()), "mrconv_gather_fwd")
        ctx.save_for_backward(edge, argk)
        ctx.cfg = (B, C, N, M, K, y is not None, x.shape, None if y is None else y.shape, int(self_centred))
        return out

    @staticmethod
    def backward(ctx, dout):
        edge, argk = ctx.saved_tensors
        B, C, N, M, K, has_y, xshape, yshape, self_centred = ctx.cfg
        dout = _c(dout)
        dx = torch.empty((B, C, N), device=dout.device, dtype=_f32)
        dy = torch.empty((B, C, M), device=dout.device, dtype=_f32) if has_y else dx
        if MR_BWD_DETERMINISTIC and lib.ge_mrconv_gather_bwd_small_ok(N, M, K, self_centred):
            # small graphs (<= 512 nodes): list inverted per workgroup in LDS, one launch
            check(lib.ge_mrconv_gather_bwd_small(_p(dout), _p(edge), _p(argk), _p(dx), _p(dy), B, C, N, M, K, _stream()),
                  "mrconv_gather_bwd_small")
            return dx.reshape(xshape), (dy.reshape(yshape) if has_y else None), None, None
        if MR_BWD_DETERMINISTIC and lib.ge_mrconv_gather_bwd_det_ok(N, M, K, self_centred):
            # inverse neighbour lists of this call's graph (fixed order), then a gather per candidate: same bits every run
            J = lib.ge_mr_inv_chunk()
            inv = torch.empty((B, N * K), device=dout.device, dtype=torch.int32)
            off = torch.empty((B, -(-N // J), M + 1), device=dout.device, dtype=torch.int32)
            check(lib.ge_mr_inv_build(_p(edge), _p(inv), _p(off), B, N, M, K, _stream()), "mr_inv_build")
            ws_n = lib.ge_mrconv_gather_bwd_det_workspace(B, C, N, M, K, int(not has_y))
            ws = torch.empty(ws_n, device=dout.device, dtype=_f32) if ws_n else None
            check(lib.ge_mrconv_gather_bwd_det(_p(dout), _p(inv), _p(off), _p(argk), _p(dx), _p(dy), _p(ws), B, C, N, M, K,
                                               _stream()), "mrconv_gather_bwd_det")
            return dx.reshape(xshape), (dy.reshape(yshape) if has_y else None), None, None
        ws_n = lib.ge_mrconv_gather_bwd_workspace(B, C, N, M, K, self_centred)
        ws = torch.empty(ws_n, device=dout.device, dtype=_f32) if ws_n else None
        check(lib.ge_mrconv_gather_bwd(_p(dout), _p(edge), _p(argk), _p(dx), _p(dy), _p(ws), B, C, N, M, K, self_centred,
                                       _stream()), "mrconv_gather_bwd")
        return dx.reshape(xshape), (dy.reshape(yshape) if has_y else None), None, None


class _EdgeGatherFn(Function):
    @staticmethod
    def forward(ctx, src, idx):
        s3 = _c(src).reshape(src.shape[0], src.shape[1], -1)
        B, C, M = s3.shape
        idx = idx.contiguous()
        if idx.dtype != torch.int64 or idx.shape[0] != B:
            raise RuntimeError("edge_gather: idx must be int64 (B, N, k)")
        N, K = idx.shape[1], idx.shape[2]
        out = torch.empty((B, C, N, K), device=s3.device, dtype=_f32)
        check(lib.ge_edge_gather_fwd(_p(s3), _p(idx), _p(out), B, C, M, N * K, _stream()), "edge_gather_fwd")
        ctx.save_for_backward(idx)
        ctx.cfg = (B, C, M, N * K, src.shape)
        return out

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        B, C, M, E, shape = ctx.cfg
        dsrc = torch.empty((B, C, M), device=dout.device, dtype=_f32)
        check(lib.ge_edge_gather_bwd(_p(_c(dout)), _p(idx), _p(dsrc), B, C, M, E, _stream()), "edge_gather_bwd")
        return dsrc.reshape(shape), None


def edge_gather(src, idx):
    """batched_index_select of the reference (vig.py:209-229): src (B, C, M[, 1]), idx (B, N, k) int64 ->
    (B, C, N, k) neighbour features."""
    return _EdgeGatherFn.apply(src, idx)


def mr_aggregate(x, edge_index, y=None):
    """MRConv2d's gather + max-relative + channel-interleaved concat: (B,C,N,1) -> (B,2C,N,1).

    Graphs built by knn_graph carry `_ge_centre_is_self` (edge_index[1][b][n][k] == n), which selects the LDS-tiled
    kernels; any other edge_index takes the general gather."""
    return _MRGatherFn.apply(x, y, edge_index, bool(getattr(edge_index, "_ge_centre_is_self", False)))


# --------------------------------------------------------------------------------------------------
# Sinkhorn
# --------------------------------------------------------------------------------------------------
SD_FUSED = True       # one-launch forward (sd_fused_kernel) when the problem fits; False: cost / iterate / finalize launches
_SD_SYNC = {}         # (device, stream) -> the two-int meeting point of the fused kernel's workgroups (zeroed per launch)


class _SinkhornDistanceFn(Function):
    @staticmethod
    def forward(ctx, x, y, eps, max_iter, thresh):
        x, y = _c(x), _c(y)
        B, P1, D = x.shape
        P2 = y.shape[1]
        dev = x.device
        Cm = torch.empty((B, P1, P2), device=dev, dtype=_f32)
        pi = torch.empty((B, P1, P2), device=dev, dtype=_f32)
        cost = torch.empty(B, device=dev, dtype=_f32)
        nits = torch.empty(1, device=dev, dtype=torch.int32)
        uh = torch.empty((B, max_iter + 1, P1), device=dev, dtype=_f32)
        vh = torch.empty((B, max_iter + 1, P2), device=dev, dtype=_f32)
        err = torch.empty((B, max_iter), device=dev, dtype=_f32)
        if SD_FUSED and lib.ge_sinkhorn_distance_fused_ok(B, P1, P2):
            key = (dev, _stream())       # one meeting point per (device, stream): launches in flight never share one
            sync = _SD_SYNC.get(key)
            if sync is None:
                sync = _SD_SYNC[key] = torch.zeros(2, device=dev, dtype=torch.int32)
            check(lib.ge_sinkhorn_distance_fwd_fused(_p(x), _p(y), _p(Cm), _p(pi), _p(cost), _p(nits), _p(uh), _p(vh),
                                                     _p(err), _p(sync), B, P1, P2, D, eps, max_iter, thresh, _stream()),
                  "sinkhorn_distance_fwd_fused")
        else:
            check(lib.ge_sinkhorn_distance_fwd(_p(x), _p(y), _p(Cm), _p(pi), _p(cost), _p(nits), _p(uh), _p(vh), _p(err),
                                               B, P1, P2, D, eps, max_iter, thresh, _stream()), "sinkhorn_distance_fwd")
        ctx.save_for_backward(x, y, Cm, uh, vh, nits)
        ctx.cfg = (eps, max_iter)
        ctx.mark_non_differentiable(nits)
        return cost, pi, Cm, nits

    @staticmethod
    def backward(ctx, g_cost, g_pi, g_C, _g_nits):
        x, y, Cm, uh, vh, nits = ctx.saved_tensors
        eps, max_iter = ctx.cfg
        B, P1, D = x.shape
        P2 = y.shape[1]
        g_cost = _c(g_cost) if g_cost is not None else None
        g_pi = _c(g_pi) if g_pi is not None else None
        g_C = _c(g_C) if g_C is not None else None
        dC = torch.empty_like(Cm)
        dx = torch.empty_like(x)
        dy = torch.empty_like(y)
        check(lib.ge_sinkhorn_distance_bwd(_p(x), _p(y), _p(Cm), _p(uh), _p(vh), _p(nits), _p(g_cost), _p(g_pi),
                                           _p(g_C), _p(dC), _p(dx), _p(dy), B, P1, P2, D, eps, max_iter, _stream()),
              "sinkhorn_distance_bwd")
        return dx, dy, None, None, None


def sinkhorn_distance(x, y, eps, max_iter, thresh=0.1):
    """(cost[B], pi[B,P1,P2], C[B,P1,P2], nits[1]) for x (B,P1,D), y (B,P2,D)."""
    return _SinkhornDistanceFn.apply(x, y, float(eps), int(max_iter), float(thresh))


_RPM_WS = {}          # (device, stream) -> exchange buffer of the co-operative sinkhorn_rpm kernels


class _SinkhornRPMFn(Function):
    @staticmethod
    def forward(ctx, log_alpha, n_iters):
        A = _c(log_alpha)
        B, N1, N2 = A.shape
        dev = A.device
        X = torch.empty_like(A)
        rho = torch.empty((n_iters, B, N1), device=dev, dtype=_f32)
        gam = torch.empty((n_iters + 1, B, N2), device=dev, dtype=_f32)
        n_ws = 0 if torch.cuda.is_current_stream_capturing() else lib.ge_sinkhorn_rpm_coop_workspace(B, N1, N2)
        ctx.coop = n_ws > 0
        if n_ws > 0:
            # the training step's sizes: one launch of 16 co-operating workgroups instead of the 41-launch chain; their exchange
            # buffer is per (device, stream) -- launches in flight never share one
            key = (dev, _stream())
            ws = _RPM_WS.get(key)
            if ws is None or ws.numel() < n_ws:
                ws = _RPM_WS[key] = torch.empty(max(n_ws, 4 + 4 * 16 * 512), device=dev, dtype=_f32)
            check(lib.ge_sinkhorn_rpm_fwd_coop(_p(A), _p(X), _p(rho), _p(gam), _p(ws), N1, N2, n_iters, _stream()),
                  "sinkhorn_rpm_fwd_coop")
        else:
            check(lib.ge_sinkhorn_rpm_fwd(_p(A), _p(X), _p(rho), _p(gam), B, N1, N2, n_iters, _stream()),
                  "sinkhorn_rpm_fwd")
        ctx.save_for_backward(A, rho, gam)
        ctx.n_iters = n_iters
        return X

    @staticmethod
    def backward(ctx, gX):
        A, rho, gam = ctx.saved_tensors
        B, N1, N2 = A.shape
        gX = _c(gX)
        gA = torch.empty_like(A)
        g_rho = torch.empty((B, N1), device=A.device, dtype=_f32)
        g_gam = torch.empty((B, N2), device=A.device, dtype=_f32)
        n_ws = lib.ge_sinkhorn_rpm_coop_workspace(B, N1, N2) if ctx.coop and not torch.cuda.is_current_stream_capturing() else 0
        if n_ws > 0:
            key = (A.device, _stream())
            ws = _RPM_WS.get(key)
            if ws is None or ws.numel() < n_ws:
                ws = _RPM_WS[key] = torch.empty(max(n_ws, 4 + 4 * 16 * 512), device=A.device, dtype=_f32)
            check(lib.ge_sinkhorn_rpm_bwd_coop(_p(A), _p(gX), _p(rho), _p(gam), _p(gA), _p(ws), N1, N2, ctx.n_iters, _stream()),
                  "sinkhorn_rpm_bwd_coop")
            return gA, None
        check(lib.ge_sinkhorn_rpm_bwd(_p(A), _p(gX), _p(rho), _p(gam), _p(gA), _p(g_rho), _p(g_gam), B, N1, N2,
                                      ctx.n_iters, _stream()), "sinkhorn_rpm_bwd")
        return gA, None


def sinkhorn_rpm(log_alpha, n_iters=5):
    """GModule.sinkhorn_rpm(log_alpha, n_iters, slack=True): (B, N1, N2) -> log of the near-doubly-stochastic plan."""
    return _SinkhornRPMFn.apply(log_alpha, int(n_iters))


class _MatchO2OFn(Function):
    """(loss, M) of GModule._forward_aff's one-to-one branch from the log plan X (N1, N2) and both label vectors."""

    @staticmethod
    def forward(ctx, X, lab1, lab2):
        X, lab1, lab2 = _c(X), _c(lab1.float()), _c(lab2.float())
        N1, N2 = X.shape
        dev = X.device
        M = torch.empty_like(X)
        idx = torch.empty(N1, device=dev, dtype=torch.int32)
        rowpart = torch.empty((N1, 4), device=dev, dtype=_f32)
        out = torch.empty(4, device=dev, dtype=_f32)      # loss, then the three scalars of the backward
        check(lib.ge_match_o2o_fwd(_p(X), _p(lab1), _p(lab2), _p(M), _p(idx), _p(rowpart), _p(out), _p(out) + 4, N1, N2, _stream()),
              "match_o2o_fwd")
        ctx.save_for_backward(M, lab1, lab2, idx, out)
        ctx.set_materialize_grads(False)
        return out[0], M

    @staticmethod
    def backward(ctx, g_loss, g_M):
        M, lab1, lab2, idx, out = ctx.saved_tensors
        N1, N2 = M.shape
        gX = torch.empty_like(M)
        gl = None if g_loss is None else _c(g_loss.reshape(1).to(_f32))
        gm = None if g_M is None else _c(g_M)
        check(lib.ge_match_o2o_bwd(_p(M), _p(lab1), _p(lab2), _p(idx), _p(out) + 4, _p(gl), _p(gm), _p(gX), N1, N2, _stream()),
              "match_o2o_bwd")
        return gX, None, None


class _MHA1Fn(Function):
    """Single-head MultiHeadAttention block (transformer.py:28-78, "v2") in one call per direction (csrc/ge_attention.hip)."""

    @staticmethod
    def forward(ctx, key, value, query, Wk, bk, Wv, bv, Wq, bq, Wf, bf, gamma, beta, mask_att, mask_out, scale, att_scale, out_scale, eps):
        ctx.params = (Wk, bk, Wv, bv, Wq, bq, Wf, bf)
        key, value, query = _c(key), _c(value), _c(query)
        Wk, Wv, Wq, Wf = _c(Wk), _c(Wv), _c(Wq), _c(Wf)
        Nk, D = key.shape
        Nq = query.shape[0]
        dev = key.device
        # one buffer for everything the backward needs: k, v [Nk][D]; q, ctx, z [Nq][D]; P [Nq][Nk]; mean, invstd [Nq]
        sizes = (Nk * D, Nk * D, Nq * D, Nq * D, Nq * D, Nq * Nk, Nq, Nq)
        save = torch.empty(sum(sizes), device=dev, dtype=_f32)
        offs, o = [], _p(save)
        for n in sizes:
            offs.append(o)
            o += 4 * n
        k_, v_, q_, c_, z_, P_, mu_, is_ = offs
        out = torch.empty((Nq, D), device=dev, dtype=_f32)
        A = torch.empty((Nq, Nk), device=dev, dtype=_f32)
        check(lib.ge_mha1_fwd(_p(key), _p(value), _p(query), _p(Wk), _p(bk), _p(Wv), _p(bv), _p(Wq), _p(bq), _p(Wf), _p(bf),
                              _p(gamma), _p(beta), _p(mask_att), _p(mask_out), k_, v_, q_, P_, _p(A), c_, z_, mu_, is_, _p(out),
                              Nk, Nq, D, scale, att_scale, out_scale, eps, _stream()), "mha1_fwd")
        ctx.save_for_backward(key, value, query, Wk, Wv, Wq, Wf, gamma, mask_att, mask_out, save, A)
        ctx.cfg = (sizes, scale, att_scale, out_scale)
        ctx.set_materialize_grads(False)
        return out, A

    @staticmethod
    def backward(ctx, d_out, d_att):
        key, value, query, Wk, Wv, Wq, Wf, gamma, mask_att, mask_out, save, A = ctx.saved_tensors
        sizes, scale, att_scale, out_scale = ctx.cfg
        Nk, D = key.shape
        Nq = query.shape[0]
        dev = key.device
        offs, o = [], _p(save)
        for n in sizes:
            offs.append(o)
            o += 4 * n
        k_, v_, q_, c_, z_, P_, mu_, is_ = offs
        d_out = torch.zeros((Nq, D), device=dev, dtype=_f32) if d_out is None else _c(d_out)
        d_att = None if d_att is None else _c(d_att)
        pk, pbk, pv, pbv, pq, pbq, pf, pbf = ctx.params
        wts, bss = (pk, pv, pq, pf), (pbk, pbv, pbq, pbf)
        # positions of (Wk, bk, Wv, bv, Wq, bq, Wf, bf) in forward's arguments: 3 .. 10.  Straight accumulation into .grad only
        # when EVERY weight (bias) wants a gradient -- a frozen parameter must not be touched
        need = ctx.needs_input_grad
        w_need, b_need = [need[i] for i in (3, 5, 7, 9)], [need[i] for i in (4, 6, 8, 10)]
        w_acc = all(w_need) and all(_direct(w) and w.grad.is_contiguous() for w in wts)
        has_b = all(b is not None for b in bss)
        if not has_b and any(b is not None for b in bss):
            raise RuntimeError("mha1: the four projections must all have a bias or none (MultiHeadAttention gates on this)")
        b_acc = has_b and all(b_need) and all(_direct(b) for b in bss)
        dW = [w.grad if w_acc else torch.empty_like(w) for w in wts]
        dB = [(b.grad if b_acc else torch.empty_like(b)) if has_b else None for b in bss]
        dgamma = torch.empty(D, device=dev, dtype=_f32) if gamma is not None else None
        dbeta = torch.empty(D, device=dev, dtype=_f32) if gamma is not None else None
        dkey, dvalue, dquery = torch.empty_like(key), torch.empty_like(value), torch.empty_like(query)
        ws = torch.empty(lib.ge_mha1_bwd_workspace(Nk, Nq, D), device=dev, dtype=_f32)
        check(lib.ge_mha1_bwd(_p(key), _p(value), _p(query), _p(Wk), _p(Wv), _p(Wq), _p(Wf), _p(gamma), _p(mask_att), _p(mask_out),
                              k_, v_, q_, P_, _p(A), c_, z_, mu_, is_, _p(d_out), _p(d_att), _p(dkey), _p(dvalue), _p(dquery),
                              _p(dW[0]), _p(dB[0]), _p(dW[1]), _p(dB[1]), _p(dW[2]), _p(dB[2]), _p(dW[3]), _p(dB[3]), int(w_acc),
                              int(b_acc), _p(dgamma), _p(dbeta), _p(ws), Nk, Nq, D, scale, att_scale, out_scale, _stream()), "mha1_bwd")
        gw = [None if (w_acc or not n) else g for g, n in zip(dW, w_need)]
        gb = [None if (b_acc or not has_b or not n) else g for g, n in zip(dB, b_need)]
        return (dkey, dvalue, dquery, gw[0], gb[0], gw[1], gb[1], gw[2], gb[2], gw[3], gb[3], dgamma if need[11] else None,
                dbeta if need[12] else None, None, None, None, None, None, None)


def mha1(key, value, query, Wk, bk, Wv, bv, Wq, bq, Wf, bf, gamma, beta, mask_att, mask_out, scale, att_scale, out_scale=None,
         eps=1e-5):
    """(LayerNorm(query + dropout(final(softmax(q k^T scale) (.) mask v))), post-dropout attention) for one head; masks: 0 / 1
    keep tensors or None, att_scale / out_scale = 1 / keep probability of the two dropout sites (out_scale None: = att_scale)."""
    return _MHA1Fn.apply(key, value, query, Wk, bk, Wv, bv, Wq, bq, Wf, bf, gamma, beta, mask_att, mask_out, float(scale),
                         float(att_scale), float(att_scale if out_scale is None else out_scale), float(eps))


def seed_bank_update(bank, nodes, table, num_classes):
    """In-place momentum update of a (num_classes, D) seed bank from nodes (N, D); table: int32 [N class ids (-1: dropped)] +
    [num_classes presence flags] on the device (GModule.update_seed, graph_matching.py:532-567)."""
    nodes = _c(nodes)
    N, D = nodes.shape
    if not bank.is_contiguous():
        raise RuntimeError("seed_bank_update: the bank must be contiguous (it is updated in place)")
    check(lib.ge_seed_bank_update(_p(bank), _p(nodes), _p(table), _p(table) + 4 * N, int(num_classes), N, D, _stream()),
          "seed_bank_update")
    # the kernel wrote through a raw pointer: bump the version counter like the bank.copy_() it replaces did, so that a tensor
    # saved for backward that aliases a bank row raises instead of silently reading the updated values
    torch.autograd.graph.increment_version(bank)
    return bank


def match_o2o_loss(log_plan, labels_1, labels_2):
    """tp_loss + fp_loss and M = exp(log_plan) of graph_matching.py:577-590 (three launches forward + backward instead of ~40)."""
    return _MatchO2OFn.apply(log_plan, labels_1, labels_2)


# --------------------------------------------------------------------------------------------------
# Affinity MLP, softmax
# --------------------------------------------------------------------------------------------------
class _AffinityFn(Function):
    @staticmethod
    def forward(ctx, P, Q, b1, w2, b2):
        P, Q, b1, w2, b2 = _c(P), _c(Q), _c(b1), _c(w2).reshape(-1), _c(b2)
        N1, H = P.shape
        N2 = Q.shape[0]
        M = torch.empty((N1, N2), device=P.device, dtype=_f32)
        check(lib.ge_affinity_fwd(_p(P), _p(Q), _p(b1), _p(w2), _p(b2), _p(M), N1, N2, H, _stream()), "affinity_fwd")
        ctx.save_for_backward(P, Q, b1, w2)
        return M

    @staticmethod
    def backward(ctx, dM):
        P, Q, b1, w2 = ctx.saved_tensors
        dM = _c(dM)
        N1, H = P.shape
        N2 = Q.shape[0]
        dev = P.device
        dP = torch.empty_like(P)
        dQ = torch.empty_like(Q)
        nblk = (N1 + 1) // 2
        part = torch.empty((nblk, H), device=dev, dtype=_f32)
        st = _stream()
        check(lib.ge_affinity_bwd(_p(P), _p(Q), _p(b1), _p(w2), _p(dM), _p(dP), _p(dQ), _p(part), N1, N2, H, st),
              "affinity_bwd")
        dw2 = torch.empty(H, device=dev, dtype=_f32)
        db1 = torch.empty(H, device=dev, dtype=_f32)
        check(lib.ge_colsum(_p(part), _p(dw2), nblk, H, st), "colsum")
        check(lib.ge_colsum(_p(dP), _p(db1), N1, H, st), "colsum")
        db2 = dM.sum().reshape(1)
        return dP, dQ, db1, dw2.reshape(1, H), db2


def affinity_mlp(P, Q, b1, w2, b2):
    """M[i,j] = b2 + w2 . relu(P[i] + Q[j] + b1);  w2: (1, H), b2: (1,)."""
    return _AffinityFn.apply(P, Q, b1, w2, b2)


class _SoftmaxFn(Function):
    @staticmethod
    def forward(ctx, x, scale):
        x = _c(x)
        D = x.shape[-1]
        y = torch.empty_like(x)
        check(lib.ge_softmax_fwd(_p(x), _p(y), x.numel() // D, D, scale, _stream()), "softmax_fwd")
        ctx.save_for_backward(y)
        ctx.scale = scale
        return y

    @staticmethod
    def backward(ctx, dy):
        (p,) = ctx.saved_tensors
        dy = _c(dy)
        D = p.shape[-1]
        dx = torch.empty_like(p)
        check(lib.ge_softmax_bwd(_p(dy), _p(p), _p(dx), p.numel() // D, D, ctx.scale, _stream()), "softmax_bwd")
        return dx, None


def softmax_lastdim(x, scale=1.0):
    return _SoftmaxFn.apply(x, float(scale))


# --------------------------------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------------------------------
class _BCEFn(Function):
    @staticmethod
    def forward(ctx, x, target, tconst):
        x = _c(x)
        t = _c(target) if target is not None else None
        partial = torch.empty(1024, device=x.device, dtype=_f32)
        loss = torch.empty(1, device=x.device, dtype=_f32)
        check(lib.ge_bce_logits_fwd(_p(x), _p(t), tconst, _p(partial), _p(loss), x.numel(), _stream()), "bce_fwd")
        ctx.save_for_backward(x, t)
        ctx.tconst = tconst
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        x, t = ctx.saved_tensors
        g = _c(g).reshape(1)
        dx = torch.empty_like(x)
        check(lib.ge_bce_logits_bwd(_p(x), _p(t), ctx.tconst, _p(g), _p(dx), x.numel(), _stream()), "bce_bwd")
        return dx, None, None


def bce_with_logits(x, target):
    """nn.BCEWithLogitsLoss(reduction='mean'); `target` is a tensor or a python float (constant target)."""
    if isinstance(target, (int, float)):
        return _BCEFn.apply(x, None, float(target))
    return _BCEFn.apply(x, target, 0.0)


class _DiceSumsFn(Function):
    """softmax over C + per-(b,c) sums (sum p*t, sum p^2, sum t^2)."""

    @staticmethod
    def forward(ctx, x, t):
        x, t = _c(x), _c(t)
        B, C = x.shape[0], x.shape[1]
        HW = x.numel() // (B * C)
        nb = lib.ge_dice_num_partials(HW)
        prob = torch.empty_like(x)
        partial = torch.empty(B * C * nb * 3, device=x.device, dtype=_f32)
        sums = torch.empty((B, C, 3), device=x.device, dtype=_f32)
        check(lib.ge_dice_fwd(_p(x), _p(t), _p(prob), _p(partial), _p(sums), B, C, HW, _stream()), "dice_fwd")
        ctx.save_for_backward(prob, t)
        return sums

    @staticmethod
    def backward(ctx, gs):
        prob, t = ctx.saved_tensors
        B, C = prob.shape[0], prob.shape[1]
        HW = prob.numel() // (B * C)
        gs = _c(gs)
        ca = gs[:, :, 0].contiguous()
        cb = gs[:, :, 1].contiguous()
        dx = torch.empty_like(prob)
        check(lib.ge_dice_bwd(_p(prob), _p(t), _p(ca), _p(cb), _p(dx), B, C, HW, _stream()), "dice_bwd")
        return dx, None


def dice_loss(predict, target, smooth=1.0):
    """DiceLoss()(predict, target) of utils/losses.py (p=2, mean over batch, mean over channels)."""
    sums = _DiceSumsFn.apply(predict, target)
    num = sums[..., 0] + smooth
    den = sums[..., 1] + sums[..., 2] + smooth
    return (1.0 - num / den).mean(0).sum() / predict.shape[1]


# --------------------------------------------------------------------------------------------------
# optimizers on flat buffers
# --------------------------------------------------------------------------------------------------
class _MeanSquareFn(Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        n = x.numel()
        part = torch.empty(lib.ge_mean_square_blocks(n), device=x.device, dtype=_f32)
        out = torch.empty((), device=x.device, dtype=_f32)
        check(lib.ge_mean_square_fwd(_p(x), _p(part), _p(out), n, _stream()), "mean_square_fwd")
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        dx = torch.empty_like(x)
        check(lib.ge_mean_square_bwd(_p(x), _p(_c(g)), _p(dx), x.numel(), _stream()), "mean_square_bwd")
        return dx


def mean_square(x):
    """mean(x * x) over the whole tensor, one read forward and one read + one write backward."""
    return _MeanSquareFn.apply(x)


# --------------------------------------------------------------------------------------------------
# front end of GModule's graph construction: boxes, location labels, sampled rows
# --------------------------------------------------------------------------------------------------
def mask_boxes(masks):
    """(N, H, W) -> (N, 4) float (x1, y1, x2, y2) of the non-zero pixels, (0, 0, W, H) for an empty mask
    (graph_matching.py:702-740)."""
    m = _c(masks)
    N, H, W = m.shape
    boxes = torch.empty((N, 4), device=m.device, dtype=_f32)
    check(lib.ge_mask_boxes(_p(m), _p(boxes), N, H, W, _stream()), "mask_boxes")
    return boxes


def fcos_labels(boxes, levels, ranges):
    """boxes (B, nc, 4); levels: [(h, w, stride)]; ranges: [(lo, hi)] -> uint8 labels (B, sum h*w), levels concatenated
    (graph_matching.py:874-959)."""
    import ctypes

    b = _c(boxes)
    B, nc = b.shape[0], b.shape[1]
    n = len(levels)
    hws = (ctypes.c_int * (3 * n))(*[int(v) for lv in levels for v in lv])
    rng = (ctypes.c_float * (2 * n))(*[float(v) for r in ranges for v in r])
    total = sum(h * w for h, w, _ in levels)
    out = torch.empty((B, total), device=b.device, dtype=torch.uint8)
    check(lib.ge_fcos_labels(_p(b), _p(out), B, nc, n, hws, rng, _stream()), "fcos_labels")
    return out


class _GatherNodesFn(Function):
    @staticmethod
    def forward(ctx, level, index, unique, present, *feats):
        feats = [_c(f) for f in feats]
        if len(feats) > 5:
            raise RuntimeError("gather_nodes: at most five pyramid levels")
        C = feats[0].shape[1]
        n = level.numel()
        out = torch.empty((n, C), device=feats[0].device, dtype=_f32)
        ptr = [_p(f) for f in feats] + [None] * (5 - len(feats))
        hw = [f.shape[2] * f.shape[3] for f in feats] + [0] * (5 - len(feats))
        check(lib.ge_gather_nodes_fwd(*ptr, *hw, C, _p(level), _p(index), _p(out), n, _stream()), "gather_nodes_fwd")
        ctx.save_for_backward(level, index)
        ctx.meta = ([f.shape for f in feats], hw, C, bool(unique), present)
        return out

    @staticmethod
    def backward(ctx, dout):
        level, index = ctx.saved_tensors
        shapes, hw, C, unique, present = ctx.meta
        dout = _c(dout)
        grads = [torch.zeros(sh, device=dout.device, dtype=_f32)
                 if ctx.needs_input_grad[4 + i] and (present is None or present[i]) else None
                 for i, sh in enumerate(shapes)]
        ptr = [_p(g) for g in grads] + [None] * (5 - len(grads))
        check(lib.ge_gather_nodes_bwd(_p(dout), _p(level), _p(index), *ptr, *hw, C, level.numel(), int(not unique),
                                      _stream()), "gather_nodes_bwd")
        return (None, None, None, None) + tuple(grads)


def gather_nodes(feats, level, index, unique=True, present=None):
    """Rows (n, C) of NCHW pyramid levels: row i = feats[level[i]][b, :, y, x] with index[i] = (b*H + y)*W + x
    (graph_matching.py:961-1013).  level / index: int64 device tensors; unique=False if a location may repeat;
    present: per level, whether any row comes from it (a level without rows gets no gradient tensor)."""
    return _GatherNodesFn.apply(level, index, unique, None if present is None else tuple(present), *feats)


def adam_step_(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    check(lib.ge_adam_step(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, weight_decay, step,
                           grad_scale, _stream()), "adam_step")
    bump_param_epoch()


def adam_step_masked_(p, g, m, v, i0, seg_end, used, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    """Adam on p[0:n] = flat[i0:i0+n] for the parameters whose device flag `used[s]` is set (seg_end: their end offsets)."""
    check(lib.ge_adam_step_masked(_p(p), _p(g), _p(m), _p(v), p.numel(), i0, _p(seg_end), _p(used), seg_end.numel(), lr,
                                  beta1, beta2, eps, weight_decay, step, grad_scale, _stream()), "adam_step_masked")
    bump_param_epoch()


def sgd_step_masked_(p, g, buf, i0, seg_end, used, started, lr, momentum, weight_decay, grad_scale=1.0):
    check(lib.ge_sgd_step_masked(_p(p), _p(g), _p(buf), p.numel(), i0, _p(seg_end), _p(used), _p(started), seg_end.numel(),
                                 lr, momentum, weight_decay, grad_scale, _stream()), "sgd_step_masked")
    bump_param_epoch()


def flags_max_(a, b):
    check(lib.ge_flags_max(_p(a), _p(b), a.numel(), _stream()), "flags_max")


def sgd_step_(p, g, buf, lr, momentum, weight_decay, first_step, grad_scale=1.0):
    check(lib.ge_sgd_step(_p(p), _p(g), _p(buf), p.numel(), lr, momentum, weight_decay, int(first_step), grad_scale,
                          _stream()), "sgd_step")
    bump_param_epoch()
