"""fp16 activation storage for conv3x3 -> BatchNorm -> ReLU (-> 2x2 max-pool) stacks (ge_half.hip).

BASELINE.json config 5 ("fp16 MFMA conv path + fp32 Sinkhorn") runs the reference's VGG16 backbone
(``/root/reference/models/fpnseg.py:18-166``, built by ``train_cardiac_uda.py:73``).  With
``functional.ACT_STORAGE = "f16"`` the stacks of that backbone keep every activation and every activation gradient in
HBM as fp16 in the channel-blocked layout ``h[b][c // 32][y][x][c % 32]`` (a torch.float16 tensor of shape
``(B, C // 32, H, W, 32)``); master weights, BatchNorm statistics / affine parameters, weight gradients and every
reduction stay fp32.  A tensor enters a stack through :func:`to_blocked` and leaves it through :func:`from_blocked`.

Loss scale: gradients inside a stack are multiplied by the loss scale where they enter (the backward of
``from_blocked``) and divided where they leave (the backward of ``to_blocked``, the weight / bias / affine gradient
kernels) -- per-pixel gradients of a mean loss over 48 x 256 x 256 pixels are ~1e-7, below fp16's normal range.  The scale
lives in device memory and follows the gradients from step to step (functional.h_scale / h_scale_update).
"""
import os

import torch
from torch.autograd import Function

from . import functional as GF
from ._lib import lib, check

_f16, _f32 = torch.float16, torch.float32
GRAD_SCALE = GF.H_GRAD_SCALE      # initial / fixed loss scale (functional.h_scale_value(device): the current one)

_p, _stream = GF._p, GF._stream


def is_blocked(t):
    return t.dtype == _f16 and t.dim() == 5 and t.shape[-1] == 32


def _hc(t):
    if not t.is_cuda or not is_blocked(t):
        raise RuntimeError("graphecho_amd.half: expected a blocked fp16 tensor (B, C/32, H, W, 32) on the HIP device")
    return t if t.is_contiguous() else t.contiguous()


def supported(B, Cin, Cout, H, W):
    return bool(lib.ge_h_conv3x3_supported(B, Cin, Cout, H, W))


class _ToBlockedFn(Function):
    @staticmethod
    def forward(ctx, x):
        x = GF._c(x)
        B, C, H, W = x.shape
        if C % 32:
            raise RuntimeError("to_blocked: channel count must be a multiple of 32")
        GF.h_scale_forward_update(x.device)
        h = torch.empty((B, C // 32, H, W, 32), device=x.device, dtype=_f16)
        check(lib.ge_h_from_f32(_p(x), _p(h), B, C, H * W, 1.0, None, _stream()), "h_from_f32")
        return h

    @staticmethod
    def backward(ctx, dh):
        dh = _hc(dh)
        B, CB, H, W, _ = dh.shape
        dx = torch.empty((B, CB * 32, H, W), device=dh.device, dtype=_f32)
        _, inv, hsp = GF.h_scale_args(dh.device)
        check(lib.ge_h_to_f32(_p(dh), _p(dx), B, CB * 32, H * W, inv, hsp, _stream()), "h_to_f32")
        return dx


class _FromBlockedFn(Function):
    @staticmethod
    def forward(ctx, h):
        h = _hc(h)
        B, CB, H, W, _ = h.shape
        x = torch.empty((B, CB * 32, H, W), device=h.device, dtype=_f32)
        check(lib.ge_h_to_f32(_p(h), _p(x), B, CB * 32, H * W, 1.0, None, _stream()), "h_to_f32")
        return x

    @staticmethod
    def backward(ctx, dx):
        dx = GF._c(dx)
        B, C, H, W = dx.shape
        dh = torch.empty((B, C // 32, H, W, 32), device=dx.device, dtype=_f16)
        S, _, hsp = GF.h_scale_args(dx.device, cast=True)
        check(lib.ge_h_from_f32(_p(dx), _p(dh), B, C, H * W, S, hsp, _stream()), "h_from_f32")
        return dh


class _ForkFn(Function):
    """A blocked fp16 tensor with TWO consumers, one inside the fp16 domain (the next stack) and one outside it (the FPN's
    lateral conv): returns (fp32 NCHW copy, alias of h).  The two gradients meet in ONE kernel that adds them in fp32 and
    rounds once, saturating -- left to autograd they would be added as fp16 tensors, and two large loss-scaled gradients
    overflow to inf there (seen on the first step of config 5's temporal workload, whose gradients reach 1e0)."""

    @staticmethod
    def forward(ctx, h):
        ctx.set_materialize_grads(False)
        h = _hc(h)
        B, CB, H, W, _ = h.shape
        x = torch.empty((B, CB * 32, H, W), device=h.device, dtype=_f32)
        check(lib.ge_h_to_f32(_p(h), _p(x), B, CB * 32, H * W, 1.0, None, _stream()), "h_to_f32")
        return x, h.view_as(h)

    @staticmethod
    def backward(ctx, dx, dh):
        if dx is None:
            return dh
        dx = GF._c(dx)
        B, C, H, W = dx.shape
        out = torch.empty((B, C // 32, H, W, 32), device=dx.device, dtype=_f16)
        S, _, hsp = GF.h_scale_args(dx.device, cast=True)
        if dh is None:
            check(lib.ge_h_from_f32(_p(dx), _p(out), B, C, H * W, S, hsp, _stream()), "h_from_f32")
        else:
            check(lib.ge_h_from_f32_add(_p(dx), _p(_hc(dh)), _p(out), B, C, H * W, S, hsp, _stream()), "h_from_f32_add")
        return out


def fork(h):
    """(fp32 NCHW copy of h, h) for a blocked tensor that continues inside the fp16 domain AND leaves it."""
    return _ForkFn.apply(h)


def to_blocked(x):
    """fp32 NCHW -> blocked fp16 (entry of a stack)."""
    return _ToBlockedFn.apply(x)


def from_blocked(h):
    """blocked fp16 -> fp32 NCHW (exit of a stack)."""
    return _FromBlockedFn.apply(h)


class _ConvHFn(Function):
    """3x3 / stride 1 / pad 1 convolution on blocked fp16 tensors; returns (z, stats)."""

    @staticmethod
    def forward(ctx, h, weight, bias, cache, want_stats):
        ctx.set_materialize_grads(False)
        h = _hc(h)
        weight = GF._c(weight)
        B, CB, H, W, _ = h.shape
        Cin = CB * 32
        Cout = weight.shape[0]
        if weight.shape[1] != Cin or tuple(weight.shape[2:]) != (3, 3):
            raise RuntimeError(f"half.conv3x3: weight {tuple(weight.shape)} does not match {Cin} input channels")
        wp = cache.get_lp(weight, 1, False, "f16") if cache is not None else GF._pack_weight_lp(weight, 1, False, "f16")
        z = torch.empty((B, Cout // 32, H, W, 32), device=h.device, dtype=_f16)
        stats = None
        if want_stats:
            stats = torch.empty((Cout, lib.ge_h_conv3x3_stat_parts(B, H, W), 3), device=h.device, dtype=_f32)
        kt = GF.KERNEL_TIMER
        t0 = kt.begin() if kt else None
        check(lib.ge_h_conv3x3_fwd(_p(h), _p(wp), _p(bias), _p(z), _p(stats), B, Cin, Cout, H, W, _stream()),
              "h_conv3x3_fwd")
        if kt:
            kt.end(t0, GF._conv_kind("convh_fwd", 3, 1, Cout, B * H * W, Cin * 9), 2.0 * B * H * W * Cout * Cin * 9,
                   2 * (h.numel() + weight.numel() + z.numel()))
        ctx.save_for_backward(h, weight)
        ctx.cache = cache
        ctx.params = (weight, bias)
        if want_stats:
            ctx.mark_non_differentiable(stats)
            return z, stats
        return z

    @staticmethod
    def backward(ctx, dz, *rest):
        h, weight = ctx.saved_tensors
        wparam, bparam = ctx.params
        cache = ctx.cache
        dz = _hc(dz)
        B, CB, H, W, _ = h.shape
        Cin, Cout = CB * 32, weight.shape[0]
        st = _stream()
        dh = dw = db = None
        kt = GF.KERNEL_TIMER
        flops = 2.0 * B * H * W * Cout * Cin * 9
        _, inv, hsp = GF.h_scale_args(h.device)
        if ctx.needs_input_grad[0]:
            wpt = cache.get_lp(weight, 1, True, "f16") if cache is not None else GF._pack_weight_lp(weight, 1, True, "f16")
            dh = torch.empty_like(h)
            t0 = kt.begin() if kt else None
            check(lib.ge_h_conv3x3_dgrad(_p(dz), _p(wpt), _p(dh), B, Cin, Cout, H, W, st), "h_conv3x3_dgrad")
            if kt:
                kt.end(t0, GF._conv_kind("convh_dgrad", 3, 1, Cin, B * H * W, Cout * 9), flops,
                       2 * (h.numel() + weight.numel() + dz.numel()))
        if ctx.needs_input_grad[1]:
            direct = GF.DIRECT_GRAD_ACCUM and getattr(wparam, "_ge_flat", None) is not None and wparam.grad is not None
            dw = wparam.grad if direct else torch.empty_like(weight)
            ws_n = lib.ge_h_conv3x3_wgrad_workspace(B, Cin, Cout, H, W)
            side = GF.WGRAD_STREAM if (direct and kt is None) else None
            if side is not None:
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    ws = torch.empty(ws_n, device=h.device, dtype=_f32)
                    check(lib.ge_h_conv3x3_wgrad(_p(h), _p(dz), _p(dw), _p(ws), B, Cin, Cout, H, W, inv, hsp,
                                                 int(direct), side.cuda_stream), "h_conv3x3_wgrad")
                h.record_stream(side)
                dz.record_stream(side)
            else:
                ws = torch.empty(ws_n, device=h.device, dtype=_f32)
                t0, t_mid = kt.begin_wgrad() if kt else (None, None)
                check(lib.ge_h_conv3x3_wgrad(_p(h), _p(dz), _p(dw), _p(ws), B, Cin, Cout, H, W, inv, hsp,
                                             int(direct), st), "h_conv3x3_wgrad")
                if kt:
                    kt.end(t0, GF._conv_kind("convh_wgrad", 3, 1, Cout, Cin * 9, B * H * W), flops,
                           2 * (h.numel() + dz.numel()) + 4 * weight.numel(), split=t_mid,
                           slab_bytes=4 * (ws_n + weight.numel()))
            if direct:
                dw = None
        if bparam is not None and ctx.needs_input_grad[2]:
            direct = GF.DIRECT_GRAD_ACCUM and getattr(bparam, "_ge_flat", None) is not None and bparam.grad is not None
            db = bparam.grad if direct else torch.empty(Cout, device=h.device, dtype=_f32)
            part = torch.empty(Cout * B * lib.ge_h_bn_slices(H * W) * 2, device=h.device, dtype=_f32)
            check(lib.ge_h_channel_sum(_p(dz), _p(part), _p(db), int(direct), inv, hsp, B, Cout, H * W, st),
                  "h_channel_sum")
            if direct:
                db = None
        return dh, dw, db, None, None


class _StemConvHFn(Function):
    """The stem: 3x3 / s1 / p1 conv of the fp32 NCHW image (1 or 3 channels, no gradient) straight into the blocked fp16 domain;
    returns (z, stats)."""

    @staticmethod
    def forward(ctx, x, weight, bias, want_stats):
        ctx.set_materialize_grads(False)
        x = GF._c(x)
        weight = GF._c(weight)
        B, Cin, H, W = x.shape
        Cout = weight.shape[0]
        GF.h_scale_forward_update(x.device)
        z = torch.empty((B, Cout // 32, H, W, 32), device=x.device, dtype=_f16)
        stats = torch.empty((Cout, B * H * W // 64, 3), device=x.device, dtype=_f32) if want_stats else None
        check(lib.ge_h_stem3x3_fwd(_p(x), _p(weight), _p(bias), _p(z), _p(stats), B, Cin, Cout, H, W, _stream()),
              "h_stem3x3_fwd")
        ctx.save_for_backward(x, weight)
        ctx.params = (weight, bias)
        if want_stats:
            ctx.mark_non_differentiable(stats)
            return z, stats
        return z

    @staticmethod
    def backward(ctx, dz, *rest):
        x, weight = ctx.saved_tensors
        wparam, bparam = ctx.params
        dz = _hc(dz)
        B, Cin, H, W = x.shape
        Cout = weight.shape[0]
        st = _stream()
        _, inv, hsp = GF.h_scale_args(x.device)
        dw = db = None
        if ctx.needs_input_grad[1]:
            direct = GF.DIRECT_GRAD_ACCUM and getattr(wparam, "_ge_flat", None) is not None and wparam.grad is not None
            dw = wparam.grad if direct else torch.empty_like(weight)
            ws = torch.empty(lib.ge_h_stem3x3_wgrad_workspace(B, Cin, Cout, H, W), device=x.device, dtype=_f32)
            check(lib.ge_h_stem3x3_wgrad(_p(x), _p(dz), _p(dw), _p(ws), B, Cin, Cout, H, W, inv, hsp, int(direct), st),
                  "h_stem3x3_wgrad")
            if direct:
                dw = None
        if bparam is not None and ctx.needs_input_grad[2]:
            direct = GF.DIRECT_GRAD_ACCUM and getattr(bparam, "_ge_flat", None) is not None and bparam.grad is not None
            db = bparam.grad if direct else torch.empty(Cout, device=x.device, dtype=_f32)
            part = torch.empty(Cout * B * lib.ge_h_bn_slices(H * W) * 2, device=x.device, dtype=_f32)
            check(lib.ge_h_channel_sum(_p(dz), _p(part), _p(db), int(direct), inv, hsp, B, Cout, H * W, st), "h_channel_sum")
            if direct:
                db = None
        return None, dw, db, None


def stem_supported(x, conv):
    return (not is_blocked(x) and x.dim() == 4 and not x.requires_grad and conv.kernel_size == (3, 3) and conv.stride == (1, 1)
            and conv.padding == (1, 1) and conv.groups == 1
            and bool(lib.ge_h_stem3x3_supported(x.shape[0], conv.in_channels, conv.out_channels, x.shape[2], x.shape[3])))


def conv3x3(h, weight, bias=None, cache=None, bn_stats=False):
    return _ConvHFn.apply(h, weight, bias, cache, bool(bn_stats))


class _BatchNormHFn(Function):
    """Train-/eval-mode BatchNorm2d (+ ReLU) on a blocked fp16 tensor: statistics from the conv epilogue's moments
    (`partial`), per segment of a merged pass, SyncBN over `group` -- the general path of functional._BatchNormFn with the
    two passes over the activation done in fp16."""

    @staticmethod
    def forward(ctx, z, gamma, beta, running_mean, running_var, training, momentum, eps, relu, group, partial, segments,
                pool=False):
        z = _hc(z)
        B, CB, H, W, _ = z.shape
        C, HW = CB * 32, H * W
        st = _stream()
        dev = z.device
        bounds = GF._segment_bounds(B, segments if training else None)
        S = len(bounds)
        plane = C * HW * 2          # bytes per sample
        world = 1
        if training:
            if partial is None:
                raise RuntimeError("half.batch_norm: train mode needs the conv epilogue's moments (bn_stats=True)")
            mean = torch.empty((S, C), device=dev, dtype=_f32)
            invstd = torch.empty((S, C), device=dev, dtype=_f32)
            nb = partial.numel() // (C * 3)          # one triple per 64 pixels, samples in order
            per = HW // 64
            if group is None:
                for s, (b0, bs) in enumerate(bounds):
                    check(lib.ge_bn_finalize(_p(partial) + b0 * per * 12, nb * 3, 3, bs * per, C, eps, momentum, None,
                                             _p(mean[s]), _p(invstd[s]), _p(running_mean), _p(running_var), st),
                          "bn_finalize")
            else:
                import torch.distributed as dist

                world = dist.get_world_size(group)
                stats = torch.empty((S, C * 3), device=dev, dtype=_f32)
                for s, (b0, bs) in enumerate(bounds):
                    check(lib.ge_bn_finalize(_p(partial) + b0 * per * 12, nb * 3, 3, bs * per, C, eps, momentum,
                                             _p(stats[s]), None, None, None, None, st), "bn_finalize_local")
                gathered = torch.empty((world, S, C * 3), device=dev, dtype=_f32)
                dist.all_gather_into_tensor(gathered.view(-1), stats.view(-1), group=group)
                GF.SYNC_BN_STATS[0] += 1
                GF.SYNC_BN_STATS[2] += 4 * stats.numel()
                for s in range(S):
                    check(lib.ge_bn_finalize(_p(gathered) + s * C * 12, 3, S * C * 3, world, C, eps, momentum, None,
                                             _p(mean[s]), _p(invstd[s]), _p(running_mean), _p(running_var), st),
                          "bn_finalize_sync")
        else:
            mean = running_mean.reshape(1, C)
            invstd = torch.rsqrt(running_var + eps).reshape(1, C)
        if pool:      # + ReLU + 2x2 max-pool in the same pass: the full-resolution activation is never written
            a = torch.empty((B, CB, H // 2, W // 2, 32), device=dev, dtype=_f16)
            for s, (b0, bs) in enumerate(bounds):
                check(lib.ge_h_bn_relu_pool_fwd(_p(z) + b0 * plane, _p(mean[s]), _p(invstd[s]), _p(gamma), _p(beta),
                                                _p(a) + b0 * plane // 4, bs, C, H, W, st), "h_bn_relu_pool_fwd")
        else:
            a = torch.empty_like(z)
            for s, (b0, bs) in enumerate(bounds):
                off = b0 * plane
                check(lib.ge_h_bn_apply(_p(z) + off, _p(mean[s]), _p(invstd[s]), _p(gamma), _p(beta), _p(a) + off, bs, C, HW,
                                        int(relu), st), "h_bn_apply")
        ctx.save_for_backward(z, gamma, beta, mean, invstd)
        ctx.cfg = (training, int(relu), group, world, bounds, bool(pool))
        ctx.params = (gamma, beta)
        return a

    @staticmethod
    def backward(ctx, da):
        z, gamma, beta, mean, invstd = ctx.saved_tensors
        training, relu, group, world, bounds, pool = ctx.cfg
        da = _hc(da)
        B, CB, H, W, _ = z.shape
        C, HW = CB * 32, H * W
        st = _stream()
        dev = z.device
        S = len(bounds)
        plane = C * HW * 2
        gparam, bparam = ctx.params
        affine = gamma is not None
        dgamma = dbeta = None
        direct = False
        if affine:
            direct = GF.DIRECT_GRAD_ACCUM and getattr(gparam, "_ge_flat", None) is not None and gparam.grad is not None \
                and getattr(bparam, "_ge_flat", None) is not None and bparam.grad is not None
            dgamma = gparam.grad if direct else torch.empty(C, device=dev, dtype=_f32)
            dbeta = bparam.grad if direct else torch.empty(C, device=dev, dtype=_f32)
        sums = torch.empty((S, C, 2), device=dev, dtype=_f32)
        slices = lib.ge_h_bn_slices(HW // 4 if pool else HW)
        _, inv, hsp = GF.h_scale_args(dev)
        for s, (b0, bs) in enumerate(bounds):
            off = b0 * plane
            part = torch.empty(C * bs * slices * 2, device=dev, dtype=_f32)
            if pool:      # da is the POOLED gradient
                check(lib.ge_h_bn_relu_pool_bwd_reduce(_p(da) + off // 4, _p(z) + off, _p(mean[s]), _p(invstd[s]), _p(gamma),
                                                       _p(beta), _p(part), _p(sums[s]), _p(dgamma), _p(dbeta),
                                                       int(direct or s > 0), inv, hsp, bs, C, H, W, st), "h_bn_relu_pool_bwd_reduce")
                continue
            check(lib.ge_h_bn_bwd_reduce(_p(da) + off, _p(z) + off, _p(mean[s]), _p(invstd[s]), _p(gamma), _p(beta), relu,
                                         _p(part), _p(sums[s]), _p(dgamma), _p(dbeta), int(direct or s > 0),
                                         inv, hsp, bs, C, HW, st), "h_bn_bwd_reduce")
        if direct:
            dgamma = dbeta = None
        scale = 1
        if not training:
            sums = torch.zeros_like(sums)
        elif group is not None:
            import torch.distributed as dist

            dist.all_reduce(sums, group=group)
            GF.SYNC_BN_STATS[1] += 1
            GF.SYNC_BN_STATS[2] += 4 * sums.numel()
            scale = world
        dz = torch.empty_like(z)
        S_host = 1.0 / inv       # the loss scale of da goes back onto the (true-unit, possibly all-reduced) sums
        for s, (b0, bs) in enumerate(bounds):
            off = b0 * plane
            if pool:
                check(lib.ge_h_bn_relu_pool_bwd_apply(_p(da) + off // 4, _p(z) + off, _p(mean[s]), _p(invstd[s]), _p(gamma),
                                                      _p(beta), _p(sums[s]), 1.0 / (bs * HW * scale), S_host, hsp, _p(dz) + off,
                                                      bs, C, H, W, st), "h_bn_relu_pool_bwd_apply")
                continue
            check(lib.ge_h_bn_bwd_apply(_p(da) + off, _p(z) + off, _p(mean[s]), _p(invstd[s]), _p(gamma), _p(beta), relu,
                                        _p(sums[s]), 1.0 / (bs * HW * scale), S_host, hsp, _p(dz) + off, bs, C, HW, st),
                  "h_bn_bwd_apply")
        return dz, dgamma, dbeta, None, None, None, None, None, None, None, None, None, None


def batch_norm(z, gamma, beta, running_mean, running_var, training, momentum=0.1, eps=1e-5, relu=False, group=None,
               partial=None, segments=None, pool=False):
    """pool=True (needs relu): + ReLU + 2x2 / stride 2 max-pool in the same pass, returns the pooled map."""
    if pool and not relu:
        raise RuntimeError("half.batch_norm: pool=True is the BatchNorm + ReLU + max-pool fusion")
    return _BatchNormHFn.apply(z, gamma, beta, running_mean, running_var, bool(training), float(momentum), float(eps),
                               bool(relu), group, partial, segments, bool(pool))


class _GroupNorm8HFn(Function):
    """nn.GroupNorm with 8 channels per group (+ ReLU) on a blocked fp16 tensor; `stats`: the conv epilogue's moments of z."""

    @staticmethod
    def forward(ctx, z, gamma, beta, eps, relu, stats):
        z = _hc(z)
        B, CB, H, W, _ = z.shape
        C, HW = CB * 32, H * W
        st = _stream()
        mean = torch.empty(B * C // 8, device=z.device, dtype=_f32)
        invstd = torch.empty(B * C // 8, device=z.device, dtype=_f32)
        check(lib.ge_h_gn8_stats(_p(stats), _p(mean), _p(invstd), B, C, HW, eps, st), "h_gn8_stats")
        a = torch.empty_like(z)
        check(lib.ge_h_gn8_apply(_p(z), _p(mean), _p(invstd), _p(gamma), _p(beta), _p(a), B, C, HW, int(relu), st), "h_gn8_apply")
        ctx.save_for_backward(z, gamma, beta, mean, invstd)
        ctx.relu = int(relu)
        ctx.params = (gamma, beta)
        return a

    @staticmethod
    def backward(ctx, da):
        z, gamma, beta, mean, invstd = ctx.saved_tensors
        gparam, bparam = ctx.params
        da = _hc(da)
        B, CB, H, W, _ = z.shape
        C, HW = CB * 32, H * W
        dev = z.device
        affine = gamma is not None
        direct = affine and GF.DIRECT_GRAD_ACCUM and getattr(gparam, "_ge_flat", None) is not None and gparam.grad is not None \
            and getattr(bparam, "_ge_flat", None) is not None and bparam.grad is not None
        dgamma = dbeta = None
        if affine:
            dgamma = gparam.grad if direct else torch.empty(C, device=dev, dtype=_f32)
            dbeta = bparam.grad if direct else torch.empty(C, device=dev, dtype=_f32)
        part = torch.empty(B * C * lib.ge_h_bn_slices(HW) * 2, device=dev, dtype=_f32)
        sums = torch.empty(B * C // 8 * 2, device=dev, dtype=_f32)
        dz = torch.empty_like(z)
        _, inv, hsp = GF.h_scale_args(dev)
        check(lib.ge_h_gn8_bwd(_p(da), _p(z), _p(mean), _p(invstd), _p(gamma), _p(beta), ctx.relu, _p(part), _p(sums), _p(dgamma),
                               _p(dbeta), int(direct), inv, hsp, _p(dz), B, C, HW, _stream()), "h_gn8_bwd")
        if direct:
            dgamma = dbeta = None
        return dz, dgamma, dbeta, None, None, None


def conv_gn8(conv, gn, h, relu=True):
    """gn(conv(h)) (+ ReLU) on a blocked fp16 tensor, for a GroupNorm with 8 channels per group (discriminator towers)."""
    z, stats = conv3x3(h, conv.weight, conv.bias, conv._pack, bn_stats=True)
    return _GroupNorm8HFn.apply(z, gn.weight, gn.bias, float(gn.eps), bool(relu), stats)


class _MaxPoolHFn(Function):
    @staticmethod
    def forward(ctx, h):
        h = _hc(h)
        B, CB, H, W, _ = h.shape
        y = torch.empty((B, CB, H // 2, W // 2, 32), device=h.device, dtype=_f16)
        check(lib.ge_h_maxpool2_fwd(_p(h), _p(y), B, CB * 32, H, W, _stream()), "h_maxpool2_fwd")
        ctx.save_for_backward(h)
        return y

    @staticmethod
    def backward(ctx, dy):
        (h,) = ctx.saved_tensors
        dy = _hc(dy)
        B, CB, H, W, _ = h.shape
        dx = torch.empty_like(h)
        check(lib.ge_h_maxpool2_bwd(_p(h), _p(dy), _p(dx), B, CB * 32, H, W, _stream()), "h_maxpool2_bwd")
        return dx


def max_pool2(h):
    return _MaxPoolHFn.apply(h)


def conv_bn(conv, bn, h, relu=True, pool=False):
    """bn(conv(h)) (+ ReLU) on a blocked fp16 tensor: the counterpart of nn.conv_bn inside a stack."""
    training = bn.training or not bn.track_running_stats
    group = None
    if training and bn.sync and torch.distributed.is_available() and torch.distributed.is_initialized() \
            and (torch.distributed.get_world_size() > 1 or bn.force_sync):
        group = bn.process_group if bn.process_group is not None else torch.distributed.group.WORLD
    segments = GF.BN_SEGMENTS if training else None
    if training and bn.track_running_stats and bn.num_batches_tracked is not None:
        bn._pending_batches += len(segments) if segments else 1
    mom = 0.1 if bn.momentum is None else bn.momentum
    rm = bn.running_mean if bn.track_running_stats else None
    rv = bn.running_var if bn.track_running_stats else None
    if not is_blocked(h):      # the stem: fp32 image in
        out = _StemConvHFn.apply(h, conv.weight, conv.bias, training)
        z, part = out if training else (out, None)
    elif training:
        z, part = conv3x3(h, conv.weight, conv.bias, conv._pack, bn_stats=True)
    else:
        z, part = conv3x3(h, conv.weight, conv.bias, conv._pack), None
    return batch_norm(z, bn.weight, bn.bias, rm, rv, training, mom, bn.eps, relu, group, part, segments, pool)
