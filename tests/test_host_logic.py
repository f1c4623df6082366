"""Host-side logic that runs on CPU: LR schedule, flat parameter buffers, the GModule sampling plan against the
oracle's literal restatement, box extraction, and the multi-process (gloo, world_size 2) gradient exchange."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.nn as nn


def test_warmup_multistep_lr_matches_reference_semantics():
    from graphecho_amd.utils.lr_scheduler import WarmupMultiStepLR

    p = nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=0.3)
    sch = WarmupMultiStepLR(opt, (5, 8), 0.1, warmup_factor=1 / 3, warmup_iters=3, warmup_method="linear")
    lrs = []
    for _ in range(10):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sch.step()
    expect = [0.3 * (1 / 3 * (1 - e / 3) + e / 3) if e < 3 else 0.3 * 0.1 ** ((e >= 5) + (e >= 8)) for e in range(10)]
    assert np.allclose(lrs, expect)
    # the reference's configuration: constant warm-up never ends within 400 epochs -> lr / 3 forever
    opt = torch.optim.SGD([p], lr=3e-4)
    sch = WarmupMultiStepLR(opt, (90000,), 0.1, 1 / 3, 1000, "constant")
    for _ in range(400):
        opt.step()
        sch.step()
    assert abs(opt.param_groups[0]["lr"] - 1e-4) < 1e-12


def test_flat_params_views_and_used_ranges():
    from graphecho_amd.optim import FlatParams

    net = nn.Sequential(nn.Linear(4, 3), nn.ReLU(), nn.Linear(3, 2), nn.Linear(2, 2))
    before = [p.detach().clone() for p in net.parameters()]
    fp = FlatParams(net)
    assert fp.numel == sum(p.numel() for p in net.parameters())
    for p, b in zip(net.parameters(), before):
        assert torch.equal(p.detach(), b) and p.data_ptr() >= fp.flat.data_ptr()
    out = net[2](net[1](net[0](torch.randn(5, 4)))).sum()   # last Linear unused
    out.backward()
    assert fp.used == [True, True, True, True, False, False]
    assert fp.used_ranges() == [(0, 4 * 3 + 3 + 3 * 2 + 2)]
    assert torch.count_nonzero(fp.grad[fp.offsets[4]:]) == 0
    g0 = net[0].weight.grad
    assert g0.data_ptr() == fp.grad.data_ptr() and torch.count_nonzero(g0) > 0
    fp.zero_grad()
    assert torch.count_nonzero(fp.grad) == 0 and fp.used == [False] * 6


def test_masks_to_boxes_and_sampling_plan_match_oracle():
    """The sync-free restatement of find_bbox / label assignment / per-level sampling used by the HIP GModule
    gives exactly the node sets of the literal oracle (which is pinned to the reference by the golden tests)."""
    from graphecho_amd.models.graph_matching import GModule
    from oracle import gmodule as og
    from oracle.weights import det_tensor, rect_masks

    gm = GModule(256, 4, "cpu")
    masks = rect_masks(3, 4, 256, 256, seed=5)
    masks[1, 2] = 0                       # an empty class channel -> full-image box
    boxes = gm.find_bbox(masks)
    for b in range(3):
        assert torch.equal(boxes[b], og.masks_to_boxes(masks[b]))
    feats = [det_tensor(f"hl.f{l}", (3, 8, s, s)) for l, s in enumerate((64, 32, 16, 8))]
    gen = gm.graph_generator
    labels = gen.label_maps(gm.compute_locations(feats), boxes)
    counts = [[int((l > 0).sum()), int((l == 0).sum())] for l in labels]
    nodes, lab = gen.sample(feats, labels, gen.plan(counts))
    ref_nodes, ref_lab = og.sample_nodes(feats, masks, 4)
    assert torch.equal(lab, ref_lab) and torch.equal(nodes, ref_nodes)
    # the host-side plan the HIP path uses (byte label maps -> level / location index / label per row) names the same
    # rows in the same order
    import numpy as np

    host, ev, n_src = gm.prepare((feats, feats), masks, masks)
    assert ev is None and n_src == 3 and host.dtype == torch.uint8 and host.shape == (6, 64 * 64 + 32 * 32 + 16 * 16 + 64)
    level, index, node_lab, unique, present = gen.plan_rows(host.numpy()[:3], [s * s for s in (64, 32, 16, 8)])
    assert np.array_equal(node_lab, ref_lab.numpy()) and unique
    rows = []
    for lv, ix in zip(level, index):
        f = feats[lv]
        hw = f.shape[2] * f.shape[3]
        rows.append(f.reshape(f.shape[0], f.shape[1], hw)[ix // hw, :, ix % hw])
    assert torch.equal(torch.stack(rows), ref_nodes)
    assert present == [bool((level == l).any()) for l in range(4)]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run_world2(target, attempts=3):
    """Spawn two gloo ranks on a free local port and collect one result per rank.  The port is probed, released and
    re-bound by the children, so another process can grab it in between: a failed rendezvous is retried on a new port."""
    import torch.multiprocessing as mp

    last = None
    for _ in range(attempts):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=target, args=(r, 2, _free_port(), q)) for r in range(2)]
        port = procs[0]._args[2]
        procs[1]._args = (1, 2, port, q)
        for p in procs:
            p.start()
        try:
            res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
            for p in procs:
                p.join(timeout=60)
            if all(p.exitcode == 0 for p in procs):
                return res
            last = RuntimeError(f"worker exit codes {[p.exitcode for p in procs]}")
        except Exception as exc:   # noqa: BLE001 -- rendezvous / queue failures of any kind are retried
            last = exc
        for p in procs:
            if p.is_alive():
                p.terminate()
            p.join(timeout=10)
    raise last


def _ddp_worker(rank, world, port, q):
    import torch.distributed as dist
    from graphecho_amd.ddp import GradSynchronizer, broadcast_parameters
    from graphecho_amd.optim import FlatParams

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)                       # different init per rank on purpose
    net = nn.Sequential(nn.Linear(6, 5), nn.Tanh(), nn.Linear(5, 3), nn.Linear(3, 3))   # last layer unused

    class Opt:                                           # stand-in for the HIP optimizers (kernel needs a GPU)
        def __init__(self, m):
            self.fp = FlatParams(m)
            self.grad_scale = 1.0

    opt = Opt(net)
    broadcast_parameters([opt.fp])
    sync = GradSynchronizer([opt], bucket_bytes=64)      # tiny buckets -> several async all-reduces
    torch.manual_seed(7 + rank)
    x = torch.randn(4, 6)
    opt.fp.zero_grad()
    sync.reset()
    net[2](net[1](net[0](x))).pow(2).sum().backward()
    sync.finish()
    # numpy, not tensors: a tensor travels through the queue as a shared-memory handle that dies with this process
    q.put((rank, opt.fp.flat.detach().numpy().copy(), (opt.fp.grad * opt.grad_scale).numpy().copy(), list(opt.fp.used),
           len(sync.buckets), x.numpy().copy()))
    dist.destroy_process_group()


def test_gradient_synchronizer_gloo_world2():
    res = _run_world2(_ddp_worker)
    t = torch.from_numpy
    (_, w0, g0, used0, nb, x0), (_, w1, g1, used1, _, x1) = [(r, t(w), t(g), u, n, t(x)) for r, w, g, u, n, x in res]
    assert nb > 2
    assert torch.equal(w0, w1), "broadcast_parameters must make replicas identical"
    assert torch.allclose(g0, g1), "all ranks must hold the same averaged gradient"
    assert used0 == used1 == [True, True, True, True, False, False]
    # reference: mean of the per-rank gradients computed in one process
    torch.manual_seed(100)
    net = nn.Sequential(nn.Linear(6, 5), nn.Tanh(), nn.Linear(5, 3), nn.Linear(3, 3))
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    assert torch.equal(flat, w0)
    grads = []
    for x in (x0, x1):
        net.zero_grad()
        net[2](net[1](net[0](x))).pow(2).sum().backward()
        grads.append(torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1)
                                for p in net.parameters()]))
    assert torch.allclose(g0, (grads[0] + grads[1]) / 2, atol=1e-6)


def _ddp_shared_worker(rank, world, port, q):
    import torch.distributed as dist
    from graphecho_amd.ddp import GradSynchronizer, broadcast_parameters
    from graphecho_amd.optim import FlatParams

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(5)
    body = nn.Sequential(nn.Linear(6, 6), nn.Tanh())          # applied TWICE per forward (like FPN's conv2 / gn1)
    head_a, head_b = nn.Linear(6, 2), nn.Linear(6, 2)         # second model: its own optimizer and buckets

    class Opt:
        def __init__(self, m):
            self.fp = FlatParams(m)
            self.grad_scale = 1.0

    opts = [Opt(body), Opt([head_a, head_b])]
    broadcast_parameters([o.fp for o in opts])
    sync = GradSynchronizer(opts, bucket_bytes=32)
    torch.manual_seed(20 + rank)
    x = torch.randn(3, 6)
    for o in opts:
        o.fp.zero_grad()
    sync.reset()
    h = body(body(x))
    # rank 1 reaches the heads in the opposite order: bucket all-reduces must still be issued in one fixed order
    loss = (head_a(h).sum() + head_b(h).pow(2).sum()) if rank == 0 else (head_b(h).pow(2).sum() + head_a(h).sum())
    loss.backward()
    sync.finish()
    q.put((rank, [(o.fp.grad * o.grad_scale).numpy().copy() for o in opts], x.numpy().copy()))
    dist.destroy_process_group()


def test_gradient_synchronizer_shared_parameters_and_two_models_gloo_world2():
    """A layer used twice in one forward must be reduced once, after its LAST contribution; two optimizers' buckets are
    launched in one fixed order on every rank even when the ranks' autograd orders differ."""
    res = _run_world2(_ddp_shared_worker)
    (_, g0, x0), (_, g1, x1) = [(r, [torch.from_numpy(a) for a in g], torch.from_numpy(x)) for r, g, x in res]
    for a, b in zip(g0, g1):
        assert torch.allclose(a, b), "replicas hold different averaged gradients"
    torch.manual_seed(5)
    body = nn.Sequential(nn.Linear(6, 6), nn.Tanh())
    head_a, head_b = nn.Linear(6, 2), nn.Linear(6, 2)
    want = []
    for x in (x0, x1):
        for m in (body, head_a, head_b):
            m.zero_grad()
        h = body(body(x))
        (head_a(h).sum() + head_b(h).pow(2).sum()).backward()
        want.append([torch.cat([p.grad.reshape(-1) for p in body.parameters()]),
                     torch.cat([p.grad.reshape(-1) for m in (head_a, head_b) for p in m.parameters()])])
    for k in range(2):
        assert torch.allclose(g0[k], (want[0][k] + want[1][k]) / 2, atol=1e-6)


def _ddp_hold_worker(rank, world, port, q):
    import torch.distributed as dist
    from graphecho_amd.ddp import GradSynchronizer, broadcast_parameters
    from graphecho_amd.optim import FlatParams

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(9)
    early, late = nn.Linear(5, 5), nn.Linear(5, 2)

    class Opt:
        def __init__(self, m):
            self.fp = FlatParams(m)
            self.grad_scale = 1.0

    opts = [Opt(early), Opt(late)]
    broadcast_parameters([o.fp for o in opts])
    sync = GradSynchronizer(opts, bucket_bytes=1 << 20)
    torch.manual_seed(30 + rank)
    x = torch.randn(4, 5)
    for o in opts:
        o.fp.zero_grad()
    sync.reset()
    # first autograd call: the early model only -- its bucket is exchanged from the hooks
    early(x).pow(2).sum().backward()
    launched_after_first = list(sync._launched)
    # second call under hold (trainer: a backward pass that runs on a side stream): hooks mark, nothing is exchanged
    sync.hold = True
    late(x.detach()).sum().backward()
    sync.hold = False
    launched_under_hold, ready_under_hold = list(sync._launched), list(sync._ready)
    sync.mark_complete([opts[1]])
    launched_after_mark = list(sync._launched)
    sync.finish()
    q.put((rank, launched_after_first, launched_under_hold, ready_under_hold, launched_after_mark,
           [(o.fp.grad * o.grad_scale).numpy().copy() for o in opts]))
    dist.destroy_process_group()


def test_gradient_synchronizer_hold_defers_the_exchange_to_mark_complete_gloo_world2():
    """GradSynchronizer.hold (round 4: set by the trainer around a backward pass that runs on GModule's stream): gradient
    hooks only mark buckets ready; mark_complete() after the streams have joined exchanges them, in the fixed order, and
    every rank ends with the same averaged gradients."""
    res = _run_world2(_ddp_hold_worker)
    for _rank, first, held, ready, marked, _g in res:
        assert first == [True, False], first               # the early model's bucket went out from its hooks
        assert held == [True, False] and ready[1], (held, ready)      # under hold: ready, not launched
        assert marked == [True, True], marked
    (_, _, _, _, _, g0), (_, _, _, _, _, g1) = res
    for a, b in zip(g0, g1):
        assert np.allclose(a, b) and np.abs(a).max() > 0


def _ddp_deferred_worker(rank, world, port, q):
    import torch.distributed as dist
    from graphecho_amd.ddp import GradSynchronizer, broadcast_parameters
    from graphecho_amd.optim import FlatParams

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(11)
    net, gm = nn.Linear(5, 5), nn.Sequential(nn.Linear(5, 4), nn.Linear(4, 2))

    class Opt:
        def __init__(self, m):
            self.fp = FlatParams(m)
            self.grad_scale = 1.0

    opts = [Opt(gm), Opt(net)]          # phased order: the deferred model in front of the FPN stand-in
    broadcast_parameters([o.fp for o in opts])
    sync = GradSynchronizer(opts, bucket_bytes=1 << 20)
    sync.defer_fps = {id(opts[0].fp)}
    torch.manual_seed(40 + rank)
    x1, x2 = torch.randn(4, 5), torch.randn(3, 5)
    for o in opts:
        o.fp.zero_grad()
    sync.reset()
    # the temporal step under GE_GM_FIRST: GModule's first call and ITS backward, then its second call and a second backward --
    # every parameter of `gm` receives gradient in both autograd calls
    gm(x1).pow(2).sum().backward()
    after_first = (list(sync._launched), list(sync._ready), list(sync._pending))
    local_first = opts[0].fp.grad.clone()
    gm(x2).sum().backward()
    after_second = (list(sync._launched), list(sync._ready), list(sync._pending))
    local_sum = opts[0].fp.grad.clone()
    sync.mark_complete([opts[0]])
    after_mark = list(sync._launched)
    net(x1).sum().backward()
    sync.finish()
    q.put((rank, after_first, after_second, after_mark, local_first.numpy(), local_sum.numpy(),
           [(o.fp.grad * o.grad_scale).numpy().copy() for o in opts]))
    dist.destroy_process_group()


def test_deferred_buckets_two_autograd_calls_gloo_world2():
    """a model in GradSynchronizer.defer_fps (GModule in the temporal step) receives gradient in TWO autograd calls.
    Its bucket must not be exchanged after the first call although every parameter has been seen once; hook counts never go
    negative; mark_complete() releases it; the exchanged gradient is the mean over ranks of the SUM of both calls."""
    res = _run_world2(_ddp_deferred_worker)
    for _rank, first, second, marked, _lf, _ls, _g in res:
        assert first[0] == [False, False] and first[1] == [False, False], first
        assert second[0] == [False, False] and second[1] == [False, False], second
        assert min(first[2] + second[2]) >= 0, (first[2], second[2])
        assert marked == [True, False], marked
    (_, _, _, _, lf0, ls0, g0), (_, _, _, _, lf1, ls1, g1) = res
    assert np.abs(ls0 - lf0).max() > 0                       # the second call did add to the buffer
    assert np.allclose(g0[0], (ls0 + ls1) / 2, atol=1e-6) and np.allclose(g0[0], g1[0])
    assert np.allclose(g0[1], g1[1]) and np.abs(g0[1]).max() > 0


def _ddp_used_map_worker(rank, world, port, q):
    import torch.distributed as dist
    from graphecho_amd.ddp import GradSynchronizer, broadcast_parameters
    from graphecho_amd.optim import FlatParams

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(11)
    trunk, branch = nn.Linear(4, 4), nn.Linear(4, 2)     # `branch` plays GModule: skipped on data-dependent steps

    class Opt:   # FlatSGD's range logic on CPU (the fused kernels need the GPU): steps exactly fp.used_ranges()
        def __init__(self, m):
            self.fp, self.grad_scale = FlatParams(m), 1.0
            self.buf = torch.zeros_like(self.fp.flat)

        def zero_grad(self):
            self.fp.zero_grad()

        @torch.no_grad()
        def step(self):
            for a, b in self.fp.used_ranges():
                self.buf[a:b].mul_(0.9).add_(self.fp.grad[a:b] * self.grad_scale)
                self.fp.flat[a:b].sub_(0.1 * self.buf[a:b])

    opt = Opt([trunk, branch])
    broadcast_parameters([opt.fp])
    sync = GradSynchronizer([opt], bucket_bytes=16)
    log = []
    for step in range(4):
        torch.manual_seed(50 + 10 * step + rank)
        x = torch.randn(3, 4)
        opt.zero_grad()
        sync.reset()
        h = trunk(x)
        # step 0: nobody uses the branch (the early return hits the FIRST step -- the stale-map bug dropped the branch for
        # the next 99 steps); step 1: only rank 1 uses it; step 2: both; step 3: nobody again
        use = {0: False, 1: rank == 1, 2: True, 3: False}[step]
        loss = h.pow(2).sum() + (branch(h).sum() if use else 0.0)
        loss.backward()
        sync.finish()
        before = opt.fp.flat.detach().clone()
        opt.step()
        log.append((list(opt.fp.used), (opt.fp.flat.detach() - before).numpy().copy()))
    q.put((rank, log, sync.used_syncs))
    dist.destroy_process_group()


def test_used_parameter_map_is_agreed_every_step_gloo_world2():
    """the cross-rank "which parameters got a gradient" map must follow the data every step.  A parameter used
    on ANY rank is stepped on ALL ranks (identically); one used nowhere is left untouched (torch's `grad is None` skip:
    no momentum / weight-decay update with a zero gradient)."""
    res = _run_world2(_ddp_used_map_worker)
    (_, log0, n0), (_, log1, n1) = res
    assert n0 == n1 == 4
    want_branch = [False, True, True, False]
    for step in range(4):
        (u0, d0), (u1, d1) = log0[step], log1[step]
        assert u0 == u1 == [True, True, want_branch[step], want_branch[step]], (step, u0, u1)
        assert np.array_equal(d0, d1), "replicas must take the same step"
        branch_delta = d0[4 * 4 + 4:]
        if want_branch[step]:
            assert np.abs(branch_delta).max() > 0
        else:
            assert np.abs(branch_delta).max() == 0, "a parameter without a gradient anywhere must not move (momentum!)"


def _ddp_rs_ag_worker(rank, world, port, q):
    import torch.distributed as dist
    from graphecho_amd.ddp import GradSynchronizer, broadcast_parameters
    from graphecho_amd.optim import FlatParams

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class Opt:   # momentum SGD over FlatParams on CPU with the optimizers' (within, finish) contract
        def __init__(self, m):
            self.fp, self.grad_scale, self.finished = FlatParams(m), 1.0, 0
            self.buf = torch.zeros_like(self.fp.flat)

        @torch.no_grad()
        def step(self, within=None, finish=True):
            for a, b in self.fp.used_ranges(within):
                self.buf[a:b].mul_(0.9).add_(self.fp.grad[a:b] * self.grad_scale)
                self.fp.flat[a:b].sub_(0.1 * self.buf[a:b])
            if finish:
                self.finish_step()

        def finish_step(self):
            self.finished += 1

    out = {}
    for mode in ("allreduce", "rs_ag"):
        torch.manual_seed(3)
        net = nn.Sequential(nn.Linear(7, 5), nn.Tanh(), nn.Linear(5, 3))      # 58 parameters: odd sizes on purpose
        side = nn.Linear(5, 2)                                                 # used on rank 0 only, then by nobody
        opts = [Opt(net), Opt(side)]
        broadcast_parameters([o.fp for o in opts])
        sync = GradSynchronizer(opts, bucket_bytes=64, mode=mode)
        assert len(sync.buckets) > 3
        if mode == "rs_ag":
            assert all((b - a) % world == 0 for _, a, b, _ in sync.buckets)
            assert any(len(v) > 1 for v in sync._of_param.values()), "no parameter straddles two buckets"
        for step in range(3):
            torch.manual_seed(100 + 10 * step + rank)
            x = torch.randn(4, 7)
            for o in opts:
                o.fp.zero_grad()
            sync.reset()
            h = net[1](net[0](x))
            loss = net[2](h).pow(2).sum() + (side(h).sum() if (rank == 0 and step < 2) else 0.0)
            loss.backward()
            sync.finish()
            sync.step_optimizers()
        assert all(o.finished == 3 for o in opts)
        out[mode] = [o.fp.flat.detach().numpy().copy() for o in opts] + [dict(sync.comm_stats)]
    q.put((rank, out))
    dist.destroy_process_group()


def test_sharded_exchange_matches_allreduce_gloo_world2():
    """GradSynchronizer(mode="rs_ag"): reduce-scatter of every bucket, optimizer on the owned shards only, all-gather of
    the updated parameters -- replicas identical to each other and to the all-reduce mode bit for bit (a two-rank sum is
    the same sum either way), including a parameter that straddles two buckets, the zero padding behind the last
    parameter, and parameters that only one rank (then no rank) used."""
    res = _run_world2(_ddp_rs_ag_worker)
    (_, o0), (_, o1) = res
    for mode in ("allreduce", "rs_ag"):
        for a, b in zip(o0[mode][:2], o1[mode][:2]):
            assert np.array_equal(a, b), f"{mode}: replicas differ"
    for a, b in zip(o0["allreduce"][:2], o0["rs_ag"][:2]):
        assert np.array_equal(a, b), "sharded exchange changed the result"
    assert o0["rs_ag"][2]["collectives"] == 2 * o0["allreduce"][2]["collectives"] or \
        o0["rs_ag"][2]["collectives"] > o0["allreduce"][2]["collectives"]


def test_cluster_pool_matches_inline_fit():
    """The seed-bank clustering worker processes return exactly what the inline scikit-learn fit returns, in
    submission order, and the worker script does not import torch (it must stay a light, GPU-free process)."""
    import numpy as np
    from graphecho_amd import _cluster_worker
    from graphecho_amd.cluster_pool import ClusterPool

    src = open(_cluster_worker.__file__).read()
    assert "import torch" not in src and "graphecho_amd" not in src.split('"""', 2)[2]
    rng = np.random.default_rng(5)
    jobs = []
    for n in (24, 37, 60, 45):
        a = rng.normal(0, 1, (n // 2, 256)).astype(np.float32)
        b = rng.normal(3, 1, (n - n // 2, 256)).astype(np.float32)
        rows = np.concatenate([a[:1] * 0.9, a, b]).astype(np.float32)
        jobs.append((rows, n // 2))
    pool = ClusterPool(workers=2)
    try:
        tickets = [pool.submit(r, k) for r, k in jobs]
        got = [pool.result(t) for t in tickets]
    finally:
        pool.close()
    for (rows, k), g in zip(jobs, got):
        ref = _cluster_worker.spectral_keep(rows, k)
        assert g.dtype == bool and g.shape == (rows.shape[0] - 1,)
        assert np.array_equal(g, ref)
        assert 0 < g.sum() < g.size   # the two blobs are separated: the seed's blob is kept, the other dropped


def test_gradient_exchange_launch_order():
    """GradSynchronizer(launch_order=...): buckets are sent model by model in the given model order, each model's buckets
    last-to-first; anything but a permutation is refused."""
    import pytest
    from graphecho_amd.ddp import GradSynchronizer
    from graphecho_amd.optim import FlatSGD

    torch.manual_seed(0)
    mods = [torch.nn.Sequential(torch.nn.Linear(64, 64), torch.nn.Linear(64, 64), torch.nn.Linear(64, 8)) for _ in range(3)]
    opts = [FlatSGD(m, lr=0.1) for m in mods]
    sync = GradSynchronizer(opts, bucket_bytes=4 * 64 * 64)          # two or three buckets per model
    per = [[i for i, b in enumerate(sync.buckets) if b[0] is o.fp] for o in opts]
    assert all(len(p) >= 2 for p in per)
    assert sync._order == per[0][::-1] + per[1][::-1] + per[2][::-1]
    sync = GradSynchronizer(opts, bucket_bytes=4 * 64 * 64, launch_order=[2, 0, 1])
    assert sync._order == per[2][::-1] + per[0][::-1] + per[1][::-1]
    with pytest.raises(ValueError):
        GradSynchronizer(opts, launch_order=[0, 0, 1])
