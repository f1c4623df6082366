"""Bodies of tests/test_graphs_gpu.py, run in a process of their own (python -m tests.helpers.graph_cases <case> [arg]):
a fault inside the HIP runtime's stream capture would otherwise take the whole pytest session down with it."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

def _batches(dev, n, hw, steps):
    from graphecho_amd.trainer import synthetic_batch

    return [(synthetic_batch(n, 3, 4, hw, dev, 100 + 2 * s), synthetic_batch(n, 3, 4, hw, dev, 101 + 2 * s)[0])
            for s in range(steps)]


def case_graphed_fpn_step_is_bitwise_the_eager_step(dev):
    """Six steps (two eager warm-up calls, the capturing call, three replays): losses, FPN parameters, Adam moments,
    BatchNorm running statistics and batch counters equal the eager trainer's exactly."""
    from graphecho_amd import nn as gnn
    from graphecho_amd.trainer import GraphEchoTrainer

    workload = "fpn"
    data = _batches(dev, 4, 128, 6)
    ref = GraphEchoTrainer(dev, workload=workload, image_size=128, seed=5)
    tr = GraphEchoTrainer(dev, workload=workload, image_size=128, seed=5, graphs=True)
    assert tr.use_graphs and not ref.use_graphs
    kept = [x.clone() for (x, _m), _xt in data]
    for s, ((x, m), _xt) in enumerate(data):
        la, lb = ref.step(x, m), tr.step(x, m)
        assert torch.equal(la, lb), f"step {s}: loss {la.item()} vs {lb.item()}"
    assert tr._net.graphs() == (1, 1)
    # the batches were preloaded on the device: a replay must never write a later batch into the tensor the caller
    # passed at capture time (only framework-owned memory is read in place)
    for s, (k, ((x, _m), _xt)) in enumerate(zip(kept, data)):
        assert torch.equal(k, x), f"the caller's batch {s} was overwritten by a graph replay"
    for name in ref.optimizers:
        a, b = ref.optimizers[name], tr.optimizers[name]
        assert torch.equal(a.fp.flat, b.fp.flat), name
        assert a.fp.used == b.fp.used, name
    assert torch.equal(ref.optimizers["Net"].m, tr.optimizers["Net"].m)
    sa, sb = ref.network.state_dict(), tr.network.state_dict()      # flushes the num_batches_tracked counters
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    nb = [m.num_batches_tracked.item() for m in tr.network.modules() if isinstance(m, gnn.BatchNorm2d)]
    assert nb and all(v == 6 for v in nb)


def case_graphed_fpn_under_the_grapher_workload(dev):
    """Config-2 shape: the FPN replayed, the Graphers eager above it (their max-relative backward accumulates with
    atomics, so two eager runs already differ in the last bits: tolerance instead of equality)."""
    from graphecho_amd.trainer import GraphEchoTrainer

    data = _batches(dev, 4, 128, 6)
    ref = GraphEchoTrainer(dev, workload="fpn_grapher", image_size=128, seed=5)
    tr = GraphEchoTrainer(dev, workload="fpn_grapher", image_size=128, seed=5, graphs=True)
    for s, ((x, m), _xt) in enumerate(data):
        la, lb = ref.step(x, m).item(), tr.step(x, m).item()
        assert abs(la - lb) <= 1e-3 * max(1.0, abs(la)), f"step {s}: {la} vs {lb}"     # Adam amplifies last-bit noise
    assert tr._net.graphs() == (1, 1)
    for name in ref.optimizers:
        a, b = ref.optimizers[name].fp.flat, tr.optimizers[name].fp.flat
        assert (a - b).abs().mean().item() <= 1e-4 * a.abs().max().item(), name
        assert ref.optimizers[name].fp.used == tr.optimizers[name].fp.used


def case_graphed_full_workload_matches_eager(dev, merge):
    """Config-3-shaped step (source + target FPN passes, GModule, discriminators): separate passes are two graph slots
    replayed before one backward, the merged pass is one slot with per-segment BatchNorm statistics.  GModule's node
    sampling is seeded per trainer, so both trainers see the same random draws."""
    from graphecho_amd.trainer import GraphEchoTrainer

    os.environ["GE_MERGE_PASSES"] = merge
    data = _batches(dev, 4, 128, 5)
    res = {}
    for graphs in (False, True):
        torch.manual_seed(0)
        tr = GraphEchoTrainer(dev, workload="full", image_size=128, seed=7, graphs=graphs)
        torch.manual_seed(1)
        losses = [tr.step(x, m, xt).item() for (x, m), xt in data]
        res[graphs] = (losses, {k: o.fp.flat.clone() for k, o in tr.optimizers.items()},
                       {k: list(o.fp.used) for k, o in tr.optimizers.items()}, tr)
    la, lb = res[False][0], res[True][0]
    # the graphed step is the phased one (backward cut at the pyramid: another association of the pyramid-gradient sums
    # than the eager single backward at this batch size); Adam and GModule's sampling amplify the last bits
    for s, (a, b) in enumerate(zip(la, lb)):
        assert abs(a - b) <= (2e-3 if s < 2 else 1e-2) * max(1.0, abs(a)), f"step {s}: {a} vs {b}"
    # the phased step replays the FPN in two pieces: backbone + top-down pathway (one slot per pass, or one for the merged
    # pass) and the segmentation head (with a tape for the source frames, forward-only for the target's pseudo-labels)
    assert res[True][3]._pyr.graphs() == ((1, 1) if merge == "1" else (2, 2))
    assert res[True][3]._head.graphs() == (2, 1)
    assert all(d.graphs() == (1, 1) for d in res[True][3]._dis.values()) and len(res[True][3]._dis) == 4
    for k in res[False][1]:
        a, b = res[False][1][k], res[True][1][k]
        assert (a - b).abs().mean().item() <= 5e-4 * a.abs().max().item(), k
        assert res[False][2][k] == res[True][2][k], k


def case_graphed_step_over_one_rank_rccl_group(dev):
    """Data-parallel shape of the step on one GPU: SyncBN exchanges captured inside the graphs, gradient buckets
    launched from the replay's notifications, counters advance per replay as in eager mode."""
    import torch.distributed as dist
    from graphecho_amd import functional as GF
    from graphecho_amd import nn as gnn
    from graphecho_amd.trainer import GraphEchoTrainer

    data = _batches(dev, 2, 128, 5)
    ref = GraphEchoTrainer(dev, workload="full", image_size=128, seed=3)
    torch.manual_seed(1)
    want = [ref.step(x, m, xt).item() for (x, m), xt in data]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        tr = GraphEchoTrainer(dev, workload="full", image_size=128, distributed=True, seed=3, graphs=True)
        tr.sync.force = True
        for mod in tr.network.modules():
            if isinstance(mod, gnn.BatchNorm2d):
                mod.force_sync = True
        torch.manual_seed(1)
        got = []
        for (x, m), xt in data:
            before = list(GF.SYNC_BN_STATS)
            got.append(tr.step(x, m, xt).item())
            delta = [a - b for a, b in zip(GF.SYNC_BN_STATS, before)]
            assert delta[0] == 50 and delta[1] == 50, delta          # one merged pass: 50 BN layers each way
            assert all(tr.sync._launched)
        assert tr._pyr.graphs() == (1, 1) and tr._head.graphs() == (2, 1)
        # the plain trainer takes its BatchNorm moments in the one-launch small-layer kernels, the SyncBN path merges
        # per-slice moments: last-bit differences that Adam and GModule's data-dependent sampling amplify step by step
        for s, (a, b) in enumerate(zip(want, got)):
            assert abs(a - b) <= (2e-3 if s < 2 else 1e-2) * max(1.0, abs(a)), f"step {s}: {a} vs {b}"
    finally:
        if created:
            dist.destroy_process_group()


def case_graphs_auto_switches_with_the_batch_size(dev):
    """graphs="auto": steps of <= GRAPHS_AUTO_MAX_FRAMES frames replay, larger ones run eager, switching back and forth on
    one trainer (slots are per shape); losses follow the eager trainer's (GModule on its own stream in both)."""
    from graphecho_amd.trainer import GraphEchoTrainer, synthetic_batch

    def batch(n, seed):
        (x, m), xt = synthetic_batch(n, 3, 4, 128, dev, seed), synthetic_batch(n, 3, 4, 128, dev, seed + 1)[0]
        return x, m, xt

    small, big = batch(2, 40), batch(10, 50)         # 4 frames per step / 20 frames per step
    ref = GraphEchoTrainer(dev, workload="full", image_size=128, seed=6)
    tr = GraphEchoTrainer(dev, workload="full", image_size=128, seed=6, graphs="auto")
    assert tr.GRAPHS_AUTO_MAX_FRAMES == 16 and tr._gm_stream is not None
    used = []
    for s, b in enumerate([small, small, small, small, big, small, big, small]):
        torch.manual_seed(100 + s)
        a = ref.step(*b).item()
        torch.manual_seed(100 + s)
        g = tr.step(*b).item()
        used.append(tr.graphs_in_use())
        # Same kernels in the same order either way: the two trainers agree to the last bit in every run observed since the
        # captured memset node was replaced by a kernel (ge_common.h: ge_init_async; before that, replays after the first big
        # eager step read stale pool memory in the stride-2 1x1 data gradient and this comparison drifted to ~2e-2).
        assert abs(a - g) <= 1e-6 * max(1.0, abs(a)), f"step {s}: eager {a} vs auto {g}"
    assert used == ["all", "all", "all", "all", False, "all", False, "all"], used
    assert tr._pyr.graphs()[0] == 1 and tr._head.graphs()[0] == 2          # captured for the small shape only
    for name in ref.optimizers:
        pa, pb = ref.optimizers[name].fp.flat, tr.optimizers[name].fp.flat
        assert (pa - pb).abs().max().item() <= 1e-5 * pa.abs().max().item(), name


def case_tgcn_recurrence_replayed(dev):
    """The temporal workload with TGCN's recurrence replayed from a HIP graph (default) against the same trainer with
    GE_TGCN_GRAPH=0: six steps (two eager warm-up calls of the runner, the capturing call, three replays), dropout off."""
    from graphecho_amd.trainer import GraphEchoTrainer, synthetic_batch

    def data(seed):
        xs, ms = synthetic_batch(2, 3, 4, 128, dev, seed)
        xt, _ = synthetic_batch(2, 3, 4, 128, dev, seed + 1)

        def clip(s, t=8):
            f, mk = synthetic_batch(t, 3, 4, 128, dev, s)
            return (f.reshape(1, t, 3, 128, 128).permute(0, 2, 3, 4, 1).contiguous(),
                    mk.reshape(1, t, 4, 128, 128).permute(0, 2, 3, 4, 1).contiguous())

        cs, cm = clip(seed + 2)
        ct, _ = clip(seed + 3)
        return xs, ms, xt, {"source": cs, "target": ct, "masks": cm}

    def nodrop(m):
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0

    batch = data(60)
    os.environ["GE_TGCN_GRAPH"] = "0"
    ref = GraphEchoTrainer(dev, workload="temporal", image_size=128, seed=4, clip_len=8, transport_method="sinkhorn_distance")
    os.environ["GE_TGCN_GRAPH"] = "1"
    tr = GraphEchoTrainer(dev, workload="temporal", image_size=128, seed=4, clip_len=8, transport_method="sinkhorn_distance")
    assert "_roll_runner" not in ref.tgcn.__dict__ and "_roll_runner" in tr.tgcn.__dict__
    for t in (ref, tr):
        nodrop(t.tgcn)
        nodrop(t.graph_model)
        t.graph_model.async_seed_update = False
    for s in range(6):
        torch.manual_seed(200 + s)
        a = ref.step(*batch).item()
        torch.manual_seed(200 + s)
        b = tr.step(*batch).item()
        assert abs(a - b) <= (2e-3 if s < 3 else 2e-2) * max(1.0, abs(a)), f"step {s}: eager {a} vs replayed {b}"
    assert tr.tgcn.__dict__["_roll_runner"].graphs() == (1, 1)
    pa, pb = ref.optimizers["tgcn_p5"].fp.flat, tr.optimizers["tgcn_p5"].fp.flat
    assert (pa - pb).abs().max().item() <= 5e-3 * pa.abs().max().item()
    assert ref.optimizers["tgcn_p5"].fp.used == tr.optimizers["tgcn_p5"].fp.used


def case_side_streams_run_beside_the_main_stream(dev):
    """streams.concurrent_stream: the stream it returns completes a kernel while spin kernels occupy the streams it was
    asked to run beside -- also with an RCCL communicator initialised (its streams take hardware-queue slots; without the
    probe GModule's stream then shared the main stream's queue)."""
    import torch.distributed as dist
    from graphecho_amd import streams

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29579")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        t = torch.ones(1024, device=dev)
        dist.all_reduce(t)
        torch.cuda.synchronize()
        main = torch.cuda.current_stream(dev)
        a = streams.concurrent_stream(dev, [main])
        b = streams.concurrent_stream(dev, [main, a])
        assert a != main and b != main and a != b
        assert streams._runs_beside(a, [main], dev), "the weight-gradient stream shares a hardware queue with the main stream"
        assert streams._runs_beside(b, [main, a], dev), "GModule's stream shares a hardware queue with a stream it must overlap"
    finally:
        dist.destroy_process_group()


def case_graphed_module_falls_back_to_eager_outside_training(dev):
    from graphecho_amd.trainer import GraphEchoTrainer, synthetic_batch

    tr = GraphEchoTrainer(dev, workload="fpn", image_size=128, seed=1, graphs=True)
    x, m = synthetic_batch(2, 3, 4, 128, dev, 3)
    for _ in range(4):
        tr.step(x, m)
    assert tr._net.graphs() == (1, 1)
    with torch.no_grad():
        a = tr._net(x)[0]
    tr.network.eval()
    b = tr._net(x)[0]
    tr.network.train()
    assert a.shape == b.shape == (2, 4, 128, 128)
    assert tr._net.graphs() == (1, 1)
    # two forwards of one call site before a backward: the first one's backward must fail loudly, not use the second's
    # activations
    a1 = tr._net(x, tag="source")[0]        # the call site the trainer's steps captured
    a2 = tr._net(x, tag="source")[0]
    assert type(a1.grad_fn).__name__.startswith("_Replay")
    a2.sum().backward()
    try:
        a1.sum().backward()
    except RuntimeError as e:
        assert "overwritten" in str(e)
    else:
        raise AssertionError("stale backward went through")
    x3, m3 = synthetic_batch(3, 3, 4, 128, dev, 4)        # another batch size: a slot of its own, eager while warming up
    tr.step(x3, m3)
    # slots: the trainer's call site, the no_grad call above (forward-only slot, still warming up), the new batch size
    assert len(tr._net.slots) == 3 and tr._net.graphs() == (1, 1)


if __name__ == "__main__":
    case = globals()["case_" + sys.argv[1]]
    case(torch.device("cuda:0"), *sys.argv[2:])
    print("case ok")
