"""HIP-graph replay of the FPN / discriminator passes (graphecho_amd/graphs.py) must reproduce the eager step: same
kernels, same order, so the comparison is bit-for-bit wherever the eager path itself is reproducible.  The bodies live
in tests/helpers/graph_cases.py and run in a child process each (a fault inside the runtime's stream capture would otherwise
take the pytest session down).  Round 2 saw one unexplained fault in ~40 runs and retried a case that died on a signal;
at this head 300 fresh-process runs of the three capture-heavy cases (tools/stress_graph_capture.sh,
profiles/r03_graph_capture_stress.txt) produced none, and a case now runs exactly once: any death fails the test."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_case(name, *args):
    cmd = [sys.executable, "-X", "faulthandler", "-m", "tests.helpers.graph_cases", name, *args]
    env = dict(os.environ, PYTHONWARNINGS="ignore")
    env.pop("GE_MERGE_PASSES", None)
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "case ok" in out.stdout, (out.stdout[-1500:] + "\n" + out.stderr[-3000:])


def test_graphed_fpn_step_is_bitwise_the_eager_step(dev):
    _run_case("graphed_fpn_step_is_bitwise_the_eager_step")


def test_graphed_fpn_under_the_grapher_workload(dev):
    _run_case("graphed_fpn_under_the_grapher_workload")


@pytest.mark.parametrize("merge", ["0", "1"])
def test_graphed_full_workload_matches_eager(dev, merge):
    _run_case("graphed_full_workload_matches_eager", merge)


def test_graphed_step_over_one_rank_rccl_group(dev):
    _run_case("graphed_step_over_one_rank_rccl_group")


def test_graphed_module_falls_back_to_eager_outside_training(dev):
    _run_case("graphed_module_falls_back_to_eager_outside_training")


def test_graphs_auto_switches_with_the_batch_size(dev):
    _run_case("graphs_auto_switches_with_the_batch_size")


def test_side_streams_run_beside_the_main_stream(dev):
    _run_case("side_streams_run_beside_the_main_stream")


def test_tgcn_recurrence_replayed(dev):
    _run_case("tgcn_recurrence_replayed")


def test_stride2_1x1_dgrad_replayed_over_dirty_memory(dev):
    """The stride-2 1x1 data gradient (ResNet downsample convs) writes every other position from the GEMM and the rest from
    an initialising pass.  That pass was a hipMemsetAsync: captured, it became a memset node that ROCm 7.2 does not order
    with the kernel nodes on replay (tools/memset_node_check.py, profiles/r05_memset_node.txt), so a replay kept whatever
    the pool block held.  Replays over NaN-filled output must give the eager result, with and without an addend."""
    import torch
    from graphecho_amd import functional as GF

    torch.manual_seed(3)
    w = torch.randn(128, 64, 1, 1, device=dev) * 0.1
    x = torch.randn(2, 64, 16, 16, device=dev, requires_grad=True)
    gy = torch.randn(2, 128, 8, 8, device=dev)

    def fn(t):
        return GF.conv2d(t, w, None, 2, 0, 1, None)

    (want,) = torch.autograd.grad(fn(x), x, gy)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        (got,) = torch.autograd.grad(fn(x), x, gy)
    for _ in range(3):
        got.fill_(float("nan"))
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(got, want)
