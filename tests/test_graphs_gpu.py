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
