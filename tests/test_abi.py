"""C-ABI boundary checks that need no GPU: the shared library loads, exports every symbol include/graphecho_hip.h
declares (and nothing the header forgot), argument validation returns error codes instead of crashing, and the
product package never imports the oracle."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from graphecho_amd._lib import lib, LIB_PATH

    if not os.path.exists(LIB_PATH):
        import __graft_entry__ as g

        g.build()
    return lib.load()


def test_header_symbols_are_exported(lib):
    from graphecho_amd._lib import parse_header, LIB_PATH

    sigs = parse_header()
    assert len(sigs) >= 50
    out = subprocess.run(["nm", "-D", "--defined-only", LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (ge_\w+)", out))
    assert set(sigs) <= exported, f"declared but not exported: {sorted(set(sigs) - exported)}"
    assert exported <= set(sigs), f"exported but not declared in the header: {sorted(exported - set(sigs))}"


def test_no_optional_families_in_the_header():
    """Round 6: the parked bf16x3 family is gone; the header has no `#ifdef GE_WITH_*` blocks -- ONE ABI, every declaration exported."""
    from graphecho_amd._lib import parse_header, HEADER

    text = open(HEADER).read()
    assert "GE_WITH_" not in text and "bx3" not in text
    sigs = parse_header()
    assert "ge_conv2d_f16_fwd" in sigs and "ge_mrconv_gather_bwd_det" in sigs and "ge_knn_topk_fused" in sigs


def test_winograd_routing_thresholds(lib):
    """csrc/ge_wino_plan.h: every routing constant of the two Winograd families pinned on either side (host functions, no GPU).
    Forward / data gradient: blocks = B * H * W / 128 * M / 64.  Weight gradient: tiles = M / 64 * C / 32, chunks = B * H / 2 * W / 16."""
    import ctypes

    fwd = lambda B, C, M, S: lib.ge_wino3x3_splits(B, C, M, S, S)
    # WN_SPLIT_TARGET = 256 (256 -> 256 @ 16 x 16: 8 blocks per image): 256 blocks = one pass, 248 = two splits, 120 = three
    assert (fwd(32, 256, 256, 16), fwd(31, 256, 256, 16), fwd(16, 256, 256, 16), fwd(15, 256, 256, 16)) == (1, 2, 2, 3)
    # WN_MIN_BLOCKS = 32: 32 blocks = eight splits (the most a routed layer gets), 24 blocks = the direct kernels
    assert (fwd(4, 256, 256, 16), fwd(3, 256, 256, 16)) == (8, 0) and lib.ge_wino3x3_supported(3, 256, 256, 16, 16) == 0
    # WN_SPLIT_MIN_CHUNKS = 4 chunks of 8 channels per split: 64 input channels allow two splits, 32 one
    assert (fwd(4, 64, 256, 16), fwd(4, 32, 256, 16)) == (2, 1)
    # workspace = (splits - 1) slabs of the output's size; geometry: W % 32 (H % 4) or W % 16 (H % 8), C % 8, M % 64
    assert lib.ge_wino3x3_workspace(4, 256, 256, 16, 16) == 7 * 4 * 256 * 16 * 16 and lib.ge_wino3x3_workspace(32, 256, 256, 64, 64) == 0
    assert [lib.ge_wino3x3_covered(2, 64, 64, h, w) for h, w in ((8, 16), (4, 32), (4, 16), (6, 32), (8, 8))] == [1, 1, 0, 0, 0]
    assert lib.ge_wino3x3_covered(2, 60, 64, 8, 16) == 0 and lib.ge_wino3x3_covered(2, 64, 96, 8, 16) == 0

    def wg(B, C, M, S):
        s, w = ctypes.c_int(), ctypes.c_int()
        routed = lib.ge_wino3x3_wgrad_plan(B, C, M, S, S, ctypes.byref(s), ctypes.byref(w))
        assert routed == lib.ge_wino3x3_wgrad_supported(B, C, M, S, S) and s.value == lib.ge_wino3x3_wgrad_splits(B, C, M, S, S)
        return routed, s.value, w.value

    # WNW_TARGET = 512 workgroups: 32 tiles -> 16 splits (two workgroups per CU kernel) on a long-K layer
    assert wg(32, 256, 256, 64) == (1, 16, 0)
    # WNW_WS_CHUNKS = 48 (256 -> 256 @ 32 x 32: 32 chunks per image): 768 chunks / 16 = 48 per split -> the warp-specialised kernel
    # (with 256 / 32 = 8 splits), 800 chunks = 50 per split -> the two-workgroup kernel
    assert wg(24, 256, 256, 32) == (1, 8, 1) and wg(25, 256, 256, 32) == (1, 16, 0)
    # WNW_MIN_CHUNKS = 16 chunks per split (256 -> 256 @ 16 x 16: 8 chunks per image): 64 chunks allow 4 splits, 96 allow 6
    assert wg(8, 256, 256, 16)[1:] == (4, 1) and wg(12, 256, 256, 16)[1:] == (6, 1)
    # WNW_MIN_GRID = 192 workgroups: 6 splits x 32 tiles = 192 routed, 5 x 32 = 160 left to the direct kernel
    assert wg(12, 256, 256, 16)[0] == 1 and wg(11, 256, 256, 16)[0] == 0 and wg(11, 256, 256, 16)[1] == 5
    # geometry: C % 32, M % 64, W % 16, H even
    assert [lib.ge_wino3x3_wgrad_covered(2, c, m, h, w) for c, m, h, w in ((32, 64, 2, 16), (16, 64, 2, 16), (32, 32, 2, 16),
                                                                           (32, 64, 3, 16), (32, 64, 2, 8))] == [1, 0, 0, 0, 0]


def test_version_and_error_channel(lib):
    assert lib.ge_abi_version() == 1
    # null pointers / bad shapes are rejected with -1 and a message, without touching the device
    rc = lib.ge_conv2d_fwd(None, None, None, None, None, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 1, 0, None)
    assert rc == -1 and "conv2d_fwd" in lib.last_error()
    rc = lib.ge_knn_topk(1, 1, 1, 1, None, 1, 1, 8, 4, 4, 9, 1, None)    # K > M
    assert rc == -1 and "K" in lib.last_error()
    rc = lib.ge_gemm(None, None, None, None, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 1.0, 0, 0, 0, None)
    assert rc == -1
    assert lib.ge_bn_num_partials(32, 64 * 64) == 32 and lib.ge_bn_num_partials(32, 128 * 128) == 128
    assert lib.ge_conv2d_wgrad_workspace(32, 256, 256, 64, 64, 3, 3, 1) > 256 * 256 * 9


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under graphecho_amd/ may import, load or execute it."""
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|oracle/|libknn_ref", re.M)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "graphecho_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not pat.search(text), f"{os.path.join(dirpath, f)} references the oracle"


def test_ops_fail_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from graphecho_amd import functional as GF

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        GF.conv2d(torch.zeros(1, 3, 8, 8), torch.zeros(4, 3, 3, 3))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        GF.relu(torch.zeros(4))


_REFERENCE_IMPORT_BLOCK = """
import graphecho_amd
graphecho_amd.install_as_reference_modules()
# the import block of the reference's trainer, in its order (train_camus_echo.py:27-36)
from utils.tools import get_world_size, get_global_rank, get_local_rank, get_master_ip
from utils.metrics import DiceScore
from utils.lr_scheduler import WarmupMultiStepLR
from utils.sinkhorn_distance import SinkhornDistance
from utils.losses import BinaryDiceLoss, DiceLoss
from models.fpnseg import FPN, Discriminator
from models.graph_matching import GModule
from models.TGCN import TGCN
import sys, utils.tools, utils.metrics, models.vig, models.affinity_layer, models.transformer, models.gradient_reversal
for cls in (FPN, Discriminator, GModule, TGCN, DiceLoss, BinaryDiceLoss, SinkhornDistance, WarmupMultiStepLR):
    assert cls.__module__.startswith("graphecho_amd."), cls
assert sys.modules["models.fpnseg"] is sys.modules["graphecho_amd.models.fpnseg"]
assert get_world_size() == "stub" and DiceScore.origin == "stub"
print("TOOLS", utils.tools.__file__)
print("METRICS", utils.metrics.__file__)
"""


@pytest.mark.parametrize("own_packages", ["regular", "namespace", "utils_only"])
def test_reference_import_block_from_a_fresh_interpreter(tmp_path, own_packages):
    """INTEGRATION.md recipe A, cold: a fresh interpreter in a directory that holds the caller's own `utils`
    (and `models`) packages -- two-line stubs written here -- runs the install call followed by the reference's
    import block. Mirrored sub-modules must come from graphecho_amd, everything else from the caller's files."""
    import subprocess
    import sys

    (tmp_path / "utils").mkdir()
    (tmp_path / "utils" / "tools.py").write_text(
        "def get_world_size(): return 'stub'\nget_global_rank = get_local_rank = get_master_ip = get_world_size\n")
    (tmp_path / "utils" / "metrics.py").write_text("class DiceScore: origin = 'stub'\n")
    # a stale same-named file of the caller must lose against the mirror
    (tmp_path / "utils" / "losses.py").write_text("raise ImportError('the caller\'s own utils.losses was imported')\n")
    if own_packages == "regular":
        (tmp_path / "utils" / "__init__.py").write_text("")
    if own_packages != "utils_only":
        (tmp_path / "models").mkdir()
        (tmp_path / "models" / "fpnseg.py").write_text("raise ImportError('the caller\'s own models.fpnseg')\n")
        if own_packages == "regular":
            (tmp_path / "models" / "__init__.py").write_text("")
    env = dict(os.environ, PYTHONPATH=ROOT, PYTHONDONTWRITEBYTECODE="1")
    out = subprocess.run([sys.executable, "-c", _REFERENCE_IMPORT_BLOCK], cwd=tmp_path, env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert f"TOOLS {tmp_path / 'utils' / 'tools.py'}" in out.stdout
    assert f"METRICS {tmp_path / 'utils' / 'metrics.py'}" in out.stdout


def test_reference_module_aliases_install_is_reversible():
    """In-process: install, import under the reference's names WITHOUT importing the mirror first, uninstall."""
    import sys
    import graphecho_amd

    def ours():
        return [k for k in sys.modules if k.partition(".")[0] in ("models", "utils")]

    saved = {k: sys.modules[k] for k in ours()}
    for k in saved:
        del sys.modules[k]
    try:
        graphecho_amd.install_as_reference_modules()
        graphecho_amd.install_as_reference_modules()         # idempotent
        assert sum(type(f).__name__ == "_ReferenceNameFinder" for f in sys.meta_path) == 1
        from models.fpnseg import FPN
        from utils.losses import DiceLoss

        assert FPN.__module__.startswith("graphecho_amd") and DiceLoss.__module__.startswith("graphecho_amd")
        with pytest.raises(ImportError):
            import utils.tools  # noqa: F401  (not mirrored, and this process has no `utils` of its own)
    finally:
        graphecho_amd.uninstall_reference_modules()
        assert not any(type(f).__name__ == "_ReferenceNameFinder" for f in sys.meta_path)
        for k in ours():
            del sys.modules[k]
        sys.modules.update(saved)


def test_fp16_storage_host_side_plans(lib):
    """Host logic of the blocked-fp16 conv family (no GPU): which layers it covers, moment parts, workspace sizes, argument
    validation without touching the device."""
    ok = lib.ge_h_conv3x3_supported
    # config 5's VGG16 layers at 48 frames of 256 x 256 -- all but the 1-channel stem
    for cin, cout, s in [(64, 64, 256), (64, 128, 128), (128, 128, 128), (128, 256, 64), (256, 256, 64), (256, 512, 32),
                         (512, 512, 32), (512, 512, 16)]:
        assert ok(48, cin, cout, s, s) == 1, (cin, cout, s)
    assert ok(48, 1, 64, 256, 256) == 0            # stem: Cin not a multiple of 32
    assert ok(48, 256, 256, 8, 8) == 0             # 8 x 8 maps: narrower than the smallest tile
    assert ok(48, 96, 128, 32, 32) == 0            # the data gradient needs Cin % 64 == 0 too
    assert ok(48, 256, 1, 64, 64) == 0             # one output channel (Discriminator.cls_logits)
    assert ok(4096, 512, 512, 64, 64) == 0         # tensors of 4 GB and more: 32-bit buffer offsets
    assert lib.ge_h_conv3x3_stat_parts(48, 64, 64) == 48 * 64 * 64 // 64
    ws = lib.ge_h_conv3x3_wgrad_workspace(48, 64, 64, 256, 256)
    assert ws % (9 * 64 * 64) == 0 and ws // (9 * 64 * 64) >= 32
    assert lib.ge_h_conv3x3_wgrad_workspace(48, 1, 64, 256, 256) == 0
    assert lib.ge_h_bn_slices(256 * 256) == 32 and lib.ge_h_bn_slices(16 * 16) == 1
    rc = lib.ge_h_conv3x3_fwd(None, None, None, None, None, 2, 64, 64, 32, 32, None)
    assert rc == -1 and "h_conv3x3_fwd" in lib.last_error()
    rc = lib.ge_h_from_f32(None, None, 1, 48, 16, 1.0, None, None)
    assert rc == -1


def test_winograd_host_side_plans(lib, monkeypatch):
    """Host logic of the fp32 Winograd family (no GPU): covered geometries, the routing plan (unsplit where the grid fills the
    chip, split over the input channels below that, nothing for grids of a few workgroups), workspace and operand sizes,
    argument validation without touching the device; functional._wino_plan's override."""
    ok, cov, splits, ws = lib.ge_wino3x3_supported, lib.ge_wino3x3_covered, lib.ge_wino3x3_splits, lib.ge_wino3x3_workspace
    # config 2's large 3x3 layers at batch 32 (forward and data gradient see the channel counts swapped): unsplit
    for c, m, s in [(256, 256, 64), (256, 128, 64), (128, 256, 64), (128, 128, 64), (64, 64, 64), (128, 128, 32), (256, 256, 32)]:
        assert ok(32, c, m, s, s) == 1 and splits(32, c, m, s, s) == 1 and ws(32, c, m, s, s) == 0, (c, m, s)
    # the per-rank steps of config 4 (8 frames): 32 x 32 and 16 x 16 levels are split over the input channels
    assert splits(8, 256, 256, 32, 32) == 1 and ws(8, 256, 256, 32, 32) == 0            # 256 workgroups: one per CU, unsplit
    s16 = splits(8, 256, 256, 16, 16)                                                    # 64 workgroups
    assert s16 >= 2 and 256 // 8 // s16 >= 4 and ws(8, 256, 256, 16, 16) == (s16 - 1) * 8 * 256 * 16 * 16
    assert splits(2, 256, 256, 64, 64) == 1                                              # two frames at 64 x 64: 256 workgroups
    assert ok(32, 512, 512, 8, 8) == 0 and cov(32, 512, 512, 8, 8) == 0                  # 8-column maps: narrower than a block of tiles
    assert ok(32, 3, 64, 256, 256) == 0           # reduction channels not a multiple of 8 (the stem)
    assert ok(32, 64, 32, 64, 64) == 0            # output channels not a multiple of 64
    assert ok(32, 64, 64, 36, 96) == 1 and ok(32, 64, 64, 34, 96) == 0      # rows: multiples of 4 (2 x 16 tiles per block)
    assert ok(512, 64, 64, 8, 16) == 1 and ok(512, 64, 64, 12, 16) == 0     # 16-column maps: 4 x 8 tiles, rows multiples of 8
    assert cov(1, 64, 64, 8, 16) == 1 and ok(1, 64, 64, 8, 16) == 0         # covered, but one workgroup: not routed
    assert lib.ge_wino3x3_weight_floats(256, 128) == 16 * 256 * 128
    assert lib.ge_wino3x3_stat_parts(32, 64, 64) == 32 * 32
    rc = lib.ge_wino3x3_fwd(None, None, None, None, None, None, None, 32, 64, 64, 64, 64, None)
    assert rc == -1 and "wino3x3_fwd" in lib.last_error()
    rc = lib.ge_wino3x3_pack_weight(None, None, 64, 64, 0, None)
    assert rc == -1
    rc = lib.ge_wino3x3_pack_weights_batched(None, None, 1, None)
    assert rc == -1
    from graphecho_amd import functional as GF

    assert GF._wino_plan(32, 256, 256, 64, 64) == (1, 0) and GF._wino_plan(1, 64, 64, 8, 16)[0] == 0
    monkeypatch.setattr(GF, "WINOGRAD_MIN_BLOCKS", 1)
    assert GF._wino_plan(1, 64, 64, 8, 16)[0] == 1 and GF._wino_plan(2, 3, 64, 256, 256)[0] == 0
    monkeypatch.setattr(GF, "WINOGRAD", False)
    assert GF._wino_plan(32, 256, 256, 64, 64)[0] == 0


def test_no_memset_or_memcpy_nodes_in_the_library():
    """Entry points may be captured into HIP graphs: initialisation is done by kernels (ge_common.h: ge_init_async), never by
    hipMemsetAsync / device-to-device hipMemcpyAsync, whose graph nodes ROCm 7.2 replays out of order with kernel nodes
    (profiles/r05_memset_node.txt).  The one host-to-device copy left is the fp16 loss scale, refused inside a capture."""
    import glob
    import re

    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "graphecho_amd", "csrc")
    hits = []
    for path in sorted(glob.glob(os.path.join(root, "*.hip")) + glob.glob(os.path.join(root, "*.h"))):
        for n, line in enumerate(open(path), 1):
            code = line.split("//")[0]
            if re.search(r"hipMemset\w*\(|hipMemcpy\w*\(", code):
                hits.append((os.path.basename(path), n, code.strip()))
    assert [h[0] for h in hits] == ["ge_half.hip"] and "hipMemcpyHostToDevice" in hits[0][2], hits


def test_sinkhorn_rpm_cooperative_plan():
    """Host side of the co-operative sinkhorn_rpm kernels: offered for one problem of the training step's sizes only (B = 1,
    N2 <= 512, 16 <= N1 <= 640); everything else keeps the launch chain (workspace 0)."""
    from graphecho_amd._lib import lib

    ws = lib.ge_sinkhorn_rpm_coop_workspace
    assert ws(1, 270, 320) == 4 + 2 * 16 * 320 * 2 and ws(1, 640, 512) > 0 and ws(1, 16, 1) > 0
    assert ws(2, 270, 320) == 0 and ws(1, 641, 100) == 0 and ws(1, 100, 513) == 0 and ws(1, 15, 100) == 0
