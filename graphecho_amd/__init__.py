"""graphecho_amd: MI355X (gfx950) implementation of GraphEcho's per-frame dense-compute training path.

Layout:
  csrc/        hand-written HIP kernels + the C ABI (libgraphecho_hip.so, include/graphecho_hip.h)
  functional   autograd bindings of the kernels
  nn           torch-compatible layers on those kernels
  models/      mirror of the reference's models/* (FPN, Discriminator, Grapher family, TGCN, GModule, ...)
  utils/       mirror of the reference's utils/* used on the hot path
  ddp / optim  flat-buffer optimizers and RCCL gradient all-reduce
"""
from ._lib import lib, build, LIB_PATH  # noqa: F401

__version__ = "0.1.0"


def install_as_reference_modules():
    """Make ``import models.fpnseg`` / ``import utils.losses`` resolve to this package, so the reference's
    train_*.py scripts pick up the HIP implementation unchanged."""
    import importlib
    import sys

    for sub in ("models", "utils"):
        pkg = importlib.import_module(f"graphecho_amd.{sub}")
        sys.modules[sub] = pkg
        for name in list(sys.modules):
            if name.startswith(f"graphecho_amd.{sub}."):
                sys.modules[name.replace("graphecho_amd.", "", 1)] = sys.modules[name]
