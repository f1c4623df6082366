"""graphecho_amd: MI355X (gfx950) implementation of GraphEcho's per-frame dense-compute training path.

Layout:
  csrc/        hand-written HIP kernels + the C ABI (libgraphecho_hip.so, include/graphecho_hip.h)
  functional   autograd bindings of the kernels
  nn           torch-compatible layers on those kernels
  models/      mirror of the reference's models/* (FPN, Discriminator, Grapher family, TGCN, GModule, ...)
  utils/       mirror of the reference's utils/* used on the hot path
  ddp / optim  flat-buffer optimizers and RCCL gradient all-reduce
"""
import importlib as _importlib
import importlib.machinery as _machinery

from ._lib import lib, build, LIB_PATH  # noqa: F401

__version__ = "0.1.0"


_MIRRORED = {
    "models": ("fpnseg", "graph_matching", "TGCN", "vig", "affinity_layer", "transformer", "gradient_reversal"),
    "utils": ("losses", "sinkhorn_distance", "lr_scheduler"),
}


class _AliasLoader:
    """Loader that hands out an already imported ``graphecho_amd.*`` module under the reference's name."""

    def __init__(self, module):
        self._module = module

    def create_module(self, spec):
        return self._module

    def exec_module(self, module):      # the module body ran under its real name
        pass


class _ReferenceNameFinder:
    """``sys.meta_path`` finder for the reference's import lines (``train_camus_echo.py:27-36``,
    ``train_cardiac_uda.py:30-44``): ``models.X`` / ``utils.X`` resolve to ``graphecho_amd.models.X`` /
    ``graphecho_amd.utils.X`` for the sub-modules this package mirrors -- the SAME module object, imported under
    its real name, so its relative imports work -- and to nothing else: ``utils.tools``, ``utils.metrics``,
    ``datasets.*`` and the parent packages themselves stay the caller's own (found on ``sys.path`` by the regular
    finders). Only when the caller has no ``models`` / ``utils`` package at all is an empty namespace stand-in made,
    so that the mirrored sub-modules still import."""

    def find_spec(self, fullname, path=None, target=None):
        parent, _, leaf = fullname.partition(".")
        if parent not in _MIRRORED:
            return None
        if not leaf:
            if _machinery.PathFinder.find_spec(fullname, None) is not None:
                return None                                   # the caller's own package: regular import
            spec = _machinery.ModuleSpec(fullname, None, is_package=True)
            spec.submodule_search_locations = []              # namespace stand-in, no files of its own
            return spec
        if leaf not in _MIRRORED[parent]:
            return None
        module = _importlib.import_module(f"graphecho_amd.{parent}.{leaf}")
        return _machinery.ModuleSpec(fullname, _AliasLoader(module), origin=getattr(module, "__file__", None))


def install_as_reference_modules():
    """Make the reference's ``from models.fpnseg import FPN`` / ``from utils.losses import DiceLoss`` lines resolve
    to this package, from a fresh interpreter, without shadowing what is not mirrored (``utils.tools``,
    ``utils.metrics``, ``datasets.*`` keep resolving to the caller's files). Idempotent. Call it before the
    reference's import block; sub-modules of the caller's own ``models`` / ``utils`` that were imported EARLIER
    under a mirrored name are replaced."""
    import sys

    if not any(isinstance(f, _ReferenceNameFinder) for f in sys.meta_path):
        sys.meta_path.insert(0, _ReferenceNameFinder())
    for parent, leaves in _MIRRORED.items():
        for leaf in leaves:
            stale = sys.modules.get(f"{parent}.{leaf}")
            if stale is not None and not getattr(stale, "__name__", "").startswith("graphecho_amd."):
                del sys.modules[f"{parent}.{leaf}"]
        pkg = sys.modules.get(parent)
        if pkg is not None and getattr(pkg, "__name__", "").startswith("graphecho_amd."):
            del sys.modules[parent]                           # an alias of the whole package from an older install


def uninstall_reference_modules():
    """Undo ``install_as_reference_modules`` (tests)."""
    import sys

    sys.meta_path[:] = [f for f in sys.meta_path if not isinstance(f, _ReferenceNameFinder)]
    for name in [n for n in sys.modules if n.partition(".")[0] in _MIRRORED]:
        mod = sys.modules[name]
        if getattr(mod, "__name__", "").startswith("graphecho_amd.") or getattr(mod, "__file__", 1) is None:
            del sys.modules[name]
