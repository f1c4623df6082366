"""Import shim for the read-only reference tree (authoring container only; never used at test/bench time).

`models.vig` / `models.TGCN` import `timm`, which is not installed: an in-memory stub provides the four names
they touch (only DropPath is ever instantiated, and only for drop_path > 0).  `np.float` was removed from
numpy and is restored for vig.py:74.
"""
import sys
import types

import numpy as np
import torch

REF = "/root/reference"


def setup():
    sys.dont_write_bytecode = True
    if REF not in sys.path:
        sys.path.insert(0, REF)
    if "timm" not in sys.modules:
        for n in ["timm", "timm.data", "timm.models", "timm.models.helpers", "timm.models.layers",
                  "timm.models.registry"]:
            sys.modules[n] = types.ModuleType(n)
        sys.modules["timm.data"].IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)
        sys.modules["timm.data"].IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)
        sys.modules["timm.models.helpers"].load_pretrained = lambda *a, **k: None
        layers = sys.modules["timm.models.layers"]

        class DropPath(torch.nn.Module):
            def __init__(self, p=0.0):
                super().__init__()
                self.p = p

            def forward(self, x):
                return x

        layers.DropPath = DropPath
        layers.to_2tuple = lambda x: (x, x)
        layers.trunc_normal_ = torch.nn.init.trunc_normal_
        sys.modules["timm.models.registry"].register_model = lambda f: f
    if not hasattr(np, "float"):
        np.float = float
