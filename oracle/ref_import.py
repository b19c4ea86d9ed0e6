"""Import the *reference* model (build container only -- /root/reference never travels).

TEST INFRASTRUCTURE ONLY.  Used by ``tests/golden/make_golden.py`` (fixture
generation) and by the CPU tests that validate the oracle against the live
reference when ``/root/reference`` exists.  ``timm`` is absent from the image;
``multiview_mpl.py:13-16`` only *executes* ``DropPath`` (identity in eval), so six
stub modules are inserted into ``sys.modules`` before the file is loaded
(SURVEY.md section 8c).
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("OPENMPL_REFERENCE", "/root/reference")
MODEL_FILE = os.path.join(REFERENCE_ROOT, "MPL", "lib", "models", "multiview_mpl.py")


def available() -> bool:
    return os.path.isfile(MODEL_FILE)


def _install_timm_stub():
    if "timm" in sys.modules and not getattr(sys.modules["timm"], "_openmpl_stub", False):
        return
    import torch.nn as nn

    class DropPath(nn.Module):
        """Stochastic depth; identity when not training or p == 0."""

        def __init__(self, drop_prob=0.0):
            super().__init__()
            self.drop_prob = float(drop_prob)

        def forward(self, x):
            if self.drop_prob == 0.0 or not self.training:
                return x
            import torch
            keep = 1.0 - self.drop_prob
            mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
            return x * mask / keep

    def _mk(name):
        m = types.ModuleType(name)
        m._openmpl_stub = True
        sys.modules[name] = m
        return m

    timm = _mk("timm")
    data = _mk("timm.data")
    models = _mk("timm.models")
    helpers = _mk("timm.models.helpers")
    layers = _mk("timm.models.layers")
    registry = _mk("timm.models.registry")
    data.IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)
    data.IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)
    helpers.load_pretrained = lambda *a, **k: None
    layers.DropPath = DropPath
    layers.to_2tuple = lambda x: (x, x)
    layers.trunc_normal_ = lambda t, std=1.0, **k: t
    registry.register_model = lambda fn: fn
    timm.data, timm.models = data, models
    models.helpers, models.layers, models.registry = helpers, layers, registry


_CACHE = {}


def load_reference_module():
    """Returns the reference's ``multiview_mpl`` python module (not copied, loaded in place)."""
    if "mod" in _CACHE:
        return _CACHE["mod"]
    if not available():
        raise FileNotFoundError(MODEL_FILE)
    _install_timm_stub()
    spec = importlib.util.spec_from_file_location("_openmpl_reference_multiview_mpl", MODEL_FILE)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    _CACHE["mod"] = mod
    return mod


def build_reference(flags: dict):
    """Instantiate the reference ``MultiView_MPL`` in eval mode with constructor kwargs ``flags``."""
    mod = load_reference_module()
    m = mod.MultiView_MPL(**flags)
    m.eval()
    return m
