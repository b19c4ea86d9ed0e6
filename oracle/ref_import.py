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


CONFIG_FILE = os.path.join(REFERENCE_ROOT, "MPL", "lib", "core", "config.py")
CONFIG_DIR = os.path.join(REFERENCE_ROOT, "MPL", "configs")


def _install_easydict_stub():
    """``easydict`` is absent from the image; core/config.py only needs attribute access on nested dicts."""
    if "easydict" in sys.modules and not getattr(sys.modules["easydict"], "_openmpl_stub", False):
        return

    class EasyDict(dict):
        def __init__(self, d=None, **kw):
            super().__init__()
            for k, v in dict(d or {}, **kw).items():
                self[k] = v

        @classmethod
        def _wrap(cls, v):
            if isinstance(v, dict) and not isinstance(v, cls):
                return cls(v)
            if isinstance(v, (list, tuple)):
                return type(v)(cls._wrap(x) for x in v)
            return v

        def __setitem__(self, k, v):
            super().__setitem__(k, self._wrap(v))

        def __setattr__(self, k, v):
            self[k] = v

        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

    m = types.ModuleType("easydict")
    m._openmpl_stub = True
    m.EasyDict = EasyDict
    sys.modules["easydict"] = m


def load_reference_config(yaml_path: str):
    """A FRESH copy of the reference's ``core/config.py`` defaults updated from one of its shipped YAMLs
    (``update_config``, config.py:359-373) -- what ``valid_mpl.py:162`` hands to ``get_multiview_mpl_net``."""
    if not os.path.isfile(CONFIG_FILE):
        raise FileNotFoundError(CONFIG_FILE)
    _install_easydict_stub()
    spec = importlib.util.spec_from_file_location("_openmpl_reference_config_%d" % len(_CACHE), CONFIG_FILE)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    _CACHE["cfg%d" % len(_CACHE)] = mod
    mod.update_config(yaml_path)
    return mod.config


def shipped_yamls():
    out = []
    for root, _, files in os.walk(CONFIG_DIR):
        out += [os.path.join(root, f) for f in sorted(files) if f.endswith(".yaml")]
    return sorted(out)
