"""Shared helpers for the test-suite (golden loading, deterministic weights)."""
import json
import os

import numpy as np
import torch

from openmpl_amd import detrng
from oracle import mpl_oracle
from tests.golden.cases import BY_NAME

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    g = {k: z[k] for k in z.files}
    g["flags"] = json.loads(bytes(g["flags"]).decode())
    g["meta"] = json.loads(bytes(g["meta"]).decode())
    assert g["flags"] == BY_NAME[name]["flags"], "fixture is stale w.r.t. cases.py"
    return g


def golden_inputs(g, device="cpu"):
    V = g["poses"].shape[0]
    mk = lambda a: [torch.from_numpy(np.ascontiguousarray(a[v])).to(device) for v in range(V)]
    return mk(g["poses"]), mk(g["rays"]), mk(g["centers"])


_SD_CACHE = {}


def golden_state_dict(name, g=None):
    """Regenerate the weights a fixture was produced with (or read them for the micro case)."""
    if name in _SD_CACHE:
        return _SD_CACHE[name]
    g = g or load_golden(name)
    stored = {k[2:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("w:")}
    if stored:
        sd = stored
    else:
        shapes = mpl_oracle.param_shapes(g["flags"])
        sd = {k: torch.from_numpy(v) for k, v in detrng.make_state_dict(shapes, seed=g["meta"]["wseed"]).items()}
    if len(_SD_CACHE) > 2:
        _SD_CACHE.clear()
    _SD_CACHE[name] = sd
    return sd
