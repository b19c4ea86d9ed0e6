"""Deterministic, platform-stable generators for synthetic weights and inputs.

Everything here is a pure function of (seed, tensor name, element index): a
counter-based SplitMix64 hash evaluated with numpy uint64 arithmetic.  No
``torch.manual_seed`` / ``np.random`` streams are involved, so this container,
the GPU box and any later round regenerate *identical bits* -- which is what
lets the golden fixtures under ``tests/golden`` omit the (large) weights
(SURVEY.md section 7 step 1, section 8c "Golden vectors").

Distributions follow SURVEY.md section 8d "Synthetic inputs":
  * Linear / Conv1d weights and biases : U(-1/sqrt(fan_in), 1/sqrt(fan_in))
  * LayerNorm / BatchNorm gamma        : U(0.5, 1.5);  beta ~ N(0, 0.1)
  * learned embeddings (zero-init in the reference, multiview_mpl.py:193-219):
    N(0, 0.02) so that those code paths are exercised
  * poses x,y ~ U(-1,1); conf ~ U(0,1) with 5% exact zeros
  * centers ~ N(0, 3) per view, constant over the batch; rays = center + N(0,1)
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, Tuple

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLD = np.uint64(0x9E3779B97F4A7C15)
_C1 = np.uint64(0xBF58476D1CE4E5B9)
_C2 = np.uint64(0x94D049BB133111EB)


def _mix(z: np.ndarray) -> np.ndarray:
    """SplitMix64 finaliser (vectorised, wrap-around uint64 arithmetic)."""
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * _C1
        z = (z ^ (z >> np.uint64(27))) * _C2
        z = z ^ (z >> np.uint64(31))
    return z


def _stream_key(seed: int, name: str, lane: int = 0) -> np.uint64:
    h = zlib.crc32(name.encode("utf-8")) & 0xFFFFFFFF
    k = (int(seed) & 0xFFFFFFFF) << 32 | h
    k = (k + 0x632BE59BD9B4E019 * (lane + 1)) & 0xFFFFFFFFFFFFFFFF
    return _mix(np.array([k], dtype=np.uint64))[0]


def uniform01(seed: int, name: str, n: int, lane: int = 0) -> np.ndarray:
    """n doubles in [0,1), exactly representable (53-bit), element i depends only on i."""
    key = _stream_key(seed, name, lane)
    out = np.empty(n, dtype=np.float64)
    chunk = 1 << 22
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        idx = np.arange(s, e, dtype=np.uint64)
        with np.errstate(over="ignore"):
            z = _mix(key + (idx + np.uint64(1)) * _GOLD)
        out[s:e] = (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return out


def uniform(seed: int, name: str, shape: Tuple[int, ...], lo: float, hi: float) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    u = uniform01(seed, name, n)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def normal(seed: int, name: str, shape: Tuple[int, ...], mean: float, std: float) -> np.ndarray:
    """Irwin-Hall(4) approximation of a normal: only exact +,-,* on doubles, no libm."""
    n = int(np.prod(shape)) if len(shape) else 1
    acc = np.zeros(n, dtype=np.float64)
    for lane in range(4):
        acc += uniform01(seed, name, n, lane=lane + 1)
    z = (acc - 2.0) * 1.7320508075688772  # var(sum of 4 U) = 1/3
    return (mean + std * z).astype(np.float32).reshape(shape)


# --------------------------------------------------------------------------- weights
_EMBED_NAMES = ("Spatial_pos_embed", "pos_3d_embed", "pos_3d_view_coding")


def _is_norm_weight(name: str, shapes: Dict[str, Tuple[int, ...]]) -> bool:
    # LayerNorm / BatchNorm affine: a 1-D ".weight" whose sibling ".bias" has the same
    # shape and which is not a Linear (Linear weights are 2-D, Conv1d 3-D).
    return name.endswith(".weight") and len(shapes[name]) == 1


def make_param(seed: int, name: str, shape: Tuple[int, ...], shapes: Dict[str, Tuple[int, ...]]) -> np.ndarray:
    base = name.split(".")[-1]
    stem = name[: -(len(base) + 1)] if "." in name else ""
    if any(name == e or name.startswith(e + ".") or ("." + e) in name for e in _EMBED_NAMES):
        return normal(seed, name, shape, 0.0, 0.02)
    if base == "num_batches_tracked":
        return np.zeros(shape, dtype=np.int64)
    if base == "running_mean":
        return normal(seed, name, shape, 0.0, 0.1)
    if base == "running_var":
        return uniform(seed, name, shape, 0.5, 1.5)
    if base == "weight":
        if len(shape) == 1:
            return uniform(seed, name, shape, 0.5, 1.5)
        fan_in = int(np.prod(shape[1:]))
        b = 1.0 / np.sqrt(float(fan_in))
        return uniform(seed, name, shape, -b, b)
    if base == "bias":
        w = stem + ".weight"
        if w in shapes and len(shapes[w]) == 1:  # norm beta
            return normal(seed, name, shape, 0.0, 0.1)
        fan_in = int(np.prod(shapes[w][1:])) if w in shapes else int(shape[0])
        b = 1.0 / np.sqrt(float(fan_in))
        return uniform(seed, name, shape, -b, b)
    return normal(seed, name, shape, 0.0, 0.02)


def make_state_dict(shapes: Dict[str, Tuple[int, ...]], seed: int = 0) -> Dict[str, np.ndarray]:
    """name -> float32 ndarray for every entry of ``shapes`` (a state_dict shape map)."""
    return {k: make_param(seed, k, tuple(v), shapes) for k, v in shapes.items()}


def fill_module_(module, seed: int = 0):
    """Overwrite every parameter/buffer of a torch module in place (any device)."""
    import torch

    sd = module.state_dict()
    shapes = {k: tuple(v.shape) for k, v in sd.items()}
    with torch.no_grad():
        for k, v in sd.items():
            arr = make_param(seed, k, shapes[k], shapes)
            v.copy_(torch.from_numpy(arr).to(v.dtype))
    return module


# --------------------------------------------------------------------------- inputs
def make_inputs(batch: int, num_views: int, num_joints: int = 17, seed: int = 0,
                y_range: float = 1.0, step: int = 0):
    """Synthetic hot-path inputs with the value distributions of the reference's
    data pipeline (joints_dataset_mpl.py:772-774, :817-820, :872-904).

    Returns (poses, rays, centers): lists of V float32 arrays of shapes
    (B,J,3), (B,J,3), (B,1,3) -- exactly what function_mpl.py:344-350 hands the model.
    """
    poses, rays, centers = [], [], []
    tag = "s%d" % step
    for v in range(num_views):
        x = uniform(seed, "pose.x.%d.%s" % (v, tag), (batch, num_joints, 1), -1.0, 1.0)
        y = uniform(seed, "pose.y.%d.%s" % (v, tag), (batch, num_joints, 1), -y_range, y_range)
        c = uniform(seed, "pose.c.%d.%s" % (v, tag), (batch, num_joints, 1), 0.0, 1.0)
        drop = uniform(seed, "pose.drop.%d.%s" % (v, tag), (batch, num_joints, 1), 0.0, 1.0) < 0.05
        c = np.where(drop, np.float32(0.0), c).astype(np.float32)
        poses.append(np.concatenate([x, y, c], axis=2))
        cen = normal(seed, "center.%d" % v, (1, 1, 3), 0.0, 3.0)
        cen = np.broadcast_to(cen, (batch, 1, 3)).copy()
        ray = cen + normal(seed, "ray.%d.%s" % (v, tag), (batch, num_joints, 3), 0.0, 1.0)
        centers.append(cen.astype(np.float32))
        rays.append(ray.astype(np.float32))
    return poses, rays, centers
